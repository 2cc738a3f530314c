"""GPU parity of the round-3 kernel forms, through the C ABI, against the CPU oracle:

  * the fp32 RESIDUAL STREAM forms of Sandwich-LN (stream in: fp32 x -> 16-bit y; stream out: 16-bit x + fp32 residual
    -> fp32 y, no rounding between the LayerNorm and the add), forward and backward, incl. dropout replay, residual-
    gradient add and column sums; the embedding writing the stream; the decode GEMV's LayerNorm prologue on the stream;
  * the deterministic, once-rounded embedding backward (fp32 segmented sums in ascending token order: what torch's
    embedding_dense_backward does under mpu/layers.py:117-133) -- bit-identical run to run, heavy id repetition,
    ids outside the shard, clamped position ids.

Tolerances: results stored in fp32 are compared at 2e-6 (they carry no storage rounding); 16-bit results at the
single-kernel bars of test_kernels_gpu.py (fp16 3e-3, bf16 2e-2).
"""
import pytest
import torch

from oracle import cogview_oracle as O

pytestmark = pytest.mark.gpu

DTYPES = [torch.float16, torch.bfloat16]
TOL = {torch.float16: 3e-3, torch.bfloat16: 2e-2}
TOL32 = 2e-6


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need an MI355X; run with -m 'not gpu' elsewhere"
    from cogview_amd import ops as _ops
    return _ops


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def rnd(shape, dtype, gen, scale=1.0):
    return (torch.randn(shape, generator=gen) * scale).to(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,h", [(37, 256), (300, 1024), (130, 2560), (9, 4096)])
def test_sandwich_ln_stream_in(ops, dtype, rows, h):
    """LN1 / LN2 / final LN: the fp32 stream is normalised into the storage type; backward returns the fp32 stream
    gradient add_in + LN'(dy)."""
    g = torch.Generator().manual_seed(rows + h)
    x = rnd((rows, h), torch.float32, g, 3.0)
    w, b = (torch.rand(h, generator=g) + 0.5).to(dtype), rnd((h,), dtype, g, 0.1)
    dy, add_in = rnd((rows, h), dtype, g), rnd((rows, h), torch.float32, g)
    xr, wr, br = x.clone().requires_grad_(True), w.float().requires_grad_(True), b.float().requires_grad_(True)
    yr = O.sandwich_layernorm(xr, wr, br, 1e-5)
    yr.backward(dy.float())
    xd = x.cuda()
    amax = ops.absmax(xd)
    assert amax.item() == x.abs().max().item()
    y, mean, rstd = ops.sandwich_ln_fwd(xd, w.cuda(), b.cuda(), 1e-5, amax)
    assert y.dtype == dtype and rel(y, yr) < TOL[dtype]
    dg, db = torch.zeros(h, dtype=dtype, device="cuda"), torch.zeros(h, dtype=dtype, device="cuda")
    dx = ops.sandwich_ln_bwd(dy.cuda(), xd, w.cuda(), mean, rstd, add_in=add_in.cuda(), dgamma=dg, dbeta=db)
    assert dx.dtype == torch.float32 and rel(dx, add_in + xr.grad) < TOL32 * 5
    assert rel(dg, wr.grad) < TOL[dtype] * 2 and rel(db, br.grad) < TOL[dtype] * 2
    dx2 = ops.sandwich_ln_bwd(dy.cuda(), xd, w.cuda(), mean, rstd)                 # without the residual-gradient add
    assert rel(dx2, xr.grad) < TOL32 * 5


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,h", [(41, 512), (300, 1024), (130, 2560)])
def test_sandwich_ln_stream_out(ops, dtype, rows, h):
    """LN3 / LN4: y = residual + LN(x) evaluated and stored in fp32 (abs-max over the fp32 values); backward takes the
    fp32 stream gradient, replays the dropout mask of the branch output and emits the branch gradient in 16 bits with
    its column sums (the bias gradient of the Linear that produced x)."""
    g = torch.Generator().manual_seed(rows * 3 + h)
    x, res = rnd((rows, h), dtype, g, 2.0), rnd((rows, h), torch.float32, g, 4.0)
    w, b = (torch.rand(h, generator=g) + 0.5).to(dtype), rnd((h,), dtype, g, 0.1)
    xd = x.cuda()
    amax = ops.absmax(xd)
    slot = ops.new_absmax_slot(xd.device)
    y, mean, rstd = ops.sandwich_ln_fwd(xd, w.cuda(), b.cuda(), 1e-5, amax, residual=res.cuda(), absmax_out=slot)
    ref = res + O.sandwich_layernorm(x.float(), w.float(), b.float())
    assert y.dtype == torch.float32 and rel(y, ref) < TOL32
    assert slot.item() == y.abs().max().item()
    dy = rnd((rows, h), torch.float32, g)
    xr = x.float().requires_grad_(True)
    O.sandwich_layernorm(xr, w.float(), b.float()).backward(dy)
    mask = torch.from_numpy(O.dropout_keep_mask(rows * h, 0.1, 7, 3)).view(rows, h)
    cs = torch.zeros(h, dtype=dtype, device="cuda")
    dx = ops.sandwich_ln_bwd(dy.cuda(), xd, w.cuda(), mean, rstd, dropout=(0.1, 7, 3), colsum=cs)
    assert dx.dtype == dtype and rel(dx, xr.grad * mask) < TOL[dtype]
    assert rel(cs, dx.float().cpu().sum(0)) < TOL[dtype] * 2


@pytest.mark.parametrize("dtype", DTYPES)
def test_stream_add_and_absmax(ops, dtype):
    g = torch.Generator().manual_seed(2)
    a, b = rnd((33, 256), torch.float32, g, 2.0), rnd((33, 256), dtype, g)
    slot = ops.new_absmax_slot(torch.device("cuda"))
    out = ops.add(a.cuda(), b.cuda(), absmax_out=slot)
    assert out.dtype == torch.float32 and torch.equal(out.cpu(), a + b.float())
    assert slot.item() == out.abs().max().item()
    a[3, 5] = float("nan")
    assert ops.absmax(a.cuda()).isnan().item()                    # x.abs().max() propagates NaN


@pytest.mark.parametrize("dtype", DTYPES)
def test_embedding_writes_the_fp32_stream(ops, dtype):
    g = torch.Generator().manual_seed(6)
    V, P, h, b, s = 96, 48, 128, 2, 40
    table, pos_table = rnd((V, h), dtype, g), rnd((P, h), dtype, g)
    ids = torch.randint(0, 128, (b, s), generator=g)
    pos = torch.arange(s).unsqueeze(0).expand(b, -1)
    vs = 16
    inside = (ids >= vs) & (ids < vs + V)
    word = torch.where(inside.unsqueeze(-1), table.float()[(ids - vs).clamp(0, V - 1)], torch.zeros(1))
    mask = torch.from_numpy(O.dropout_keep_mask(b * s * h, 0.1, 3, 1)).view(b, s, h)
    slot = ops.new_absmax_slot(torch.device("cuda"))
    out = ops.embedding_fwd(ids.cuda(), table.cuda(), vs, pos.cuda(), pos_table.cuda(), dropout=(0.1, 3, 1), absmax_out=slot,
                            out_f32=True)
    assert out.dtype == torch.float32 and rel(out, (word + pos_table.float()[pos]) * mask) < TOL32
    assert slot.item() == out.abs().max().item()
    # backward from the fp32 stream gradient
    dout = rnd((b, s, h), torch.float32, g)
    dt, dp = torch.zeros((V, h), dtype=dtype, device="cuda"), torch.zeros((P, h), dtype=dtype, device="cuda")
    ops.embedding_bwd(dout.cuda(), ids.cuda(), dt, vs, pos.cuda(), dp, dropout=(0.1, 3, 1))
    dm = dout * mask
    rt = torch.zeros(V, h).index_add_(0, (ids - vs).clamp(0, V - 1).view(-1), (dm * inside.unsqueeze(-1)).view(-1, h))
    rp = torch.zeros(P, h).index_add_(0, pos.reshape(-1), dm.view(-1, h))
    assert torch.equal(dt.cpu(), rt.to(dtype)) or rel(dt, rt) < TOL[dtype] * 0.5
    assert rel(dp, rp) < TOL[dtype] * 0.5


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("d32", [False, True])
def test_embedding_backward_is_deterministic_and_rounds_once(ops, dtype, d32):
    """Heavy repetition (a handful of ids over thousands of tokens, as padding / frequent image codes produce), ids
    outside the shard, position ids repeated per batch row and out of range (clamped like the forward): the table
    gradient must equal round(previous + fp32 sum over the tokens) -- the reference's embedding_dense_backward rounds
    once -- and two runs must agree bit for bit."""
    g = torch.Generator().manual_seed(17)
    V, P, h, b, s, vs = 300, 70, 2560 + 8, 6, 700, 40           # h > 2048: two column slabs, the last one ragged
    ids = torch.randint(0, 400, (b, s), generator=g)
    ids[:, ::3] = 77                                            # one id on a third of all tokens
    ids[2, 100:400] = 123
    pos = (torch.arange(s) % 90).unsqueeze(0).expand(b, -1).contiguous()      # beyond P - 1: clamped
    dout = rnd((b, s, h), torch.float32 if d32 else dtype, g)
    prev_t, prev_p = rnd((V, h), dtype, g), rnd((P, h), dtype, g)
    mask = torch.from_numpy(O.dropout_keep_mask(b * s * h, 0.1, 5, 9)).view(b, s, h)
    dm = (dout.double() * mask.double())
    inside = (ids >= vs) & (ids < vs + V)
    rt = prev_t.double().index_add_(0, (ids - vs).clamp(0, V - 1).view(-1), (dm * inside.unsqueeze(-1)).view(-1, h))
    rp = prev_p.double().index_add_(0, pos.clamp(0, P - 1).reshape(-1), dm.view(-1, h))
    runs = []
    for _ in range(2):
        dt, dp = prev_t.cuda().clone(), prev_p.cuda().clone()
        ops.embedding_bwd(dout.cuda(), ids.cuda(), dt, vs, pos.cuda(), dp, dropout=(0.1, 5, 9))
        runs.append((dt.cpu(), dp.cpu()))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    for got, ref in ((runs[0][0], rt), (runs[0][1], rp)):
        # ONE rounding of an fp32 sum: equal to the rounded exact (fp64) value except where the fp32 accumulation error
        # (~1e-6 relative over a few thousand terms) flips a round-to-nearest decision (~0.2 % of the elements); never
        # further than one unit in the last place.  The 16-bit-atomics kernel of rounds 1-2 rounded once per add.
        exact = ref.to(dtype)
        half_ulp_rel = 2.0 ** -11 if dtype == torch.float16 else 2.0 ** -8
        # (+ the fp32 accumulation error itself, which matters where 1400 unit-size terms cancel to a small sum)
        assert bool(((got.double() - ref).abs() <= ref.abs() * (2 * half_ulp_rel) + 3e-4).all())
        assert (got == exact).float().mean().item() > 0.98
    # rows nobody referenced keep their previous value
    untouched = torch.ones(V, dtype=torch.bool)
    untouched[(ids[inside] - vs).unique()] = False
    assert torch.equal(runs[0][0][untouched], prev_t[untouched])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,K,N,post,gelu", [(1, 512, 768, False, False), (3, 1024, 512, True, True), (8, 2560, 1024, True, False),
                                              (2, 4096, 256, True, True), (1, 2560, 7680, True, False), (2, 2560, 24, False, True),
                                              (4, 1536, 136, True, False), (1, 1024, 40, False, True)])
def test_gemv_layernorm_prologue_on_the_fp32_stream(ops, dtype, M, K, N, post, gelu):
    """cogv_gemv_ln with stream_f32: residual / t (and the plain input) are fp32 rows; t = residual + LN_post(z) without
    intermediate roundings == what the Sandwich-LN kernel writes in its stream-out form, bit for bit."""
    g = torch.Generator().manual_seed(M * 100 + K + N)
    z = rnd((M, K), dtype if post else torch.float32, g, 3.0)
    res = rnd((M, K), torch.float32, g)
    w, bias = rnd((N, K), dtype, g, 0.05), rnd((N,), dtype, g)
    gp, bp = (1.0 + 0.1 * torch.randn(K, generator=g)).to(dtype), (0.1 * torch.randn(K, generator=g)).to(dtype)
    gn, bn = (1.0 + 0.1 * torch.randn(K, generator=g)).to(dtype), (0.1 * torch.randn(K, generator=g)).to(dtype)
    eps = 1e-5
    zd, resd = z.cuda(), res.cuda()
    zmax = ops.absmax(zd)
    if post:
        slot_t = ops.new_absmax_slot(zd.device)
        t_ref, _, _ = ops.sandwich_ln_fwd(zd, gp.cuda(), bp.cuda(), eps, zmax, residual=resd, absmax_out=slot_t, save_stats=False)
    else:
        t_ref, slot_t = zd, zmax
    x_ref, _, _ = ops.sandwich_ln_fwd(t_ref, gn.cuda(), bn.cuda(), eps, slot_t, save_stats=False)
    out_ref = ops.gemm(x_ref, w.cuda(), bias=bias.cuda(), gelu=gelu)
    out, t = ops.gemv_ln(zd, w.cuda(), bias.cuda(), gn.cuda(), bn.cuda(), eps, z_absmax=zmax, post=(gp.cuda(), bp.cuda()) if post else None,
                         residual=resd if post else None, want_t=post, gelu=gelu)
    if post:
        assert t.dtype == torch.float32 and rel(t, t_ref) < 1e-6
        out2, t2 = ops.gemv_ln(zd, w.cuda(), bias.cuda(), gn.cuda(), bn.cuda(), eps, z_absmax=None, post=(gp.cuda(), bp.cuda()),
                               residual=resd, want_t=True, gelu=gelu)                 # max|z| taken inside the kernel
        assert torch.equal(out2, out) and torch.equal(t2, t)
    assert out.dtype == dtype
    assert rel(out, out_ref) < (2e-3 if dtype == torch.float16 else 1.5e-2)
    tf = (res + O.sandwich_layernorm(z.float(), gp.float(), bp.float(), eps)) if post else z
    xin = O.sandwich_layernorm(tf, gn.float(), bn.float(), eps).to(dtype).float()
    ref = O.linear(xin, w.float(), bias.float())
    if gelu:
        ref = O.gelu(ref.to(dtype).float())
    assert rel(out, ref) < TOL[dtype]


def test_add_gradient_dtypes_and_operand_order():
    """functional.add joins the fp32 residual stream (first operand) with a 16-bit branch: each input receives its gradient
    in ITS OWN dtype (explicitly, round-3 advisor item), and the swapped order -- which would reinterpret fp32 bytes as 16-bit
    values -- is refused."""
    from cogview_amd import _lib as L, functional as F_, ops
    a = torch.randn(4, 64, device="cuda", dtype=torch.float32, requires_grad=True)
    b = torch.randn(4, 64, device="cuda", dtype=torch.float16, requires_grad=True)
    y = F_.add(a, b)
    assert y.dtype == torch.float32
    y.backward(torch.ones_like(y))
    assert a.grad.dtype == torch.float32 and b.grad.dtype == torch.float16
    assert torch.equal(a.grad, torch.ones_like(a)) and torch.equal(b.grad, torch.ones_like(b))
    with pytest.raises(L.CogviewHipError):
        ops.add(b.detach(), a.detach())


def test_clip_grad_norm_is_one_pass_per_storage_and_exact():
    """mpu.clip_grad_norm (mpu/grads.py:28-74): gradients that are views of ONE buffer (the flat arena) go through a single
    chunk table -- one cogv_grad_stats launch, not one per parameter -- together with a stand-alone tensor, a view at an offset
    that is not a multiple of 8 elements, and an fp32 gradient; the norm equals the float64 reference and the clip scales
    every tensor."""
    from cogview_amd import mpu, ops
    g = torch.Generator().manual_seed(3)
    arena = (torch.randn(10000, generator=g) * 0.5).half().cuda()
    views = [arena[0:1000], arena[1024:1024 + 3000].view(30, 100), arena[4096:4096 + 333], arena[5003:5003 + 64]]   # last: odd offset
    lone = (torch.randn(777, generator=g)).to(torch.bfloat16).cuda()
    f32 = torch.randn(50, generator=g).cuda()
    params = []
    for t in views + [lone, f32]:
        p = torch.nn.Parameter(torch.zeros_like(t))
        p.grad = t
        p.model_parallel = False
        params.append(p)
    ref = sum(float((p.grad.double() ** 2).sum()) for p in params) ** 0.5
    calls = []
    real = ops.grad_stats
    ops.grad_stats = lambda *a, **k: (calls.append(a[1].numel()), real(*a, **k))[1]
    try:
        before = [p.grad.clone() for p in params]
        total = mpu.clip_grad_norm(params, ref / 2)
    finally:
        ops.grad_stats = real
    assert abs(total - ref) < 1e-6 * ref
    # arena views sharing a table (3 chunks), the odd-offset view, the bf16 tensor, the fp32 tensor (cogv_grad_stats takes fp32 too)
    assert sorted(calls) == [1, 1, 1, 3], calls
    for p, b0 in zip(params, before):
        assert ((p.grad.float() - 0.5 * b0.float()).abs().max() <= 4e-3 * b0.float().abs().max()).item()


def test_fp32_gradients_take_the_same_kernels():
    """mpu/grads.py:62-84 on fp32 gradients (FP16_Optimizer's master gradients outside the flat path, fp32 models): the 2-norm
    and the overflow flag are cogv_grad_stats in its fp32 instantiation, the inf-norm cogv_absmax -- no torch reductions; views
    of one fp32 buffer share one launch; a non-finite value raises the flag; lengths that are not multiples of 8 / 4."""
    from cogview_amd import mpu, ops
    from cogview_amd.fp16.loss_scaler import DynamicLossScaler
    g = torch.Generator().manual_seed(11)
    buf = torch.randn(40000, generator=g).cuda()
    grads = [buf[0:12345], buf[12352:12352 + 20000].view(200, 100), torch.randn(1001, generator=g).cuda(),
             torch.randn(8, 128, generator=g).cuda()]
    params = []
    for t in grads:
        p = torch.nn.Parameter(torch.zeros_like(t))
        p.grad = t
        p.model_parallel = False
        params.append(p)
    ref2 = sum(float((p.grad.double() ** 2).sum()) for p in params) ** 0.5
    refinf = max(float(p.grad.abs().max()) for p in params)
    calls = []
    real = ops.grad_stats
    ops.grad_stats = lambda *a, **k: (calls.append((a[0].dtype, a[1].numel())), real(*a, **k))[1]
    try:
        total = mpu.clip_grad_norm(params, 1e9)
        assert not DynamicLossScaler().has_overflow_serial(params)
        params[1].grad.view(-1)[19999] = float("nan")
        assert DynamicLossScaler().has_overflow_serial(params)
        params[1].grad.view(-1)[19999] = 0.25
        params[2].grad[1000] = float("inf")             # in the scalar tail of a length that is not a multiple of 8
        assert DynamicLossScaler().has_overflow_serial(params)
        params[2].grad[1000] = 0.0
    finally:
        ops.grad_stats = real
    assert abs(total - ref2) < 1e-6 * ref2
    assert all(dt == torch.float32 for dt, _ in calls) and sorted(n for _, n in calls[:3]) == [1, 1, 2], calls
    assert mpu.clip_grad_norm(params, 1e9, float("inf")) == pytest.approx(refinf, rel=1e-7)
    before = [p.grad.clone() for p in params]
    now2 = sum(float((b0.double() ** 2).sum()) for b0 in before) ** 0.5          # two elements were edited above
    assert mpu.clip_grad_norm(params, now2 / 4) == pytest.approx(now2, rel=1e-6)
    for p, b0 in zip(params, before):
        assert torch.allclose(p.grad, 0.25 * b0, rtol=1e-5, atol=0)


def test_grad_stats_and_clip_grad_norm_on_a_side_stream():
    """The norm / overflow pass under torch.cuda.stream(non-default): its per-stream scratch is keyed on the stream handle (round-4
    advisor finding: a ctypes handle in a '%x' format raised TypeError on every stream but the default one); same value as on
    the default stream, and the two streams get separate scratch buffers."""
    from cogview_amd import mpu, ops
    g = torch.Generator().manual_seed(5)
    flat = (torch.randn(3 * 4096 + 100, generator=g)).half().cuda()
    ref = float((flat.double() ** 2).sum()) ** 0.5

    def run():
        p = torch.nn.Parameter(torch.zeros_like(flat))
        p.grad = flat.clone()
        p.model_parallel = False
        return mpu.clip_grad_norm([p], 1e9)

    want = run()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    n_ws = len(ops._WS)
    with torch.cuda.stream(side):
        got = run()
    side.synchronize()
    assert got == want and abs(got - ref) < 1e-6 * ref
    assert len(ops._WS) == n_ws + 1                        # a scratch buffer of its own for the side stream


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("rows", [7, 1000, 26112])
def test_layernorm_backward_lean_dropout_replay_form_equals_the_regular_one(dtype, rows, monkeypatch):
    """COGV_LN_BWD_LEAN=1 (round 5): LN3' / LN4' (fp32 stream gradient in, dropout replay, column sums, no add_in) in the form that
    keeps the rows' 16-bit x raw across the barrier (128 registers, three workgroups per CU): dx bit-identical with the regular
    form, the three column sums equal within the fp32 summation order (different workgroup counts)."""
    from cogview_amd import ops
    g = torch.Generator().manual_seed(rows)
    h = 2560
    x = torch.randn(rows, h, generator=g).to(dtype).cuda()
    gam = (torch.rand(h, generator=g) + 0.5).to(dtype).cuda()
    bet = torch.zeros(h, dtype=dtype, device="cuda")
    stream = torch.randn(rows, h, generator=g).cuda()
    dy = torch.randn(rows, h, generator=g).cuda()
    _, mean, rstd = ops.sandwich_ln_fwd(x, gam, bet, 1e-5, ops.absmax(x), residual=stream)
    res = {}
    for lean in ("0", "1"):
        monkeypatch.setenv("COGV_LN_BWD_LEAN", lean)
        dg, db, cs = (torch.zeros(h, dtype=dtype, device="cuda") for _ in range(3))
        dx = ops.sandwich_ln_bwd(dy, x, gam, mean, rstd, dropout=(0.1, 3, 4), dgamma=dg, dbeta=db, colsum=cs)
        res[lean] = (dx, dg, db, cs)
    assert torch.equal(res["0"][0], res["1"][0])
    assert abs(float((res["1"][0] == 0).float().mean()) - 0.1) < (0.02 if rows > 100 else 0.1)
    for a, b in zip(res["0"][1:], res["1"][1:]):
        assert ((a.float() - b.float()).norm() / (a.float().norm() + 1e-30)).item() < 2e-3


def _bits16(t):
    return t.contiguous().view(torch.int16)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(300, 2560, 128), (1100, 1024, 64), (70, 256, 64), (256, 2560, 2560)])
def test_gemm_dropout_epilogue_marks_dropped_elements_with_minus_zero(ops, dtype, M, N, K):
    """Marked zeros (round 6, gemm_shared.cuh epilogue8): the GEMM dropout epilogue writes a dropped element as -0.0 (16-bit
    pattern 0x8000) and NO kept element as -0.0 -- so the output carries the oracle's keep mask bit for bit -- in every GEMM
    generation the shapes select (256-tile persistent kernel, the smaller tiles, K not a multiple of 256).  Column 0 of the weight
    is built so that kept outputs of column 0 underflow to a NEGATIVE value below the storage type's smallest subnormal
    (fp16; an fp32 denormal for bf16): rounding alone would produce -0.0 there; they must come out as +0.0."""
    g = torch.Generator().manual_seed(M + N)
    a, w = rnd((M, K), dtype, g), rnd((N, K), dtype, g, 0.05)
    bias = rnd((N,), dtype, g, 0.1)
    tiny = 1e-9 if dtype == torch.float16 else 1e-30
    w[0].zero_(); w[0, 0] = -(tiny ** 0.5) if dtype == torch.float16 else -1e-20
    a[:, 0] = (tiny ** 0.5) if dtype == torch.float16 else 1e-20        # bf16: 1e-20 * -1e-20 = -1e-40, an fp32 denormal
    bias[0] = 0.0
    mask = torch.from_numpy(O.dropout_keep_mask(M * N, 0.1, 11, 6)).view(M, N)
    slot = ops.new_absmax_slot(torch.device("cuda"))
    out = ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), dropout=(0.1, 11, 6), absmax=slot)
    marked = (_bits16(out).cpu() == -32768)                               # 0x8000
    assert torch.equal(marked, mask == 0)
    col0 = out[:, 0].cpu()
    assert bool((col0 == 0).all()) and torch.equal(torch.signbit(col0), mask[:, 0] == 0)
    ref = (a.float() @ w.float().t() + bias.float()) * mask
    assert rel(out, ref) < TOL[dtype]
    assert abs(slot.item() - out.float().abs().max().item()) == 0.0


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,N,K,cuts", [(1024, 512, 256, (0, 256, 768, 1024)), (600, 256, 64, (0, 256, 600)), (300, 1024, 128, (0, 40, 300))])
def test_gemm_row_chunks_draw_the_whole_tensor_dropout_mask(ops, dtype, M, N, K, cuts):
    """cogv_gemm_desc.dropout_row0: a GEMM over rows [r0, r1) of a larger output draws the mask of those rows of the WHOLE tensor
    -- the chunks together are bit-identical to the one-call result (marked zeros included), with all CUs and with CUs reserved
    for a concurrent collective (cogv_gemm_reserve_cus)."""
    g = torch.Generator().manual_seed(M + K)
    a, w = rnd((M, K), dtype, g).cuda(), rnd((N, K), dtype, g, 0.1).cuda()
    bias = rnd((N,), dtype, g, 0.1).cuda()
    whole = ops.gemm(a, w, bias=bias, dropout=(0.1, 5, 9))
    for reserve in (0, 16, 100000):                 # 100000: more than the GPU has -- the launch keeps 8 workgroups
        prev = ops.gemm_reserve_cus(reserve)
        try:
            out = torch.empty_like(whole)
            for r0, r1 in zip(cuts[:-1], cuts[1:]):
                ops.gemm(a[r0:r1], w, bias=bias, dropout=(0.1, 5, 9), dropout_row0=r0, out=out[r0:r1])
        finally:
            assert ops.gemm_reserve_cus(prev) == reserve
        assert torch.equal(_bits16(out), _bits16(whole))
    mask = torch.from_numpy(O.dropout_keep_mask(M * N, 0.1, 5, 9)).view(M, N)
    assert torch.equal((_bits16(whole).cpu() == -32768), mask == 0)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("rows,h,stream_out", [(7, 2560, True), (1000, 2560, True), (26112, 2560, True), (300, 1024, True),
                                               (70, 256, True), (130, 2560, False), (300, 1024, False)])
def test_layernorm_backward_reads_the_mask_from_marked_zeros_bit_identically(ops, dtype, rows, h, stream_out):
    """cogv_sandwich_ln_bwd_marked (round 6): LN3' / LN4' take the hidden-dropout mask from their input's marked zeros instead of
    re-hashing it.  x is a real output of the marking GEMM epilogue; dx must be bit-identical to the regenerating form with the
    GEMM's (p, seed, stream), and so must dgamma / dbeta / column sums (same kernel structure, same summation order).  Both
    stream forms (fp32 gradient in, and all-16-bit), lean / two-row / four-row kernels (h = 2560 / 1024 / 256)."""
    g = torch.Generator().manual_seed(rows + h)
    K = 64
    a, w = rnd((rows, K), dtype, g), rnd((h, K), dtype, g, 0.2)
    drop = (0.1, 21, 8)
    slot = ops.new_absmax_slot(torch.device("cuda"))
    x = ops.gemm(a.cuda(), w.cuda(), dropout=drop, absmax=slot)
    gam = (torch.rand(h, generator=g) + 0.5).to(dtype).cuda()
    bet = torch.zeros(h, dtype=dtype, device="cuda")
    if stream_out:
        stream = torch.randn(rows, h, generator=g).cuda()
        dy = torch.randn(rows, h, generator=g).cuda()
        _, mean, rstd = ops.sandwich_ln_fwd(x, gam, bet, 1e-5, slot, residual=stream)
        add = None
    else:
        dy = rnd((rows, h), dtype, g).cuda()
        _, mean, rstd = ops.sandwich_ln_fwd(x, gam, bet, 1e-5, slot)
        add = rnd((rows, h), dtype, g).cuda() if rows == 130 else None
    # (regenerating form, marked with two rows in flight -- the regenerating form's geometry --, marked as shipped: four rows in
    #  flight at wide rows.  dx is a per-row result: bit-identical in all three.  dgamma / dbeta / column sums are sums over the
    #  rows a workgroup walks: the same rows in the same order at two rows in flight -> bit-identical; at four the rows are dealt
    #  to the workgroups differently -> equal up to the fp32 summation order, i.e. to one unit in the last place of the 16-bit sums.)
    import os
    res = []
    for marked, rows_in_flight in ((False, None), (True, "2"), (True, None)):
        dg, db, cs = (torch.zeros(h, dtype=dtype, device="cuda") for _ in range(3))
        old = os.environ.pop("COGV_LN_BWD_MARKED_ROWS", None)
        if rows_in_flight is not None:
            os.environ["COGV_LN_BWD_MARKED_ROWS"] = rows_in_flight
        try:
            dx = ops.sandwich_ln_bwd(dy, x, gam, mean, rstd, add_in=add, dropout=drop, dgamma=dg, dbeta=db, colsum=cs, marked=marked)
        finally:
            os.environ.pop("COGV_LN_BWD_MARKED_ROWS", None)
            if old is not None:
                os.environ["COGV_LN_BWD_MARKED_ROWS"] = old
        res.append((dx, dg, db, cs))
    for u, v in zip(res[0], res[1]):
        assert torch.equal(_bits16(u), _bits16(v))
    assert torch.equal(_bits16(res[0][0]), _bits16(res[2][0]))
    ulp = 2.0 ** (-10 if dtype == torch.float16 else -7)
    for u, v in zip(res[0][1:], res[2][1:]):
        assert float((u.float() - v.float()).abs().max()) <= ulp * max(float(u.float().abs().max()), 1e-6)
    if add is None:
        assert abs(float((res[1][0] == 0).float().mean()) - 0.1) < (0.02 if rows > 100 else 0.1)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("rows,h,p_drop", [(7, 2560, 0.1), (1000, 2560, 0.1), (26112, 2560, 0.1), (300, 2048, 0.0), (513, 4096, 0.1), (1, 1600, 0.1), (33, 3080, 0.1)])
def test_layernorm_backward_pair_is_the_two_launches_bit_for_bit(ops, dtype, rows, h, p_drop):
    """cogv_sandwich_ln_bwd_pair (round 6): LN2' (stream in + add) and LN3' (stream out, mask from marked zeros) of a layer in one
    pass over the rows.  dy (fp32) and d_ao (16-bit) must be bit-identical to the two launches; the five column reductions agree
    to one unit in the last place of their 16-bit storage (rows are dealt to workgroups differently: fp32 summation order).
    Overwrite and accumulate forms."""
    g = torch.Generator().manual_seed(rows + h)
    K = 64
    a, w = rnd((rows, K), dtype, g), rnd((h, K), dtype, g, 0.2)
    drop = (p_drop, 31, 4) if p_drop > 0 else None
    slot = ops.new_absmax_slot(torch.device("cuda"))
    ao = ops.gemm(a.cuda(), w.cuda(), dropout=drop, absmax=slot)                  # LN3's input: a marking GEMM's output
    gam3 = (torch.rand(h, generator=g) + 0.5).to(dtype).cuda()
    gam2 = (torch.rand(h, generator=g) + 0.5).to(dtype).cuda()
    bet = torch.zeros(h, dtype=dtype, device="cuda")
    x = torch.randn(rows, h, generator=g).cuda()                                  # the residual stream entering the layer
    slot_y = ops.new_absmax_slot(torch.device("cuda"))
    y, m3, r3 = ops.sandwich_ln_fwd(ao, gam3, bet, 1e-5, slot, residual=x, absmax_out=slot_y)       # y = x + LN3(ao), fp32
    _, m2, r2 = ops.sandwich_ln_fwd(y, gam2, bet, 1e-5, slot_y)                                      # c = LN2(y)
    dc = rnd((rows, h), dtype, g).cuda()
    dout = torch.randn(rows, h, generator=g).cuda()
    for accumulate in (False, True):
        init = [rnd((h,), dtype, g).cuda() for _ in range(5)]
        ref = [t.clone() for t in init]
        dy_ref = ops.sandwich_ln_bwd(dc, y, gam2, m2, r2, add_in=dout, dgamma=ref[0], dbeta=ref[1], accumulate=accumulate)
        dao_ref = ops.sandwich_ln_bwd(dy_ref, ao, gam3, m3, r3, dropout=drop, dgamma=ref[2], dbeta=ref[3], colsum=ref[4],
                                      accumulate=accumulate, marked=True)
        got = [t.clone() for t in init]
        dy, dao = ops.sandwich_ln_bwd_pair(dc, y, gam2, m2, r2, dout, ao, gam3, m3, r3, dropout_p=p_drop, dgamma2=got[0],
                                           dbeta2=got[1], dgamma3=got[2], dbeta3=got[3], colsum=got[4], accumulate=accumulate)
        assert torch.equal(dy, dy_ref)
        assert torch.equal(_bits16(dao), _bits16(dao_ref))
        ulp = 2.0 ** (-10 if dtype == torch.float16 else -7)
        for u, v in zip(ref, got):
            assert float((u.float() - v.float()).abs().max()) <= ulp * max(float(u.float().abs().max()), 1e-6)
    if p_drop > 0:
        assert abs(float((dao == 0).float().mean()) - p_drop) < (0.02 if rows > 100 else 0.1)

