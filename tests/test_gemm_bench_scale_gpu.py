"""GEMM parity at the shapes bench.py actually runs (4B config, b = 24 -> M = 26112 rows): the persistent kernel's
per-XCD work queues, tails and the `sched` self-reset are exercised with 1000+ tiles over 256 workgroups, not only by
"the loss goes down".

Reference: the contraction's definition (oracle.linear = x W^T (+ b), reference mpu/layers.py:243,319 and
model/gpt2_modeling.py:117) evaluated (a) by the CPU oracle in fp32 on a 128-row slab of the output and (b) for the
whole output by an fp32 torch.matmul on the GPU over the same 16-bit inputs (checker only; (a) ties it to the oracle).
Every launch is issued twice back to back and must be bit-identical (work-queue counters reset by the last workgroup).
Tolerance: one rounding of the fp32 accumulator to the storage type -- rel-L2 fp16 1e-3, bf16 6e-3.
"""
import pytest
import torch

from oracle import cogview_oracle as O

pytestmark = pytest.mark.gpu

TOL = {torch.float16: 1e-3, torch.bfloat16: 6e-3}
M_BENCH = 26112          # 24 sequences x 1088 positions


def rel(a, b):
    return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()


def rnd(shape, dtype, seed, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(shape, generator=g, device="cuda") * scale).to(dtype)


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from cogview_amd import ops as _ops
    return _ops


UNIT = {torch.float16: 2.0 ** -11, torch.bfloat16: 2.0 ** -8}       # half an ulp, relative: the one rounding of the fp32 accumulator


def _tile_guard(out, ref, dtype, where):
    """Worst element of every 256 x 256 output tile (round-4 verdict: the aggregate rel-L2 lets a single corrupted row through
    in bf16): |out - ref| <= 1.5 x half-ulp x the tile's largest |ref| -- one rounding of a correct fp32 sum stays below
    1.0 x; a row of wrong products is off by about the tile's RMS, hundreds of times the bound.  out, ref: one row block."""
    import torch.nn.functional as Fn
    d = (out.float() - ref).abs()[None, None]
    worst = Fn.max_pool2d(d, 256, ceil_mode=True)[0, 0]
    big = Fn.max_pool2d(ref.abs()[None, None], 256, ceil_mode=True)[0, 0]
    bad = worst > 1.5 * UNIT[dtype] * big + 1e-30
    assert not bool(bad.any()), (where, [(int(i), int(j), float(worst[i, j]), float(big[i, j])) for i, j in bad.nonzero()[:4].tolist()])


def _slab_check(out, a_rows, w, bias, rows, dtype):
    """CPU oracle on a slab: out[rows] == oracle.linear(a[rows], w, bias)."""
    ref = O.linear(a_rows.float().cpu(), w.float().cpu(), None if bias is None else bias.float().cpu())
    assert rel(out[rows].float().cpu(), ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("N,K", [(10240, 2560), (2560, 10240), (7680, 2560), (58240, 2560)])
def test_forward_nt_at_bench_scale(ops, dtype, N, K):
    """h->4h, 4h->h, QKV and the tied-logits GEMM of the 4B model (generation 4 explicitly: kernel_variant 10)."""
    a, w, bias = rnd((M_BENCH, K), dtype, 1), rnd((N, K), dtype, 2, 0.02), rnd((N,), dtype, 3)
    out = ops.gemm(a, w, bias=bias, variant=10)
    again = ops.gemm(a, w, bias=bias, variant=10)
    assert torch.equal(out, again), "back-to-back launches differ: work-queue state leaked between launches"
    assert torch.equal(out, ops.gemm(a, w, bias=bias)), "auto dispatch does not pick generation 4 at this shape"
    # whole output against fp32 matmul on the GPU, in row blocks to bound the fp32 temporaries
    num = den = 0.0
    for r0 in range(0, M_BENCH, 4352):
        ref = a[r0:r0 + 4352].float() @ w.float().t() + bias.float()
        d = out[r0:r0 + 4352].float() - ref
        num += float(d.double().pow(2).sum())
        den += float(ref.double().pow(2).sum())
        _tile_guard(out[r0:r0 + 4352], ref, dtype, ("forward", N, K, r0))
    assert (num / den) ** 0.5 < TOL[dtype]
    for r0 in (0, M_BENCH - 128, 13000):                      # first tile row, the last one, one in the middle
        _slab_check(out, a[r0:r0 + 128], w, bias, slice(r0, r0 + 128), dtype)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("N,K", [(2560, 10240), (10240, 2560), (2560, 58240)])
def test_dgrad_nn_at_bench_scale(ops, dtype, N, K):
    """dX[M,N] = dY[M,K] W[K,N] (W stored [K][N], read with transposing LDS loads)."""
    dy, w = rnd((M_BENCH, K), dtype, 4, 0.05), rnd((K, N), dtype, 5, 0.02)
    out = ops.gemm(dy, w, trans_b=True, variant=10)
    assert torch.equal(out, ops.gemm(dy, w, trans_b=True, variant=10))
    num = den = 0.0
    for r0 in range(0, M_BENCH, 4352):
        ref = dy[r0:r0 + 4352].float() @ w.float()
        d = out[r0:r0 + 4352].float() - ref
        num += float(d.double().pow(2).sum())
        den += float(ref.double().pow(2).sum())
        _tile_guard(out[r0:r0 + 4352], ref, dtype, ("dgrad", N, K, r0))
    assert (num / den) ** 0.5 < TOL[dtype]
    r0 = M_BENCH - 128
    _slab_check(out, dy[r0:], w.t().contiguous(), None, slice(r0, M_BENCH), dtype)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_grouped_wgrad_at_bench_scale(ops, dtype):
    """The four weight gradients of one 4B layer in ONE persistent launch: K = 26112 rows of tokens, 1200 tiles over
    256 workgroups, accumulate into existing gradients; twice back to back (second launch accumulates again)."""
    K, h = M_BENCH, 2560
    shapes = [(h, 4 * h), (4 * h, h), (h, h), (3 * h, h)]          # (out features, in features): W2, W1, Wo, Wqkv
    probs, refs = [], []
    for i, (Mo, Ni) in enumerate(shapes):
        dy, x = rnd((K, Mo), dtype, 10 + i, 0.05), rnd((K, Ni), dtype, 20 + i)
        prev = rnd((Mo, Ni), dtype, 30 + i, 2.0)
        probs.append((dy, x, prev.clone()))
        refs.append((dy.float().t() @ x.float(), prev.float()))
    ops.gemm_grouped(probs, trans_a=True, trans_b=True, accumulate=True)
    for i, ((dy, x, out), (prod, prev)) in enumerate(zip(probs, refs)):
        assert rel(out.float(), prod + prev) < TOL[dtype]
        _tile_guard(out, prod + prev, dtype, ("wgrad", i))
    first = [out.clone() for _, _, out in probs]
    ops.gemm_grouped(probs, trans_a=True, trans_b=True, accumulate=True)
    for (dy, x, out), (prod, prev), f in zip(probs, refs, first):
        assert rel(out.float(), prod + f.float()) < TOL[dtype]
    # a 128-row slab of dW_qkv against the CPU oracle's contraction (dW = dY^T X is linear(dY^T, X^T))
    dy, x, _ = probs[3]
    ref = O.linear(dy[:, :128].float().t().cpu(), x.float().t().cpu()) + refs[3][1][:128].cpu()
    assert rel(first[3][:128].float().cpu(), ref) < TOL[dtype]
    # the same four problems as single launches agree with the grouped launch (same tiles, same k order)
    for (dy, x, _), (prod, prev), f in zip(probs, refs, first):
        one = prev.to(dtype).clone()
        ops.gemm(dy, x, trans_a=True, trans_b=True, out=one, accumulate=True, variant=10)
        assert rel(one.float(), f.float()) < 2e-3


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_logits_wgrad_at_bench_scale(ops, dtype):
    """dE[V, h] += dlogits[M, V]^T x[M, h] with V = 58240 (228 row tiles, not a multiple of the 8 XCD queues)."""
    V, h = 58240, 2560
    dl, x = rnd((M_BENCH, V), dtype, 41, 0.01), rnd((M_BENCH, h), dtype, 42)
    out = torch.zeros((V, h), dtype=dtype, device="cuda")
    ops.gemm(dl, x, trans_a=True, trans_b=True, out=out, accumulate=True)
    num = den = 0.0
    for v0 in range(0, V, 7280):
        ref = dl[:, v0:v0 + 7280].float().t() @ x.float()
        d = out[v0:v0 + 7280].float() - ref
        num += float(d.double().pow(2).sum())
        den += float(ref.double().pow(2).sum())
        _tile_guard(out[v0:v0 + 7280], ref, dtype, ("logits wgrad", v0))
    assert (num / den) ** 0.5 < TOL[dtype]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("layout", ["NT", "NN"])
def test_cross_item_prefetch_equals_the_prologue_path(dtype, layout):
    """Generation-4 GEMM, round 4: the first two k-tiles of the NEXT output tile ride in the DMA slots of the current tile's
    last two k-tiles (cross-item prefetch) whenever both tiles are interior; edge tiles, the first tile of a workgroup and
    launches that do not qualify take the set-up + prologue path.  Shapes with interior AND edge tiles in both dimensions
    (17 x 17 tiles of 256, the last row / column partial) and the two forward / dgrad layouts: the result must be BIT-IDENTICAL
    with the prefetch switched off (COGV_GEMM_XP=0: same arithmetic, same order), twice in a row (work-queue counters re-arm),
    and match an fp32 matmul."""
    import os
    from cogview_amd import ops
    g = torch.Generator().manual_seed(7)
    M, N, K = 4096 + 72, 4096 + 136, 512
    a = (torch.randn(M, K, generator=g) * 0.5).to(dtype).cuda()
    w = (torch.randn(N, K, generator=g) * 0.5).to(dtype).cuda()
    bias = (torch.randn(N, generator=g) * 0.1).to(dtype).cuda()
    b_op = w if layout == "NT" else w.t().contiguous()
    run = lambda: ops.gemm(a, b_op, trans_b=(layout == "NN"), bias=bias)
    old = os.environ.get("COGV_GEMM_XP")
    try:
        os.environ["COGV_GEMM_XP"] = "0"
        ref0 = run()
        os.environ["COGV_GEMM_XP"] = "1"
        y1, y2 = run(), run()
    finally:
        if old is None:
            os.environ.pop("COGV_GEMM_XP", None)
        else:
            os.environ["COGV_GEMM_XP"] = old
    assert torch.equal(y1, ref0) and torch.equal(y2, ref0)
    ref = a.float() @ w.float().t() + bias.float()
    e = ((y1.float() - ref).norm() / ref.norm()).item()
    assert e < (1e-3 if dtype == torch.float16 else 6e-3), e
