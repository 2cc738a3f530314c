"""Thin tensor-level wrappers over the C ABI (one Python function per entry point of include/cogview_hip.h).

torch is used for device memory, streams and shapes only; every FLOP / byte of the hot path happens inside
libcogview_hip.so.  All functions require CUDA(HIP) tensors and raise otherwise -- there is no CPU path.
"""
import ctypes as C
import os

import torch

from . import _lib as L

_DT = {torch.float16: L.F16, torch.bfloat16: L.BF16, torch.float32: L.F32}


def dt_code(t):
    try:
        return _DT[t.dtype if isinstance(t, torch.Tensor) else t]
    except KeyError:
        raise L.CogviewHipError(f"unsupported dtype {t.dtype if isinstance(t, torch.Tensor) else t}")


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise L.CogviewHipError(
                "cogview_amd ops run only on an MI355X (HIP) device: got a CPU tensor and there is no CPU fallback")


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# ------------------------------------------------------------------------------------------ scratch
_WS = {}


def workspace(tag, nbytes, device, floor=1 << 20):
    """Stream-ordered scratch (one buffer per tag and device, grown geometrically; `floor`: smallest allocation -- 0 for the
    small per-stream tags)."""
    key = (tag, device.index if device.index is not None else torch.cuda.current_device())
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes * 1.25), int(floor), 256), dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf


class _ScalarPool:
    """Zero-initialised fp32 device scalars (abs-max slots for Sandwich-LN).  Slots are never recycled:
    a fresh zeroed slab (one memset) is allocated every `n` requests and kept alive by its views."""

    def __init__(self, n=2048):
        self.n, self.slab, self.i = n, {}, {}

    def next(self, device):
        key = device.index if device.index is not None else torch.cuda.current_device()
        if key not in self.slab or self.i[key] >= self.n:
            self.slab[key] = torch.zeros(self.n, dtype=torch.float32, device=device)
            self.i[key] = 0
        s = self.slab[key][self.i[key]:self.i[key] + 1]
        self.i[key] += 1
        return s


_scalars = _ScalarPool()


class scalar_slab:
    """Context manager: abs-max slots come, in order, from the caller's zeroed fp32 slab instead of the pool.  A captured
    decode step (generation/decoder.py) needs slots at FIXED addresses that its own first node clears on every replay."""

    def __init__(self, slab):
        self.slab, self.i = slab, 0

    def next(self, device):
        assert self.i < self.slab.numel(), "scalar slab exhausted"
        s = self.slab[self.i:self.i + 1]
        self.i += 1
        return s

    def __enter__(self):
        global _scalars
        self.saved, _scalars = _scalars, self
        self.i = 0
        return self

    def __exit__(self, *exc):
        global _scalars
        _scalars = self.saved
        return False


def new_absmax_slot(device):
    return _scalars.next(device)


# ------------------------------------------------------------------------------------------ GEMM
_GEMM_TIMING = None      # list of (variant, flops, start_event, end_event) while bench.py's timed region runs


def enable_gemm_timing():
    """Bracket every cogv_gemm launch with HIP events on the launch stream (bench.py's roofline leg)."""
    global _GEMM_TIMING
    _GEMM_TIMING = []
    return _GEMM_TIMING


_GEMM_TIMING_PARKED = None


def sample_gemm_timing(active):
    """Inside the timed region: bracket the launches of THIS step (active) or let it run bare.  bench.py samples one step
    in four -- 870 event records per sampled 4B step cost 0.6 % of the step when every step carries them
    (profiles/r04_bench_event_overhead_ab.log)."""
    global _GEMM_TIMING, _GEMM_TIMING_PARKED
    if active and _GEMM_TIMING is None and _GEMM_TIMING_PARKED is not None:
        _GEMM_TIMING, _GEMM_TIMING_PARKED = _GEMM_TIMING_PARKED, None
    elif not active and _GEMM_TIMING is not None:
        _GEMM_TIMING_PARKED, _GEMM_TIMING = _GEMM_TIMING, None


def collect_gemm_timing():
    """Synchronise, resolve the events and return aggregate statistics; disables timing."""
    global _GEMM_TIMING, _GEMM_TIMING_PARKED
    if _GEMM_TIMING is None:
        _GEMM_TIMING, _GEMM_TIMING_PARKED = _GEMM_TIMING_PARKED, None
    rec, _GEMM_TIMING = _GEMM_TIMING, None
    torch.cuda.synchronize()
    tot_ms, tot_fl, tot_by, by, shapes = 0.0, 0.0, 0.0, {}, {}
    for rec_i in rec:
        variant, fl, nbytes, s, e = rec_i[:5]
        ms = s.elapsed_time(e)
        tot_ms += ms
        tot_fl += fl
        tot_by += nbytes
        a = by.setdefault(variant, [0.0, 0.0, 0, 0.0])
        a[0] += fl
        a[1] += ms
        a[2] += 1
        a[3] += nbytes
        if len(rec_i) > 5:
            a = shapes.setdefault(f"{variant} {rec_i[5]}", [0.0, 0.0, 0, 0.0])
            a[0] += fl
            a[1] += ms
            a[2] += 1
            a[3] += nbytes
    n = max(len(rec), 1)

    def agg(keys):
        """totals over the kernel families `keys` (the GEMM roofline counts GEMM launches only)"""
        fl, ms, cnt, nb = (sum(by[k][i] for k in keys if k in by) for i in range(4))
        return {"launches": cnt, "total_ms": ms, "avg_ms": ms / max(cnt, 1), "algo_bytes": nb, "tflops": fl / max(ms, 1e-9) / 1e9}

    def entry(v):
        return {"tflops": v[0] / max(v[1], 1e-9) / 1e9, "launches": v[2], "avg_ms": v[1] / v[2], "total_ms": v[1],
                "algo_gbytes_per_s": v[3] / max(v[1], 1e-9) / 1e6}

    out = agg(list(by))
    out.update({"by_variant": {k: entry(v) for k, v in by.items()},
                "by_shape": {k: {"tflops": round(v[0] / max(v[1], 1e-9) / 1e9, 1), "launches": v[2], "avg_ms": round(v[1] / v[2], 4),
                                 "gbytes_per_s": round(v[3] / max(v[1], 1e-9) / 1e6, 1)}
                             for k, v in sorted(shapes.items(), key=lambda kv: -kv[1][1])},
                "gemm": agg(GEMM_FAMILIES)})
    return out


GEMM_FAMILIES = ("NT_fwd", "NN_dgrad", "TN_wgrad", "TN_other")


class timed_launch:
    """Bracket a launch of any other kernel family with HIP events on the launch stream while bench.py's timed region
    runs (same record list as the GEMMs): `with timed_launch("conv", flops, bytes, "label"): <launch>`."""
    __slots__ = ("rec",)

    def __init__(self, family, flops, nbytes, label):
        # flops / label may be callables: they are only evaluated while a timed region runs (~1500 launches per 4B step build
        # their label strings for nothing otherwise)
        self.rec = None
        if _GEMM_TIMING is not None:
            self.rec = [family, float(flops() if callable(flops) else flops), float(nbytes), torch.cuda.Event(enable_timing=True),
                        torch.cuda.Event(enable_timing=True), label() if callable(label) else label]

    def __enter__(self):
        if self.rec is not None:
            self.rec[3].record()
        return self

    def __exit__(self, *exc):
        if self.rec is not None and _GEMM_TIMING is not None:
            self.rec[4].record()
            _GEMM_TIMING.append(tuple(self.rec))
        return False


def gemm(a, b, trans_a=False, trans_b=False, out=None, bias=None, gelu=False, gelu_aux=None, dgelu_aux=None,
         dropout=None, absmax=None, accumulate=False, splitk=None, out_dtype=None, variant=0, colsum_out=None,
         colsum_accumulate=True, gelu_daux=None, mul_aux=None, dropout_row0=0):
    """C[M,N] = epilogue(A_op[M,K] . B_op[N,K]^T); a, b 2-D, last dim contiguous.
    trans_a: `a` is stored [K, M];  trans_b: `b` is stored [K, N].
    gelu_aux [M,N]: receives the rounded pre-activation; gelu_daux [M,N] (instead): receives gelu'(pre-activation), which
    the backward GEMM applies with mul_aux (out = x * mul_aux) -- one multiply where dgelu_aux re-evaluates the sigmoid.
    dropout = (p, seed, stream_id); dropout_row0: the call computes rows [row0, row0 + M) of a larger tensor whose mask it
    draws (row chunks of a row-parallel Linear).  colsum_out [N]: (+)= column sums of C (the bias gradient of the layer whose
    output gradient C is), fused into the epilogue when the shape allows, otherwise a separate pass.  Returns C."""
    _need_gpu(a, b)
    assert a.dim() == 2 and b.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1
    if trans_a:
        K, M = a.shape
    else:
        M, K = a.shape
    if trans_b:
        Kb, N = b.shape
    else:
        N, Kb = b.shape
    assert K == Kb, f"contraction mismatch {K} vs {Kb}"
    d = L.GemmDesc()
    d.dtype = dt_code(a)
    d.trans_a, d.trans_b = int(trans_a), int(trans_b)
    d.M, d.N, d.K = M, N, K
    d.A, d.lda = a.data_ptr(), a.stride(0)
    d.B, d.ldb = b.data_ptr(), b.stride(0)
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype or a.dtype, device=a.device)
        assert not accumulate
    assert out.shape == (M, N) and out.stride(1) == 1
    d.C, d.ldc = out.data_ptr(), out.stride(0)
    d.out_f32 = int(out.dtype == torch.float32)
    flags = 0
    if bias is not None:
        flags |= L.EPI_BIAS
        d.bias = bias.data_ptr()
    if gelu:
        flags |= L.EPI_GELU
        assert gelu_aux is None or gelu_daux is None
        if gelu_aux is not None:
            d.aux, d.ldaux = gelu_aux.data_ptr(), gelu_aux.stride(0)
        if gelu_daux is not None:
            flags |= L.EPI_GELU_DAUX
            d.aux, d.ldaux = gelu_daux.data_ptr(), gelu_daux.stride(0)
    if dgelu_aux is not None:
        flags |= L.EPI_DGELU
        d.aux, d.ldaux = dgelu_aux.data_ptr(), dgelu_aux.stride(0)
    if mul_aux is not None:
        assert dgelu_aux is None and not gelu
        flags |= L.EPI_MULAUX
        d.aux, d.ldaux = mul_aux.data_ptr(), mul_aux.stride(0)
    if dropout is not None and dropout[0] > 0.0:
        flags |= L.EPI_DROPOUT
        d.dropout_p, d.seed, d.stream_id = float(dropout[0]), int(dropout[1]), int(dropout[2])
        d.dropout_row0 = int(dropout_row0)
    if absmax is not None:
        flags |= L.EPI_ABSMAX
        d.absmax = absmax.data_ptr()
    if accumulate:
        flags |= L.EPI_ACCUM
    lib = L.lib()
    fuse_colsum = (colsum_out is not None and M >= 256 and N >= 256 and K % 64 == 0 and variant in (0, 9, 10)
                   and out.dtype != torch.float32 and a.stride(0) * a.shape[0] * 2 < 2 ** 32
                   and b.stride(0) * b.shape[0] * 2 < 2 ** 32 and (trans_b is False or N % 8 == 0) and not trans_a)
    if fuse_colsum:
        flags |= L.EPI_COLSUM
        cs_rows = lib.cogv_gemm_colsum_rows(M)
        cs_ws = workspace("gemm_colsum", cs_rows * N * 4, a.device)
        d.colsum_partial = cs_ws.data_ptr()
        splitk = 1
    d.flags = flags
    if splitk is None:
        splitk = lib.cogv_gemm_pick_splitk(M, N, K)
    d.splitk = int(splitk)
    d.kernel_variant = int(variant)
    if d.splitk > 1:
        nbytes = lib.cogv_gemm_workspace_bytes(C.byref(d))
        ws = workspace("gemm_splitk", nbytes, a.device)
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
    if _GEMM_TIMING is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        L.check(lib.cogv_gemm(C.byref(d), _stream()), "cogv_gemm")
        ev1.record()
        variant = ("T" if trans_a else "N") + ("T" if trans_b else "N")
        _GEMM_TIMING.append(({"NN": "NT_fwd", "NT": "NN_dgrad", "TT": "TN_wgrad", "TN": "TN_other"}[variant],
                             2.0 * M * N * K, 2.0 * (M * K + N * K + M * N), ev0, ev1, f"{M}x{N}x{K} epi={flags}"))
    else:
        L.check(lib.cogv_gemm(C.byref(d), _stream()), "cogv_gemm")
    if colsum_out is not None:
        if fuse_colsum:
            L.check(lib.cogv_colsum_finalize(dt_code(out), d.colsum_partial, cs_rows, N, _p(colsum_out), int(colsum_accumulate),
                                             _stream()), "cogv_colsum_finalize")
        else:
            colsum(out, out=colsum_out, accumulate=colsum_accumulate)
    return out


def gemm_reserve_cus(n):
    """The persistent GEMM launches leave `n` CUs free from now on (cogv_gemm_reserve_cus; room for a concurrent collective's
    channel workgroups); returns the previous setting."""
    return int(L.lib().cogv_gemm_reserve_cus(int(n)))


def gemv_ln(z, w, bias, gamma, beta, eps, z_absmax=None, post=None, residual=None, want_t=False, gelu=False, absmax=None):
    """Decode-step GEMV with its LayerNorms as prologue (cogv_gemv_ln): z [M, K] (M <= 8), w [N, K].
        t = residual + SandwichLN(z; post = (gamma_p, beta_p), scale z_absmax)   (post given)   else   t = z
        out = epilogue(SandwichLN(t; gamma, beta) . w^T + bias)
    Returns (out [M, N], t [M, K] or None)."""
    _need_gpu(z, w)
    assert z.dim() == 2 and w.dim() == 2 and z.is_contiguous() and w.stride(1) == 1 and z.shape[1] == w.shape[1]
    M, K = z.shape
    N = w.shape[0]
    # fp32 residual stream (see sandwich_ln_fwd): the residual of the post-LN form, or z itself in the plain-input form
    stream32 = (residual.dtype == torch.float32) if post is not None else (z.dtype == torch.float32)
    if post is not None:
        assert z.dtype == w.dtype
    out = torch.empty((M, N), dtype=w.dtype, device=z.device)
    d = L.GemmDesc()
    d.dtype = dt_code(w)
    d.M, d.N, d.K = M, N, K
    d.A, d.lda = z.data_ptr(), K
    d.B, d.ldb = w.data_ptr(), w.stride(0)
    d.C, d.ldc = out.data_ptr(), N
    flags = 0
    if bias is not None:
        flags |= L.EPI_BIAS
        d.bias = bias.data_ptr()
    if gelu:
        flags |= L.EPI_GELU
    if absmax is not None:
        flags |= L.EPI_ABSMAX
        d.absmax = absmax.data_ptr()
    d.flags, d.splitk = flags, 1
    ln = L.LnPrologue()
    ln.z = z.data_ptr()
    ln.z_absmax = None if z_absmax is None else z_absmax.data_ptr()
    t = None
    if post is not None:
        assert residual is not None and residual.is_contiguous() and residual.shape == z.shape     # z_absmax None: taken in the kernel
        ln.gamma_post, ln.beta_post, ln.residual = post[0].data_ptr(), post[1].data_ptr(), residual.data_ptr()
        if want_t:
            t = torch.empty_like(residual)
            ln.t_out = t.data_ptr()
    ln.gamma, ln.beta, ln.eps = gamma.data_ptr(), beta.data_ptr(), float(eps)
    ln.stream_f32 = int(stream32)
    L.check(L.lib().cogv_gemv_ln(C.byref(d), C.byref(ln), _stream()), "cogv_gemv_ln")
    return out, t


def gemm_grouped(problems, trans_a=True, trans_b=True, accumulate=True):
    """Up to 16 GEMMs of one layout in ONE persistent launch (cogv_gemm_grouped): `problems` is a list of
    (a, b, out) or (a, b, out, accumulate) -- the flag is per problem, `accumulate` its default.  Used for the weight gradients of one or several transformer layers, which fill the 256 CUs together.
    Falls back to one cogv_gemm per problem when the library reports a shape the grouped kernel does not take."""
    assert 1 <= len(problems) <= 16
    lib = L.lib()
    descs = (L.GemmDesc * len(problems))()
    tiles, kmin, flops, nbytes = 0, None, 0.0, 0.0
    shapes = []
    problems = [(pr[0], pr[1], pr[2], pr[3] if len(pr) > 3 else accumulate) for pr in problems]
    for d, (a, b, out, acc) in zip(descs, problems):
        _need_gpu(a, b, out)
        assert a.dim() == 2 and b.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1 and out.stride(1) == 1
        K, M = a.shape if trans_a else a.shape[::-1]
        Kb, N = b.shape if trans_b else b.shape[::-1]
        assert K == Kb and out.shape == (M, N)
        d.dtype = dt_code(a)
        d.trans_a, d.trans_b = int(trans_a), int(trans_b)
        d.M, d.N, d.K = M, N, K
        d.A, d.lda = a.data_ptr(), a.stride(0)
        d.B, d.ldb = b.data_ptr(), b.stride(0)
        d.C, d.ldc = out.data_ptr(), out.stride(0)
        d.out_f32 = int(out.dtype == torch.float32)
        d.flags = L.EPI_ACCUM if acc else 0
        shapes.append((M, N, K))
        tiles += ((M + 255) // 256) * ((N + 255) // 256)
        kmin = K if kmin is None else min(kmin, K)
        flops += 2.0 * M * N * K
        nbytes += 2.0 * (M * K + N * K + M * N)
    ok = all(M >= 256 and N >= 256 and K % 64 == 0 for M, N, K in shapes)
    if not ok:
        for a, b, out, acc in problems:
            gemm(a, b, trans_a=trans_a, trans_b=trans_b, out=out, accumulate=acc)
        return
    splitk = lib.cogv_gemm_pick_splitk_tiles(tiles, kmin)
    if splitk > 1:
        sizes = [splitk * M * N * 4 for M, N, _ in shapes]
        ws = workspace("gemm_grouped_splitk", sum(sizes), problems[0][0].device)
        off = 0
        for d, sz in zip(descs, sizes):
            d.splitk, d.workspace, d.workspace_bytes = splitk, ws.data_ptr() + off, sz
            off += sz
    else:
        for d in descs:
            d.splitk = 1
    if _GEMM_TIMING is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        L.check(lib.cogv_gemm_grouped(descs, len(problems), _stream()), "cogv_gemm_grouped")
        ev1.record()
        key = {(False, False): "NT_fwd", (False, True): "NN_dgrad", (True, True): "TN_wgrad"}.get((trans_a, trans_b), "TN_other")
        _GEMM_TIMING.append((key, flops, nbytes, ev0, ev1, f"grouped, {tiles} tiles of 256x256 ({tiles / 256:.2f} rounds), K={kmin}"))
    else:
        L.check(lib.cogv_gemm_grouped(descs, len(problems), _stream()), "cogv_gemm_grouped")


# ------------------------------------------------------------------------------------------ Sandwich-LN
LN_ALL_T, LN_STREAM_IN, LN_STREAM_OUT = 0, 1, 2       # cogview_hip.h COGV_LN_*
_LN_MODE_NAME = {0: "16-bit", 1: "stream in (fp32 -> 16-bit)", 2: "stream out (16-bit + fp32 -> fp32)"}


def sandwich_ln_fwd(x, gamma, beta, eps, absmax_in, residual=None, absmax_out=None, save_stats=True):
    """y = [residual +] SandwichLN(x).  The fp32 RESIDUAL STREAM is recognised by dtype: an fp32 `x` (with 16-bit
    gamma) is the stream feeding a branch -> y in gamma's type (LN1, LN2, final LN); an fp32 `residual` is the stream
    a branch output joins -> y fp32 = residual + LN(x), no rounding in between (LN3, LN4)."""
    _need_gpu(x, gamma, beta)
    h = x.shape[-1]
    x2 = x.reshape(-1, h)
    assert x2.is_contiguous()
    rows = x2.shape[0]
    mode, ydt = LN_ALL_T, gamma.dtype
    r2 = None
    if residual is not None:
        r2 = residual.reshape(-1, h)
        assert r2.is_contiguous()
        if residual.dtype == torch.float32:
            mode, ydt = LN_STREAM_OUT, torch.float32
        elif residual.dtype != gamma.dtype:
            raise L.CogviewHipError("Sandwich-LN residual must be fp32 (the residual stream) or the storage type")
    if x.dtype == torch.float32:
        if mode == LN_STREAM_OUT:
            raise L.CogviewHipError("Sandwich-LN: an fp32 input with an fp32 residual is not a form the layer uses")
        mode = LN_STREAM_IN
    elif x.dtype != gamma.dtype:
        raise L.CogviewHipError(f"Sandwich-LN input {x.dtype} vs parameters {gamma.dtype}")
    y = torch.empty((rows, h), dtype=ydt, device=x.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device) if save_stats else None
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if save_stats else None
    # algorithmic HBM bytes: x + y (+ residual), each in its own width
    nbytes = rows * h * (x2.element_size() + y.element_size() + (r2.element_size() if r2 is not None else 0))
    with timed_launch("layernorm", 0.0, nbytes, lambda: "ln_fwd " + _LN_MODE_NAME[mode] + (" + residual" if r2 is not None and mode == LN_ALL_T else "")):
        L.check(L.lib().cogv_sandwich_ln_fwd(dt_code(gamma), _p(x2), _p(gamma), _p(beta), _p(r2), _p(y), _p(mean), _p(rstd),
                                             _p(absmax_in), _p(absmax_out), rows, h, float(eps), mode, _stream()),
                "cogv_sandwich_ln_fwd")
    return y.view(x.shape), mean, rstd


def sandwich_ln_bwd(dy, x, gamma, mean, rstd, add_in=None, dropout=None, dgamma=None, dbeta=None, colsum=None,
                    accumulate=False, marked=False):
    """dx = [add_in +] mask(LN'(dy)).  dgamma/dbeta/colsum: preallocated [h] tensors (or None).
    marked: x is the output of a gemm(..., dropout=...) -- its dropped elements are -0.0 and no kept element is -0.0, so
    the mask is read from x (cogv_sandwich_ln_bwd_marked) instead of being regenerated from (seed, stream); `dropout`
    still supplies p.  Bit-identical to the regenerating form on such an x.
    Stream forms by dtype, mirroring sandwich_ln_fwd: fp32 `x` (LN1, LN2: the saved stream) -> dx fp32, add_in fp32;
    fp32 `dy` (LN3, LN4: the stream's gradient) with a 16-bit x -> dx 16-bit."""
    _need_gpu(dy, x)
    h = x.shape[-1]
    dy2, x2 = dy.reshape(-1, h), x.reshape(-1, h)
    assert dy2.is_contiguous() and x2.is_contiguous()
    rows = x2.shape[0]
    mode = LN_ALL_T
    if x.dtype == torch.float32:
        mode = LN_STREAM_IN
        if dy.dtype != gamma.dtype or (add_in is not None and add_in.dtype != torch.float32):
            raise L.CogviewHipError("Sandwich-LN backward (stream input): dy must be 16-bit and add_in fp32")
    elif dy.dtype == torch.float32:
        mode = LN_STREAM_OUT
        if add_in is not None and add_in.dtype != x.dtype:
            raise L.CogviewHipError("Sandwich-LN backward (stream output): add_in must have x's type")
    elif dy.dtype != x.dtype or (add_in is not None and add_in.dtype != x.dtype):
        raise L.CogviewHipError("Sandwich-LN backward: mixed 16-bit types")
    dx = torch.empty_like(x2)
    a2 = None
    if add_in is not None:
        a2 = add_in.reshape(-1, h)
        assert a2.is_contiguous()
    lib = L.lib()
    nbytes = lib.cogv_ln_bwd_workspace_bytes(rows, h)
    ws = workspace("ln_bwd", nbytes, x.device)
    p, seed, sid = (0.0, 0, 0) if dropout is None else dropout
    nb = rows * h * (dy2.element_size() + x2.element_size() + dx.element_size() + (a2.element_size() if a2 is not None else 0))
    marked = bool(marked) and p > 0.0
    if marked and mode == LN_STREAM_IN:
        raise L.CogviewHipError("Sandwich-LN backward: marked zeros need a 16-bit x")
    label = lambda: "ln_bwd " + _LN_MODE_NAME[mode] + ((" + dropout from marked zeros" if marked else " + dropout replay") if p > 0.0 else "") + \
        (" + add" if a2 is not None else "")
    with timed_launch("layernorm", 0.0, nb, label):
        if marked:
            L.check(lib.cogv_sandwich_ln_bwd_marked(dt_code(gamma), _p(dy2), _p(x2), _p(gamma), _p(mean), _p(rstd), _p(a2), _p(dx),
                                                    _p(dgamma), _p(dbeta), _p(colsum), int(accumulate), rows, h, float(p),
                                                    _p(ws), ws.numel(), mode, _stream()), "cogv_sandwich_ln_bwd_marked")
            return dx.view(x.shape)
        L.check(lib.cogv_sandwich_ln_bwd(dt_code(gamma), _p(dy2), _p(x2), _p(gamma), _p(mean), _p(rstd), _p(a2), _p(dx),
                                         _p(dgamma), _p(dbeta), _p(colsum), int(accumulate), rows, h, float(p), int(seed),
                                         int(sid), _p(ws), ws.numel(), mode, _stream()), "cogv_sandwich_ln_bwd")
    return dx.view(x.shape)


def ln_bwd_pair_supported(h):
    """cogv_sandwich_ln_bwd_pair takes rows of at least four waves of 8-column lanes (h >= 1537 .. 4096)."""
    return 1536 < h <= 4096


def sandwich_ln_bwd_pair(dc, y, gamma2, mean2, rstd2, dout, ao, gamma3, mean3, rstd3, dropout_p=0.0, dgamma2=None, dbeta2=None,
                         dgamma3=None, dbeta3=None, colsum=None, accumulate=False):
    """LN2' and LN3' of a layer in one pass (cogv_sandwich_ln_bwd_pair): dy = dout + LN2'(dc; y) (fp32), d_ao = mask(LN3'(dy; ao))
    (16-bit; dropout_p > 0: ao carries the producing GEMM's marked zeros).  Returns (dy, d_ao); both bit-identical to
    sandwich_ln_bwd(dc, y, ..., add_in=dout) followed by sandwich_ln_bwd(dy, ao, ..., marked=True)."""
    _need_gpu(dc, y, ao)
    h = y.shape[-1]
    dc2, y2, do2, ao2 = dc.reshape(-1, h), y.reshape(-1, h), dout.reshape(-1, h), ao.reshape(-1, h)
    assert dc2.is_contiguous() and y2.is_contiguous() and do2.is_contiguous() and ao2.is_contiguous()
    if y.dtype != torch.float32 or dout.dtype != torch.float32 or dc.dtype != gamma2.dtype or ao.dtype != gamma2.dtype:
        raise L.CogviewHipError("Sandwich-LN backward pair: y / dout must be the fp32 stream, dc / ao the 16-bit storage type")
    rows = y2.shape[0]
    dy = torch.empty_like(y2)
    d_ao = torch.empty_like(ao2)
    lib = L.lib()
    ws = workspace("ln_bwd_pair", lib.cogv_ln_bwd_pair_workspace_bytes(rows, h), y.device)
    nb = rows * h * (2 + 4 + 4 + 4 + 2 + 2)
    with timed_launch("layernorm", 0.0, nb, lambda: "ln_bwd pair: stream in + add, then stream out" + (" + dropout from marked zeros" if dropout_p > 0.0 else "")):
        L.check(lib.cogv_sandwich_ln_bwd_pair(dt_code(gamma2), _p(dc2), _p(y2), _p(gamma2), _p(mean2), _p(rstd2), _p(do2), _p(dy),
                                              _p(dgamma2), _p(dbeta2), _p(ao2), _p(gamma3), _p(mean3), _p(rstd3), _p(d_ao),
                                              _p(dgamma3), _p(dbeta3), _p(colsum), int(accumulate), rows, h, float(dropout_p),
                                              _p(ws), ws.numel(), _stream()), "cogv_sandwich_ln_bwd_pair")
    return dy.view(y.shape), d_ao.view(ao.shape)


# ------------------------------------------------------------------------------------------ attention
def attention_executed_flops(b, H, s_q, s_k, sep=0, dense=True):
    """FLOPs the forward attention kernel EXECUTES: 64 x 64 score blocks that hold at least one visible key (left-to-right rule
    with the fully visible prefix `sep`; a masked block is skipped, a diagonal block computed whole), 2 products of
    2 * 64 * 64 * 64 FLOPs each.  s = 1088: 153 of 289 blocks ("causal-discounted").  Forms that visit
    every block (arbitrary mask tensors, gathered / sparse keys) count all of them."""
    nq, nk = (s_q + 63) // 64, (s_k + 63) // 64
    if dense:
        off = s_k - s_q
        sep_i = int(sep) if isinstance(sep, int) else 0
        blocks = 0
        for i in range(nq):
            last_key = max(min(s_q, 64 * i + 64) - 1 + off, sep_i - 1)        # last visible key of the block's last query
            blocks += max(0, min(nk, last_key // 64 + 1))
    else:
        blocks = nq * nk
    return float(b) * H * blocks * 4.0 * 64 * 64 * 64


def _attn_strides(t):
    # t: [b, s, heads, 64] view with d contiguous and heads packed (stride 64)
    assert t.dim() == 4 and t.shape[-1] == 64 and t.stride(3) == 1 and (t.stride(2) == 64 or t.shape[2] == 1)
    return t.stride(0), t.stride(1)


def _attn_desc(q, k, v, o, sep, dropout):
    d = L.AttnDesc()
    d.dtype = dt_code(q)
    d.B, d.s_q, d.H = q.shape[0], q.shape[1], q.shape[2]
    d.s_k = k.shape[1]
    d.head_dim = 64
    d.sep = int(sep)
    d.scale = 0.125
    p, seed, sid = (0.0, 0, 0) if dropout is None else dropout
    d.dropout_p, d.seed, d.stream_id = float(p), int(seed), int(sid)
    d.q, d.k, d.v, d.o = q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr()
    d.q_bs, d.q_rs = _attn_strides(q)
    d.k_bs, d.k_rs = _attn_strides(k)
    d.v_bs, d.v_rs = _attn_strides(v)
    d.o_bs, d.o_rs = _attn_strides(o)
    return d


STORE_KEEP_BITS = os.environ.get("COGV_ATTN_KEEP_BITS", "1") != "0"       # A/B switch of the stored dropout keep bits


def _attn_mask(d, mask, q, k):
    """mask: [B or 1, s_q, s_k] contiguous, q's dtype -- an arbitrary mask tensor (cogv_attn_desc.mask)."""
    b, s_q, s_k = q.shape[0], q.shape[1], k.shape[1]
    assert mask.dtype == q.dtype and mask.is_contiguous() and mask.dim() == 3 and mask.shape[1:] == (s_q, s_k)
    assert mask.shape[0] in (1, b)
    d.mask, d.mask_bs = mask.data_ptr(), (s_q * s_k if mask.shape[0] == b and b > 1 else 0)


def attention_fwd(q, k, v, sep=0, dropout=None, kv_index=None, sparse=None, keep_bits=False, mask=None):
    """q [b,s_q,H,64], k/v [b,s_k,H,64] (strided views are fine).  Returns (o [b,s_q,H,64] contiguous, lse).
    keep_bits=True (dense form with dropout; the caller will run attention_bwd): returns (o, lse, bits) -- the forward
    kernel stores its dropout keep decisions (1 bit per score, uint8 buffer of cogv_attention_keep_bits_bytes) and
    attention_bwd(keep_bits=bits) reads them instead of regenerating the draws; bits is None where that does not apply.
    kv_index [b, n] int32 (forward only): key slot j is row kv_index[b, j] of k / v -- the gathered form of
    sparse_attention_inference; the left-to-right rule then applies to slots (the last s_q slots are the queries)."""
    _need_gpu(q, k, v)
    b, s_q, H, _ = q.shape
    o = torch.empty((b, s_q, H, 64), dtype=q.dtype, device=q.device)
    lse = torch.empty((b, H, s_q), dtype=torch.float32, device=q.device)
    d = _attn_desc(q, k, v, o, sep, dropout)
    if kv_index is not None and sparse is None:
        assert kv_index.dtype == torch.int32 and kv_index.dim() == 2 and kv_index.shape[0] == b and kv_index.is_contiguous()
        assert kv_index.shape[1] >= s_q
        d.s_k = kv_index.shape[1]
        d.kv_index, d.kv_index_bs = kv_index.data_ptr(), kv_index.stride(0)
    elif sparse is not None:
        _sparse_desc(d, kv_index, sparse, b, s_q)
    d.lse = lse.data_ptr()
    bits = None
    if mask is not None:
        assert kv_index is None and sparse is None
        _attn_mask(d, mask, q, k)
    if keep_bits and STORE_KEEP_BITS and dropout is not None and dropout[0] > 0.0 and kv_index is None and sparse is None and mask is None:
        bits = torch.empty(L.lib().cogv_attention_keep_bits_bytes(b, H, s_q, k.shape[1]), dtype=torch.uint8, device=q.device)
        d.keep_bits = bits.data_ptr()
    with timed_launch("attention", lambda: attention_executed_flops(b, H, s_q, d.s_k, sep, dense=(kv_index is None and sparse is None and mask is None)),
                      0.0, lambda: f"attn_fwd {b}x{H}x{s_q}x{d.s_k}" + (" dropout" if dropout is not None else "")):
        L.check(L.lib().cogv_attention_fwd(C.byref(d), _stream()), "cogv_attention_fwd")
    return (o, lse, bits) if keep_bits else (o, lse)


_DECODE_WS = {}


def attention_decode(qkv, cache, pos_index, heads, combine=True):
    """combine=False: only the key-split kernel runs; returns the partials workspace for gemv_attn (the attention-output
    projection recombines the splits in its prologue: one launch less).
    One decode step's attention (cogv_attention_decode): qkv [b, 1, 3 * heads * 64] (q | k | v of the new token),
    cache [b, capacity, 2 * heads * 64] (keys | values), pos_index: device int64 scalar = slot of the new token (slots
    [0, pos] are attended; the new key / value are written into slot pos by the kernel).  Returns out [b, 1, heads * 64].
    The workspace (partial results of the key splits) is kept per (device, shape): fixed addresses, so the call is
    replayable inside a captured graph."""
    _need_gpu(qkv, cache, pos_index)
    b, cap = cache.shape[0], cache.shape[1]
    hp = heads * 64
    assert qkv.shape[0] == b and qkv.shape[-1] == 3 * hp and qkv.numel() == b * 3 * hp and cache.shape[2] == 2 * hp
    assert qkv.stride(-1) == 1 and cache.stride(2) == 1 and pos_index.dtype == torch.int64
    lib = L.lib()
    key = (qkv.device.index, b, heads, cap)
    ws = _DECODE_WS.get(key)
    if ws is None:
        ws = _DECODE_WS[key] = torch.zeros(lib.cogv_attention_decode_workspace_bytes(b, heads, cap), dtype=torch.uint8, device=qkv.device)
    out = torch.empty((b, 1, hp), dtype=qkv.dtype, device=qkv.device) if combine else None
    d = L.AttnDecodeDesc()
    d.dtype, d.B, d.H, d.capacity, d.head_dim, d.scale = dt_code(qkv), b, heads, cap, 64, 0.125
    d.qkv, d.qkv_bs = qkv.data_ptr(), qkv.stride(0)
    d.cache, d.cache_bs, d.cache_rs = cache.data_ptr(), cache.stride(0), cache.stride(1)
    d.out, d.out_bs = (out.data_ptr(), out.stride(0)) if combine else (None, hp)
    d.pos = pos_index.data_ptr()
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
    d.skip_combine = 0 if combine else 1
    L.check(lib.cogv_attention_decode(C.byref(d), _stream()), "cogv_attention_decode")
    return out if combine else ws


def gemv_attn(partials, batch, heads, capacity, w, bias=None, absmax=None):
    """Attention-output projection of a decode step with the split-combine as its prologue (cogv_gemv_attn):
    out [batch, N] = (combined attention output [batch, heads * 64]) . w^T + bias.  `partials`: what
    attention_decode(..., combine=False) returned for the same (batch, heads, capacity)."""
    _need_gpu(partials, w)
    N, K = w.shape
    assert K == heads * 64 and w.stride(1) == 1 and batch <= 8
    out = torch.empty((batch, N), dtype=w.dtype, device=w.device)
    d = L.GemmDesc()
    d.dtype = dt_code(w)
    d.M, d.N, d.K = batch, N, K
    d.A, d.lda = None, K
    d.B, d.ldb = w.data_ptr(), w.stride(0)
    d.C, d.ldc = out.data_ptr(), N
    flags = 0
    if bias is not None:
        flags |= L.EPI_BIAS
        d.bias = bias.data_ptr()
    if absmax is not None:
        flags |= L.EPI_ABSMAX
        d.absmax = absmax.data_ptr()
    d.flags, d.splitk = flags, 1
    L.check(L.lib().cogv_gemv_attn(C.byref(d), _p(partials), int(heads), int(capacity), _stream()), "cogv_gemv_attn")
    return out


def _sparse_desc(d, kv_index, sparse, b, s_q):
    """Sparse training form in slot space: kv_index [b, s_q // w, n_slots] int32 (bit 31 = masked slot),
    sparse = (w, n_pivots, pivot_bias)."""
    w, n_piv, bias = sparse
    assert kv_index.dtype == torch.int32 and kv_index.dim() == 3 and kv_index.is_contiguous()
    assert kv_index.shape[0] == b and kv_index.shape[1] == s_q // w and s_q % w == 0 and w % 128 == 0
    d.s_k = kv_index.shape[2]
    d.kv_index, d.kv_index_bs, d.kv_index_gs = kv_index.data_ptr(), kv_index.stride(0), kv_index.stride(1)
    d.sparse_window, d.sparse_pivots, d.sparse_pivot_bias = int(w), int(n_piv), float(bias)


def sparse_attention_bwd(dout, q, k, v, o, lse, kv_index, sparse, pivot_inv, times, dropout=None):
    """Backward of the sparse training form (attention_fwd with sparse=...).  The dK/dV kernel works per (batch, query
    block) on the block's slots and leaves slot-space gradients; cogv_sparse_slot_reduce folds them onto the keys.
    pivot_inv [b, s] int32: pivot slot of each key or -1."""
    _need_gpu(dout, q, k, v, o)
    b, s, H, _ = q.shape
    w, n_piv, _bias = sparse
    G, n_slots = s // w, kv_index.shape[2]
    assert n_slots == n_piv + times * w and pivot_inv.dtype == torch.int32 and pivot_inv.shape == (b, s) and pivot_inv.is_contiguous()
    dq = torch.empty((b, s, H, 64), dtype=q.dtype, device=q.device)
    dks = torch.empty((b * G, n_slots, H, 64), dtype=q.dtype, device=q.device)
    dvs = torch.empty_like(dks)
    dvec = torch.empty((2, b, H, s), dtype=torch.float32, device=q.device)
    d = _attn_desc(q, k, v, o, 0, dropout)
    _sparse_desc(d, kv_index, sparse, b, s)
    d.lse, d.dvec = lse.data_ptr(), dvec.data_ptr()
    d.dout, d.dq, d.dk, d.dv = dout.data_ptr(), dq.data_ptr(), dks.data_ptr(), dvs.data_ptr()
    d.do_bs, d.do_rs = _attn_strides(dout)
    d.dq_bs, d.dq_rs = _attn_strides(dq)
    d.dk_bs, d.dk_rs = _attn_strides(dks)
    d.dv_bs, d.dv_rs = _attn_strides(dvs)
    L.check(L.lib().cogv_attention_bwd(C.byref(d), _stream()), "cogv_attention_bwd")
    dk = torch.empty((b, s, H, 64), dtype=q.dtype, device=q.device)
    dv = torch.empty_like(dk)
    L.check(L.lib().cogv_sparse_slot_reduce(dt_code(q), _p(dks), _p(dvs), _p(pivot_inv), _p(dk), _p(dv),
                                            dk.stride(0), dk.stride(1), dv.stride(0), dv.stride(1), b, s, H, int(w),
                                            int(times), int(n_piv), _stream()), "cogv_sparse_slot_reduce")
    return dq, dk, dv


def attention_bwd(dout, q, k, v, o, lse, sep=0, dropout=None, dq=None, dk=None, dv=None, colsum_out=None,
                  colsum_accumulate=True, keep_bits=None, mask=None):
    """colsum_out [3*H*64]: (+)= column sums of (dq | dk | dv) over all tokens -- the bias gradient of the fused
    QKV projection -- taken from the kernels' accumulators instead of re-reading the three outputs."""
    _need_gpu(dout, q, k, v, o)
    b, s_q, H, _ = q.shape
    if dq is None:
        dq = torch.empty((b, s_q, H, 64), dtype=q.dtype, device=q.device)
    if dk is None:
        dk = torch.empty((b, k.shape[1], H, 64), dtype=q.dtype, device=q.device)
    if dv is None:
        dv = torch.empty((b, k.shape[1], H, 64), dtype=q.dtype, device=q.device)
    dvec = torch.empty((2, b, H, s_q), dtype=torch.float32, device=q.device)
    d = _attn_desc(q, k, v, o, sep, dropout)
    d.lse, d.dvec = lse.data_ptr(), dvec.data_ptr()
    if mask is not None:
        _attn_mask(d, mask, q, k)
    if keep_bits is not None:
        assert mask is None and keep_bits.numel() == L.lib().cogv_attention_keep_bits_bytes(b, H, s_q, k.shape[1])
        d.keep_bits = keep_bits.data_ptr()
    d.dout, d.dq, d.dk, d.dv = dout.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr()
    d.do_bs, d.do_rs = _attn_strides(dout)
    d.dq_bs, d.dq_rs = _attn_strides(dq)
    d.dk_bs, d.dk_rs = _attn_strides(dk)
    d.dv_bs, d.dv_rs = _attn_strides(dv)
    fuse = colsum_out is not None and k.shape[1] == s_q
    if fuse:
        rows = b * ((s_q + 127) // 128)
        ws = workspace("attn_colsum", rows * 3 * H * 64 * 4, q.device)
        d.colsum_partial = ws.data_ptr()
    # 5 block products against the forward's 2
    with timed_launch("attention", lambda: 2.5 * attention_executed_flops(b, H, s_q, k.shape[1], sep, dense=mask is None), 0.0,
                      lambda: f"attn_bwd {b}x{H}x{s_q}x{k.shape[1]} (D, dQ, dK.dV)" + (" dropout" if dropout is not None else "")):
        L.check(L.lib().cogv_attention_bwd(C.byref(d), _stream()), "cogv_attention_bwd")
    if fuse:
        L.check(L.lib().cogv_colsum_finalize(dt_code(q), d.colsum_partial, rows, 3 * H * 64, _p(colsum_out),
                                             int(colsum_accumulate), _stream()), "cogv_colsum_finalize")
    elif colsum_out is not None:
        for i, t in enumerate((dq, dk, dv)):
            colsum(t.reshape(-1, H * 64), out=colsum_out[i * H * 64:(i + 1) * H * 64], accumulate=colsum_accumulate)
    return dq, dk, dv


# ------------------------------------------------------------------------------------------ embedding
def embedding_fwd(ids, table, vocab_start, pos_ids=None, pos_table=None, dropout=None, absmax_out=None, x_in=None,
                  out_f32=False):
    """out_f32: the result is the transformer's fp32 residual stream (word + position summed in fp32)."""
    _need_gpu(table if table is not None else x_in)
    src = table if table is not None else x_in
    h = src.shape[-1]
    if ids is not None:
        ids_c = ids.contiguous()
        n_tok, shape = ids_c.numel(), tuple(ids.shape) + (h,)
    else:
        ids_c = None
        x_in = x_in.contiguous()
        n_tok, shape = x_in.numel() // h, tuple(x_in.shape)
    pos_c = None
    if pos_table is not None:
        pos_c = pos_ids.expand(shape[:-1]).contiguous()
    out = torch.empty(shape, dtype=torch.float32 if out_f32 else src.dtype, device=src.device)
    p, seed, sid = (0.0, 0, 0) if dropout is None else dropout
    vend = vocab_start + (table.shape[0] if table is not None else 0)
    L.check(L.lib().cogv_embedding_fwd(dt_code(src), _p(ids_c), _p(table), int(vocab_start), int(vend), _p(x_in),
                                       _p(pos_c), _p(pos_table), 0 if pos_table is None else pos_table.shape[0],
                                       _p(out), _p(absmax_out), n_tok, h, float(p), int(seed), int(sid), int(out_f32),
                                       _stream()), "cogv_embedding_fwd")
    return out


def embedding_bwd(dout, ids, dtable, vocab_start, pos_ids=None, dpos=None, dropout=None, dx=None):
    """dtable[id - vocab_start] += sum of the (masked) gradient rows of the tokens carrying id; dpos likewise.  fp32
    sums in ascending token order, one rounding, no floating-point atomics: deterministic.  dout (and dx) may be fp32
    (the residual stream's gradient) or the tables' 16-bit type."""
    _need_gpu(dout)
    h = dout.shape[-1]
    dout_c = dout.contiguous()
    n_tok = dout_c.numel() // h
    ids_c = None if ids is None else ids.contiguous()
    pos_c = None if pos_ids is None else pos_ids.expand(dout.shape[:-1]).contiguous()
    p, seed, sid = (0.0, 0, 0) if dropout is None else dropout
    vend = vocab_start + (dtable.shape[0] if dtable is not None else 0)
    tab = dtable if dtable is not None else dpos
    d32 = dout.dtype == torch.float32
    assert dx is None or dx.dtype == dout.dtype
    lib = L.lib()
    nbytes = lib.cogv_embedding_bwd_workspace_bytes(0 if dtable is None else dtable.shape[0], 0 if dpos is None else dpos.shape[0])
    ws = workspace("embedding_bwd", nbytes, dout.device) if nbytes else None
    L.check(lib.cogv_embedding_bwd(dt_code(tab if tab is not None else dout), _p(dout_c), _p(ids_c), _p(dtable),
                                   int(vocab_start), int(vend), _p(pos_c), _p(dpos), 0 if dpos is None else dpos.shape[0],
                                   _p(dx), n_tok, h, float(p), int(seed), int(sid), _p(ws), 0 if ws is None else ws.numel(),
                                   int(d32), _stream()), "cogv_embedding_bwd")


# ------------------------------------------------------------------------------------------ element-wise
def _flat(t):
    assert t.is_contiguous() and t.numel() % 8 == 0, "element-wise ops need contiguous tensors with numel % 8 == 0"
    return t


def gelu_fwd(x):
    _need_gpu(x)
    y = torch.empty_like(_flat(x))
    L.check(L.lib().cogv_gelu_fwd(dt_code(x), _p(x), _p(y), x.numel(), _stream()), "cogv_gelu_fwd")
    return y


def gelu_bwd(dy, x):
    _need_gpu(dy, x)
    dx = torch.empty_like(_flat(x))
    L.check(L.lib().cogv_gelu_bwd(dt_code(x), _p(_flat(dy)), _p(x), _p(dx), x.numel(), _stream()), "cogv_gelu_bwd")
    return dx


def dropout(x, p, seed, stream_id, absmax_out=None):
    _need_gpu(x)
    y = torch.empty_like(_flat(x))
    L.check(L.lib().cogv_dropout(dt_code(x), _p(x), _p(y), x.numel(), float(p), int(seed), int(stream_id),
                                 _p(absmax_out), _stream()), "cogv_dropout")
    return y


def add(a, b, absmax_out=None):
    """a + b.  An fp32 `a` is the residual stream joined by the 16-bit branch output `b` (fp32 result)."""
    _need_gpu(a, b)
    if b.dtype == torch.float32 and a.dtype != torch.float32:
        raise L.CogviewHipError("ops.add: the fp32 operand (the residual stream) must come first")
    assert a.numel() == b.numel()
    if a.dtype == torch.float32 and b.dtype != torch.float32:
        out = torch.empty_like(_flat(a))
        L.check(L.lib().cogv_add_stream(dt_code(b), _p(a), _p(_flat(b)), _p(out), a.numel(), _p(absmax_out), _stream()),
                "cogv_add_stream")
        return out
    out = torch.empty_like(_flat(a))
    L.check(L.lib().cogv_add(dt_code(a), _p(a), _p(_flat(b)), _p(out), a.numel(), _p(absmax_out), _stream()), "cogv_add")
    return out


def scale(x, s):
    _need_gpu(x)
    y = torch.empty_like(_flat(x))
    L.check(L.lib().cogv_scale(dt_code(x), _p(x), _p(y), x.numel(), float(s), _stream()), "cogv_scale")
    return y


def absmax(x, out=None):
    """atomicMax of |x| into `out` (a zeroed fp32 scalar; allocated when None)."""
    _need_gpu(x)
    assert x.is_contiguous()
    if out is None:
        out = new_absmax_slot(x.device)
    L.check(L.lib().cogv_absmax(dt_code(x), _p(x), x.numel(), _p(out), _stream()), "cogv_absmax")
    return out


def colsum(dy, out=None, accumulate=False):
    _need_gpu(dy)
    assert dy.dim() == 2 and dy.stride(1) == 1
    M, N = dy.shape
    if out is None:
        out = torch.empty(N, dtype=dy.dtype, device=dy.device)
        assert not accumulate
    lib = L.lib()
    ws = workspace("colsum", lib.cogv_colsum_workspace_bytes(M, N), dy.device)
    L.check(lib.cogv_colsum(dt_code(dy), _p(dy), M, N, dy.stride(0), _p(out), int(accumulate), _p(ws), ws.numel(),
                            _stream()), "cogv_colsum")
    return out


# ------------------------------------------------------------------------------------------ cross entropy
def ce_fwd(logits2d, target1d, vocab_start, want_loss=True):
    _need_gpu(logits2d, target1d)
    assert logits2d.dim() == 2 and logits2d.is_contiguous() and target1d.is_contiguous()
    rows, v = logits2d.shape
    f = lambda: torch.empty(rows, dtype=torch.float32, device=logits2d.device)
    rowmax, sumexp, pred = f(), f(), f()
    loss = f() if want_loss else None
    L.check(L.lib().cogv_ce_fwd(dt_code(logits2d), _p(logits2d), _p(target1d), int(vocab_start), rows, v, _p(rowmax),
                                _p(sumexp), _p(pred), _p(loss), _stream()), "cogv_ce_fwd")
    return rowmax, sumexp, pred, loss


def ce_bwd(logits2d, target1d, vocab_start, gmax, gsum, grad, out=None):
    _need_gpu(logits2d)
    rows, v = logits2d.shape
    if out is None:
        out = torch.empty_like(logits2d)
    L.check(L.lib().cogv_ce_bwd(dt_code(logits2d), _p(logits2d), _p(target1d), int(vocab_start), rows, v, _p(gmax),
                                _p(gsum), _p(grad), _p(out), _stream()), "cogv_ce_bwd")
    return out


# ------------------------------------------------------------------------------------------ optimizer
def grad_stats(flat_grads, chunk_start, chunk_len, chunk_norm, stats):
    _need_gpu(flat_grads)
    lib = L.lib()
    # the partial sums of the pass: scratch owned by this side of the C ABI, one buffer per (device, stream) so that two
    # streams running the pass concurrently do not share it
    # (keyed on the stream handle as an INT: _stream() is a ctypes.c_void_p, which '%x' refuses on any non-default stream)
    sid = int(torch.cuda.current_stream(flat_grads.device).cuda_stream or 0)
    ws = workspace("grad_stats_%x" % sid, lib.cogv_grad_stats_workspace_bytes(), flat_grads.device, floor=0)
    with timed_launch("grad_stats", 0.0, flat_grads.numel() * flat_grads.element_size(), "overflow flag + per-chunk sum of squares"):
        L.check(lib.cogv_grad_stats(dt_code(flat_grads), _p(flat_grads), _p(chunk_start), _p(chunk_len),
                                    _p(chunk_norm), chunk_start.numel(), _p(stats), _p(ws), ws.numel(), _stream()), "cogv_grad_stats")


def adamw_step(params, grads, master, exp_avg, exp_avg_sq, chunk_start, chunk_len, chunk_group, lrs, wds, beta1,
               beta2, eps, step, inv_loss_scale=1.0, max_grad_norm=0.0, stats=None, sumsq_override=None,
               bias_correction=True, adam_w_mode=True):
    _need_gpu(params, grads, master)
    d = L.AdamDesc()
    d.dtype = dt_code(params)
    d.params, d.grads = params.data_ptr(), grads.data_ptr()
    d.master, d.exp_avg, d.exp_avg_sq = master.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr()
    d.chunk_start, d.chunk_len, d.chunk_group = chunk_start.data_ptr(), chunk_len.data_ptr(), chunk_group.data_ptr()
    d.nchunks = chunk_start.numel()
    assert len(lrs) <= 8
    for i, (lr, wd) in enumerate(zip(lrs, wds)):
        d.lr[i], d.weight_decay[i] = float(lr), float(wd)
    d.beta1, d.beta2, d.eps = float(beta1), float(beta2), float(eps)
    d.beta1_d, d.beta2_d = float(beta1), float(beta2)
    d.step, d.bias_correction, d.adam_w_mode = int(step), int(bias_correction), int(adam_w_mode)
    d.inv_loss_scale, d.max_grad_norm = float(inv_loss_scale), float(max_grad_norm)
    d.stats = None if stats is None else stats.data_ptr()
    d.norm_sumsq_override = None if sumsq_override is None else sumsq_override.data_ptr()
    # algorithmic bytes per parameter: 16-bit gradient in, master + two moments read and written (3 x 8), 16-bit parameter out
    with timed_launch("adamw", 0.0, params.numel() * (2 * params.element_size() + 24), "fused unscale + clip + AdamW + 16-bit copy-out"):
        L.check(L.lib().cogv_adamw_step(C.byref(d), _stream()), "cogv_adamw_step")


def cast_flat(src_half, dst_f32):
    _need_gpu(src_half, dst_f32)
    L.check(L.lib().cogv_cast_flat(dt_code(src_half), _p(src_half), _p(dst_f32), src_half.numel(), _stream()),
            "cogv_cast_flat")


def cast_flat_back(src_f32, dst_half):
    _need_gpu(src_f32, dst_half)
    L.check(L.lib().cogv_cast_flat_back(dt_code(dst_half), _p(src_f32), _p(dst_half), src_f32.numel(), _stream()),
            "cogv_cast_flat_back")
