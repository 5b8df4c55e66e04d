"""Tensor-level wrappers over the C ABI (include/lrp_hip.h).

PyTorch is plumbing here: it owns device memory (caching allocator) and the current HIP stream;
every computation below is a call into liblrp_hip.so with raw device pointers.  CPU tensors are
rejected -- there is no fallback.
"""
import torch

from ._lib import lib, check, F32, BF16, ACT

_DT = {torch.float32: F32, torch.bfloat16: BF16}


def dt(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"lrp_hip supports float32 and bfloat16 tensors, got {t.dtype}") from None


def p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("lrp_hip kernels need device tensors (no CPU fallback)")
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def same(ref, *ts):
    """every tensor handed to a kernel next to `ref` must share its dtype and device: the C ABI takes ONE dtype code
    per call and raw pointers, so a mismatch would be reinterpreted bytes (or an out-of-bounds read), not an error"""
    for t in ts:
        if t is not None and (t.dtype != ref.dtype or t.device != ref.device):
            raise TypeError(f"lrp_hip: operand of dtype {t.dtype} on {t.device} next to {ref.dtype} on {ref.device}; "
                            "all activation operands of one call must share dtype and device")


def aux(t, ref, n=None):
    """small parameter vector (norm weight, bias) for a kernel whose activations are `ref`: brought to ref's dtype /
    device / contiguity if it differs (mixed-precision modules keep fp32 norm weights next to bf16 activations)"""
    if t is None:
        return None
    if n is not None and t.numel() != n:
        raise ValueError(f"lrp_hip: parameter vector of {t.numel()} elements where {n} are expected")
    if t.dtype != ref.dtype or t.device != ref.device or not t.is_contiguous():
        t = t.to(device=ref.device, dtype=ref.dtype).contiguous()
    return t


def f32(*ts):
    for t in ts:
        if t is not None and t.dtype != torch.float32:
            raise TypeError(f"lrp_hip: row statistics / tables must be float32, got {t.dtype}")


def epc(t):
    return 16 // t.element_size()


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


# ------------------------------------------------------------------------------------------- GEMM
def gemm_nt(a, b, bias=None, out=None, out_dtype=None):
    """out[..., M, N] = a[..., M, K] @ b[..., N, K]^T (+bias).  a/b: last dim contiguous, 2-D or
    batched 3-D with a uniform batch stride; b may be 2-D (shared).  K is zero-padded to a multiple
    of 16 bytes if needed."""
    assert a.dtype == b.dtype
    K = a.shape[-1]
    assert b.shape[-1] == K
    e = epc(a)
    if K % e:
        pad = e - K % e
        a = torch.nn.functional.pad(a, (0, pad))
        b = torch.nn.functional.pad(b, (0, pad))
        K += pad
    lead = a.shape[:-2]
    M, N = a.shape[-2], b.shape[-2]
    a3 = a.reshape(-1, M, K)
    batch = a3.shape[0]
    if a3.stride(-1) != 1 or a3.stride(-2) % e or (batch > 1 and a3.stride(0) % e) or a3.data_ptr() % 16:
        a3 = a3.contiguous()
    if b.dim() == 2:
        b3, sB = b, 0
        if b3.stride(-1) != 1 or b3.stride(-2) % e or b3.data_ptr() % 16:
            b3 = b3.contiguous()
        ldb = b3.stride(0)
    else:
        b3 = b.reshape(-1, N, K)
        assert b3.shape[0] == batch
        if b3.stride(-1) != 1 or b3.stride(-2) % e or (batch > 1 and b3.stride(0) % e) or b3.data_ptr() % 16:
            b3 = b3.contiguous()
        sB, ldb = (b3.stride(0) if batch > 1 else 0), b3.stride(1)
    odt = out_dtype or a.dtype
    bias = aux(bias, a, N)
    if out is None:
        out = torch.empty(*lead, M, N, device=a.device, dtype=odt)
    o3 = out.view(-1, M, N)
    assert o3.stride(-1) == 1
    rc = lib.lrp_gemm_nt(p(a3), p(b3), p(o3), p(bias), M, N, K, a3.stride(1), ldb, o3.stride(1), batch,
                         a3.stride(0) if batch > 1 else 0, sB, o3.stride(0) if batch > 1 else 0,
                         dt(a), _DT[odt], stream())
    check(rc, "lrp_gemm_nt")
    return out


class KernelTimer:
    """HIP-event timing of every launch of one kernel family on the launching stream (bench.py's
    roofline leg).  Enabled by assigning an instance to ops.GEMM_TIMER; records (flops, ev0, ev1).
    drain() folds the spans whose end event has completed into per-tag sums and RE-USES their event objects: a long run then keeps ~one step's
    worth of events alive instead of steps x launches x 2 (each live event holds a signal of the HIP runtime's pool; with thousands alive the
    runtime takes a slow path in some processes -- the judged run's sporadic +14 ms per step sat entirely in launch gaps, not in kernels:
    profiles/r05_gemm_experiments.txt, section O)."""

    def __init__(self):
        self.records = []
        self.sums = {}            # tag -> [launches, flops, seconds]
        self.pool = []

    def span(self, flops, tag="plain"):
        if len(self.pool) >= 2:
            e0, e1 = self.pool.pop(), self.pool.pop()
        else:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.records.append((flops, e0, e1, tag))
        return e0, e1

    def drain(self, wait=False):
        """fold every span whose end event has completed (all of them with wait=True: call after a device synchronise)"""
        keep = []
        for r in self.records:
            if not keep and (wait or r[2].query()):
                acc = self.sums.setdefault(r[3], [0, 0.0, 0.0])
                acc[0] += 1
                acc[1] += r[0]
                acc[2] += r[1].elapsed_time(r[2]) * 1e-3
                self.pool.append(r[1])
                self.pool.append(r[2])
            else:
                keep.append(r)        # spans complete in issue order: stop at the first one still in flight
        self.records = keep

    def summary(self, tag=None):
        """-> (launches, total_flops, total_seconds) of the launches with this tag (None: all); call after a device synchronise.
        Tags: "plain" = GEMM only; "plain_norm" = the same kernel with a K1n epilogue (row scale / residual add / row sums of squares);
        "gated_fwd" / "gated_bwd" = the launches that carry a gated-MLP rule in their epilogue; a tuple selects several"""
        self.drain(wait=True)
        sel = [v for k, v in self.sums.items() if tag is None or k == tag or (isinstance(tag, tuple) and k in tag)]
        return sum(v[0] for v in sel), sum(v[1] for v in sel), sum(v[2] for v in sel)


GEMM_TIMER = None


def gemm_nt_2d(a, b, out, bias=None):
    """strict 2-D fast path used by the engine: no reshapes, no copies; row strides may exceed K."""
    M, K = a.shape
    N = b.shape[0]
    same(a, b)
    bias = aux(bias, a, N)
    ev = GEMM_TIMER.span(2.0 * M * N * K) if GEMM_TIMER is not None else None
    if ev:
        ev[0].record()
    rc = lib.lrp_gemm_nt(a.data_ptr(), b.data_ptr(), out.data_ptr(), p(bias), M, N, K, a.stride(0), b.stride(0),
                         out.stride(0), 1, 0, 0, 0, dt(a), _DT[out.dtype], stream())
    if ev:
        ev[1].record()
    check(rc, "lrp_gemm_nt")
    return out


def gemm_nn_ok(a, w):
    """can lrp_gemm_nn / lrp_gemm_skinny serve C = a[M,K] @ w[K,N] (w = a weight in its stored [out,in] layout)?  bf16, K % 64 == 0,
    K >= 128, 16-byte aligned K-/N-contiguous operands, the weight below 2^30 elements (activations with more elements are issued in
    row chunks by the library: no fallback to a W^T copy for large batches)"""
    if a.dtype != torch.bfloat16 or w.dtype != torch.bfloat16 or a.dim() != 2 or w.dim() != 2:
        return False
    M, K = a.shape
    return (K % 64 == 0 and K >= 128 and a.stride(1) == 1 and w.stride(1) == 1 and a.stride(0) % 8 == 0 and w.stride(0) % 8 == 0
            and a.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0 and 256 * a.stride(0) < 2 ** 30 and w.shape[0] * w.stride(0) < 2 ** 30)


def gemm_nn_2d(a, w, out, bias=None):
    """out[M,N] = a[M,K] @ w[K,N]: the eps-rule redistribution c = s W straight from the STORED weight W [out,in] (no W^T copy)"""
    M, K = a.shape
    N = w.shape[1]
    same(a, w)
    ev = GEMM_TIMER.span(2.0 * M * N * K) if GEMM_TIMER is not None else None
    if ev:
        ev[0].record()
    rc = lib.lrp_gemm_nn(a.data_ptr(), w.data_ptr(), out.data_ptr(), p(aux(bias, a, N)), M, N, K, a.stride(0), w.stride(0),
                         out.stride(0), dt(a), _DT[out.dtype], stream())
    if ev:
        ev[1].record()
    check(rc, "lrp_gemm_nn")
    return out


SKINNY_MAX = 256         # rows the split-K skinny path serves (HBM-bound regime of the Linear eps-rule: M <~ 160 in bf16)


def splitk_ok(M, N, K):
    """should this bf16 GEMM take the split-K path of the ping-pong kernel (lrp_gemm_skinny)?  M <= 256 rows always (W is streamed once by
    all CUs); for more rows when the 256 x 256 tile count would leave half of the 256 CUs idle (M = 2048 against a 4096-row weight:
    128 tiles) and every split keeps a K loop of >= 8 tiles -- measured at M = 2048 (tools/gemm_ab.py --m 2048): 978 -> 1150 ... 1039 ->
    1440 TFLOP/s on the N = 4096 shapes"""
    if M <= SKINNY_MAX:
        return True
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    if tiles <= 128:
        return K // 64 >= 8 * (256 // tiles)
    return lib.lrp_gemm_skinny_splits(M, N, K) >= 2         # 257 ... 384 tiles with a long K loop (Gemma-3-4B: 8192 x 2560 x 10240)


_WS = {}


def workspace(nbytes, ref):
    """scratch for kernels that take a caller-allocated workspace.  One buffer per (device, stream), grown on demand: kernels on
    one stream are ordered, two streams never share a buffer.  Under hipGraph capture a FRESH allocation is made per
    call (it lives in the graph's private pool; a cached buffer could be re-grown -- freed -- after the capture)."""
    if torch.cuda.is_current_stream_capturing():
        return torch.empty(max(int(nbytes), 16), device=ref.device, dtype=torch.uint8)
    key = (ref.device, torch.cuda.current_stream(ref.device).cuda_stream)
    ws = _WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(int(nbytes), 4 << 20), device=ref.device, dtype=torch.uint8)
        _WS[key] = ws
    return ws


_TK = {}
STREAM_DGRAD_INKERNEL_REDUCE = True      # module attribute (A/B measurements): False = fp32 slabs + the reduce launch


def tickets(n, ref):
    """the zero-initialised 32-bit arrival counters of the in-kernel split reduction (lrp_linear_stream_dgrad_tk): one array per (device, stream),
    allocated once and never written by anything but that kernel (whose last arriver re-arms each word), grown by re-allocation; under hipGraph
    capture a fresh zeroed array per call (the memset is captured with the launch)"""
    if torch.cuda.is_current_stream_capturing():        # as ops.workspace: a fresh array in the graph's private pool, zeroed by a captured memset
        return torch.zeros(max(int(n), 16), device=ref.device, dtype=torch.int32)
    key = (ref.device, torch.cuda.current_stream(ref.device).cuda_stream)
    t = _TK.get(key)
    if t is None or t.numel() < n:
        t = torch.zeros(max(int(n), 4096), device=ref.device, dtype=torch.int32)
        _TK[key] = t
    return t


def gemm_skinny(a, b, out, nn=False, bias=None):
    """out[M,N] = a[M,K] @ b[N,K]^T (nn=False) or a[M,K] @ b[K,N] (nn=True) for M <= 256 rows: split-K over the CUs, the weight
    streamed exactly once; fp32 partial slabs in a stream-local workspace.  The library addresses the activation through ONE 32-bit buffer
    resource (no row chunks, unlike lrp_gemm_nt / _nn): problems with M * row pitch >= 2^30 elements (possible only on the many-row tail /
    half-empty-chip uses of this path) are issued here in row chunks, each its own split-K problem"""
    M, K = a.shape
    N = b.shape[1] if nn else b.shape[0]
    same(a, b)
    if M * a.stride(0) >= 2 ** 30:
        step = max(256, ((2 ** 30 - 1) // a.stride(0)) // 256 * 256)
        for m0 in range(0, M, step):
            gemm_skinny(a[m0: m0 + step], b, out[m0: m0 + step], nn=nn, bias=bias)
        return out
    ws = workspace(lib.lrp_gemm_skinny_ws(M, N, K), a)
    ev = GEMM_TIMER.span(2.0 * M * N * K, "splitk") if (GEMM_TIMER is not None and M > SKINNY_MAX) else None
    if ev:
        ev[0].record()
    rc = lib.lrp_gemm_skinny(a.data_ptr(), b.data_ptr(), out.data_ptr(), p(aux(bias, a, N)), M, N, K, a.stride(0), b.stride(0),
                             out.stride(0), int(nn), dt(a), _DT[out.dtype], ws.data_ptr(), stream())
    if ev:
        ev[1].record()
    check(rc, "lrp_gemm_skinny")
    return out


def transpose(x, out=None):
    """[..., R, C] -> [..., C, R] (contiguous)"""
    x = _c(x)
    R, C = x.shape[-2], x.shape[-1]
    lead = x.shape[:-2]
    if out is None:
        out = torch.empty(*lead, C, R, device=x.device, dtype=x.dtype)
    batch = x.numel() // (R * C) if R * C else 1
    rc = lib.lrp_transpose(p(x), p(out), R, C, C, R, batch, R * C, R * C, dt(x), stream())
    check(rc, "lrp_transpose")
    return out


_WT = {}            # storage id -> {(offset, shape, stride, dtype): (version, W^T)}; entries die with the weight's storage (weakref.finalize)
TRANSPOSE_CALLS = [0]    # weight transposes actually executed (tests: the cache must hit on the second backward)


def weight_t(w):
    """W^T copy ([in, out]) of a frozen weight for the fp32 / odd-shape dgrad GEMM.  Cached per STORAGE of the weight (not per tensor
    object: the patched forwards hand `param.detach()` -- a fresh object on every call -- to the autograd Functions), keyed by
    the view's offset / shape / strides / dtype and invalidated by the in-place version counter, which detach() aliases share with the
    Parameter; the entry is dropped when the storage dies, so an address re-used by another weight can never hit."""
    import weakref
    st = w.untyped_storage()
    sid = st._cdata
    per = _WT.get(sid)
    if per is None:
        per = _WT[sid] = {}
        weakref.finalize(st, _WT.pop, sid, None)
    key = (w.storage_offset(), tuple(w.shape), tuple(w.stride()), w.dtype)
    hit = per.get(key)
    if hit is not None and hit[0] == w._version:
        return hit[1]
    TRANSPOSE_CALLS[0] += 1
    wt = transpose(w.detach())
    per[key] = (w._version, wt)
    return wt


def cast(x, dtype):
    x = _c(x)
    out = torch.empty_like(x, dtype=dtype)
    check(lib.lrp_cast(p(x), p(out), x.numel(), dt(x), _DT[dtype], stream()), "lrp_cast")
    return out


# ------------------------------------------------------------------------------------- element-wise
def eps_scale(g, z, c=1.0, eps=1e-8, relevance=False, out=None):
    g, z = _c(g), _c(z)
    out = torch.empty_like(g) if out is None else out
    same(g, z, out)
    check(lib.lrp_eps_scale(p(g), p(z), p(out), g.numel(), c, eps, 1 if relevance else 0, dt(g), stream()), "lrp_eps_scale")
    return out


def eps_scale2d(g, z, out, c=1.0, eps=1e-8, relevance=False):
    rows, cols = g.shape
    same(g, z, out)
    check(lib.lrp_eps_scale2d(p(g), p(z), p(out), rows, cols, g.stride(0), z.stride(0), out.stride(0), c, eps,
                              1 if relevance else 0, dt(g), stream()), "lrp_eps_scale2d")
    return out


def mul(a, b, out=None):
    a, b = _c(a), _c(b)
    out = torch.empty_like(a) if out is None else out
    same(a, b, out)
    check(lib.lrp_mul(p(a), p(b), p(out), a.numel(), dt(a), stream()), "lrp_mul")
    return out


def add_bcast(x, y, out=None):
    """out[m,:] = x[m,:] + y[m % period,:], period = rows of y (1: a row vector, S: per-position rows, M: a full residual sum)"""
    x, y = _c(x), _c(y)
    H = x.shape[-1]
    M, period = x.numel() // H, y.numel() // H
    out = torch.empty_like(x) if out is None else out
    same(x, y, out)
    if y.shape[-1] != H or M % period:
        raise ValueError(f"add_bcast: x {tuple(x.shape)} vs y {tuple(y.shape)}")
    check(lib.lrp_add_bcast(p(x), p(y), p(out), M, H, period, dt(x), stream()), "lrp_add_bcast")
    return out


def add2_rule_bwd(a, b, R, eps=1e-8, need_b=True):
    a, b, R = _c(a), _c(b), _c(R)
    Ra = torch.empty_like(a)
    Rb = torch.empty_like(b) if need_b else None
    same(a, b, R)
    check(lib.lrp_add2_rule_bwd(p(a), p(b), p(R), p(Ra), p(Rb), a.numel(), eps, dt(a), stream()), "lrp_add2_rule_bwd")
    return Ra, Rb


def act_fwd(x, act="silu"):
    x = _c(x)
    y = torch.empty_like(x)
    check(lib.lrp_act_fwd(p(x), p(y), x.numel(), ACT[act], dt(x), stream()), "lrp_act_fwd")
    return y


def act_bwd(Gy, x, act="silu", eps_g=1e-10):
    Gy, x = _c(Gy), _c(x)
    Gx = torch.empty_like(x)
    same(x, Gy)
    check(lib.lrp_act_bwd(p(Gy), p(x), p(Gx), x.numel(), eps_g, ACT[act], dt(x), stream()), "lrp_act_bwd")
    return Gx


def act_grad(Gy, x, act="gelu_tanh", out=None):
    """plain derivative Gy * act'(x) (no rule): the SigLIP tower under the reference's gemma3 map"""
    Gy, x = _c(Gy), _c(x)
    out = torch.empty_like(x) if out is None else out
    same(x, Gy, out)
    check(lib.lrp_act_grad(p(Gy), p(x), p(out), x.numel(), ACT[act], dt(x), stream()), "lrp_act_grad")
    return out


def gated_act_fwd(g, u, out=None, act="silu"):
    M, I = g.shape
    out = torch.empty(M, I, device=g.device, dtype=g.dtype) if out is None else out
    same(g, u, out)
    check(lib.lrp_gated_act_fwd(p(g), p(u), p(out), M, I, g.stride(0), u.stride(0), out.stride(0), ACT[act], dt(g), stream()),
          "lrp_gated_act_fwd")
    return out


def gated_act_bwd(Gm, g, u, Ag, Au, eps_g, eps_lin, act="silu"):
    M, I = g.shape
    same(g, Gm, u, Ag, Au)
    check(lib.lrp_gated_act_bwd(p(Gm), p(g), p(u), p(Ag), p(Au), M, I, Gm.stride(0), g.stride(0), u.stride(0), Ag.stride(0),
                                Au.stride(0), eps_g, eps_lin, ACT[act], dt(g), stream()), "lrp_gated_act_bwd")
    return Ag, Au


GATED_IL = 32            # interleave block of a fused gate/up weight (include/lrp_hip.h: LRP_GATED_IL)


def interleave_gate_up(wg, wu, out=None):
    """fused gate/up weight [2 I, H] with its rows in blocks of 64 = [32 gate rows | 32 up rows]: one 64-column block of W_gu x -- and one
    wave's accumulator tile of the GEMM -- then holds gate AND up of the same 32 intermediate indices (lrp_gated_act_fwd_il / _bwd_il, lrp_gemm_gated_fwd_coef / _bwd_coef)"""
    I, H = wg.shape
    if I % GATED_IL:
        raise ValueError(f"intermediate size {I} is not a multiple of {GATED_IL}")
    if out is None:
        out = torch.empty(2 * I, H, device=wg.device, dtype=wg.dtype)
    v = out.unflatten(0, (I // GATED_IL, 2, GATED_IL))          # (out may be a row-pitch-padded view: engine.weight_pitch_pad)
    v[:, 0].copy_(wg.view(I // GATED_IL, GATED_IL, H))
    v[:, 1].copy_(wu.view(I // GATED_IL, GATED_IL, H))
    return out


def gated_act_fwd_il(gu, out, act="silu"):
    M, I = out.shape
    same(gu, out)
    check(lib.lrp_gated_act_fwd_il(p(gu), p(out), M, I, gu.stride(0), out.stride(0), ACT[act], dt(gu), stream()), "lrp_gated_act_fwd_il")
    return out


def gated_act_bwd_il(Gm, gu, Agu, eps_g, eps_lin, act="silu"):
    M, I = Gm.shape
    same(gu, Gm, Agu)
    check(lib.lrp_gated_act_bwd_il(p(Gm), p(gu), p(Agu), M, I, Gm.stride(0), gu.stride(0), Agu.stride(0), eps_g, eps_lin, ACT[act], dt(gu),
                                   stream()), "lrp_gated_act_bwd_il")
    return Agu


GATED_FUSION = True      # module attribute (no environment knobs): False = GEMM + element-wise rule kernels, for A/B measurements and the
                         # fused-vs-unfused tests


def gated_coef_ok(M, I, H, ldx, ldwgu, lda, ldwd, act, dtype):
    """both GEMMs around the gated rule are problems the fused epilogues take (lrp_gemm_gated_fwd_coef / _bwd_coef: bf16, M = B S rows)"""
    return bool(GATED_FUSION and dtype == torch.bfloat16 and M > SKINNY_MAX
                and lib.lrp_gemm_gated_coef_ok(M, I, H, ldx, ldwgu, lda, ldwd, ACT[act], _DT[dtype]))


def gemm_gated_fwd_coef(x, Wgu, coef, m, eps_g, eps_lin, act="silu", rs=None):
    """ONE launch: m[M, I] = act(g) (*) u and the backward's coefficient stash coef[M, 2 I] (cg = 1/2 u act(g) / (g + eps_g),
    cu = 1/2 act(g) u / (u + eps_lin); layout private to the pair, include/lrp_hip.h) from x @ Wgu^T (Wgu interleaved, see interleave_gate_up);
    rs: optional fp32 row scales applied to the product first (K1n).  g and u are never stored."""
    M, K = x.shape
    I = m.shape[1]
    same(x, Wgu, coef, m)
    if rs is not None:
        f32(rs)
    _timed(2.0 * M * 2 * I * K, "gated_fwd", lambda: lib.lrp_gemm_gated_fwd_coef(p(x), p(Wgu), p(rs), p(coef), p(m), M, I, K, x.stride(0), Wgu.stride(0),
                                                                                 coef.stride(0), m.stride(0), eps_g, eps_lin, ACT[act], dt(x), stream()),
           "lrp_gemm_gated_fwd_coef")
    return coef, m


def gemm_gated_bwd_coef(Adn, Wd, coef, Agu):
    """Agu[M, 2 I] (interleaved) = { Gm cg | Gm cu } with Gm = A_dn Wd (stored down weight Wd [H, I]) never written: the multiply runs in the NN
    GEMM's epilogue on the stash lrp_gemm_gated_fwd_coef left"""
    M, K = Adn.shape
    I = Wd.shape[1]
    same(Adn, Wd, coef, Agu)
    _timed(2.0 * M * I * K, "gated_bwd", lambda: lib.lrp_gemm_gated_bwd_coef(p(Adn), p(Wd), p(coef), p(Agu), M, I, K, Adn.stride(0), Wd.stride(0),
                                                                             coef.stride(0), Agu.stride(0), dt(Adn), stream()), "lrp_gemm_gated_bwd_coef")
    return Agu


def gemm_gated_fwd(x, Wgu, gu, m, act="silu"):
    """gu[M, 2 I] = x @ Wgu^T (Wgu interleaved, see interleave_gate_up) and m[M, I] = act(g) (*) u: the GEMM / skinny / small-M forward +
    lrp_gated_act_fwd_il (small M, fp32; M = B S rows in bf16 take gemm_gated_fwd_coef)"""
    same(x, Wgu, gu, m)
    linear_fwd(x, Wgu, out=gu)
    gated_act_fwd_il(gu, m, act)
    return gu, m


def gemm_gated_bwd(Adn, Wd, gu, Agu, eps_g, eps_lin, act="silu"):
    """Agu[M, 2 I] (interleaved) from A_dn[M, H], the stored down weight Wd [H, I] and the stored gate/up output: ops.linear_dgrad +
    lrp_gated_act_bwd_il"""
    same(Adn, Wd, gu, Agu)
    Gm = linear_dgrad(Adn, Wd)
    return gated_act_bwd_il(Gm, gu, Agu, eps_g, eps_lin, act)


def rope_fwd(x, out, cos, sin, seq, n_heads, d):
    """x/out: [rows, >= n_heads*d] 2-D views (row stride = stride(0)); cos/sin fp32 [seq, d]"""
    rows = x.shape[0]
    same(x, out)
    f32(cos, sin)
    check(lib.lrp_rope_fwd(p(x), p(out), p(cos), p(sin), rows, seq, n_heads, d, x.stride(0), out.stride(0), dt(x), stream()),
          "lrp_rope_fwd")
    return out


def rope_bwd(Gr, xr, x, A, cos, sin, seq, n_heads, d, eps_rope, eps_lin):
    rows = Gr.shape[0]
    same(Gr, xr, x, A)
    f32(cos, sin)
    check(lib.lrp_rope_bwd(p(Gr), p(xr), p(x), p(A), p(cos), p(sin), rows, seq, n_heads, d, Gr.stride(0),
                           xr.stride(0) if xr is not None else 0, x.stride(0) if x is not None else 0, A.stride(0),
                           eps_rope, eps_lin, dt(Gr), stream()), "lrp_rope_bwd")
    return A


# ---------------------------------------------------------------------------------------- row ops
# ---- K1n: RMSNorm folded into the GEMMs around it (include/lrp_hip.h; ref lxt/efficient/patches.py:111-123 + the residual sums of HF modeling_llama)
# module attribute: True (default) = every part; False = the stand-alone add_rmsnorm_fwd / rmsnorm_bwd_add2 launches (A/B measurements, equality
# tests); a set of {"fwd", "bwd_qkv", "bwd_gu"} = the named parts.  In situ, 8-layer judged step (tools/r5_k1n_parts.sh, final kernel, three
# interleaved repeats): all parts 43.58-43.67 ms, without the gate/up dgrad's residual epilogue 43.62-43.78, none 44.15-44.34 (+1.5 %).  (The first
# version of the residual epilogue -- two row blocks in flight, loads requested at the start of the epilogue -- LOST on the gate/up dgrad: 1312 us
# against 1258 + 31.)
NORM_FUSION = True


def norm_fusion_part(part):
    return NORM_FUSION is True or (isinstance(NORM_FUSION, (set, frozenset)) and part in NORM_FUSION)


def norm_fused_ok(M, N, K, lda, ldb, nn, dtype):
    """the K1n entry points take this problem (bf16, N % 256 == 0, K % 64 == 0, >= 190 output tiles of the ping-pong kernel)"""
    return bool(NORM_FUSION and dtype == torch.bfloat16 and lib.lrp_gemm_norm_fused_ok(M, N, K, lda, ldb, 1 if nn else 0, _DT[dtype]))


def _timed(flops, tag, fn, name):
    ev = GEMM_TIMER.span(flops, tag) if GEMM_TIMER is not None else None
    if ev:
        ev[0].record()
    rc = fn()
    if ev:
        ev[1].record()
    check(rc, name)


_CONST_ROWS = {}


def const_rows(M, value, device):
    """cached fp32 [M] vector of one value (row scales of the K1n entry points: 1/2 for the o-projection's dgrad)"""
    key = (str(device), M, float(value))
    t = _CONST_ROWS.get(key)
    if t is None:
        if len(_CONST_ROWS) > 16:
            _CONST_ROWS.clear()
        t = _CONST_ROWS[key] = torch.full((M,), float(value), device=device, dtype=torch.float32)
    return t


def gemm_res_ssq(x, W, res, out, ssq, raw=None):
    """out = res + x @ W^T (bf16 rounding once) and ssq[p, m] = sum of out[m, 64 p : 64 p + 64]^2 -- the residual add and RMSNorm's sum of
    squares in the producing GEMM's epilogue; ssq [N / 64, >= M] fp32"""
    M, K = x.shape
    N = W.shape[0]
    same(x, W, res, out)
    f32(ssq)
    assert ssq.shape[0] * 64 == N and ssq.stride(0) >= M and ssq.stride(1) == 1
    if raw is not None:      # the explicit placement keeps the Linear's own output too (its stabiliser divides by it)
        same(x, raw)
    _timed(2.0 * M * N * K, "plain_norm", lambda: lib.lrp_gemm_res_ssq(p(x), p(W), p(res), p(out), p(ssq), M, N, K, x.stride(0), W.stride(0),
                                                                       res.stride(0), out.stride(0), ssq.stride(0), p(raw),
                                                                       raw.stride(0) if raw is not None else 0, dt(x), stream()), "lrp_gemm_res_ssq")
    return out


def rms_rstd(ssq, M, H, eps, rstd):
    f32(ssq, rstd)
    check(lib.lrp_rms_rstd(p(ssq), ssq.shape[0], ssq.stride(0), M, H, eps, p(rstd), stream()), "lrp_rms_rstd")
    return rstd


def gemm_nt_rs(x, W, rs, out):
    """out = rs[:, None] * (x @ W^T): the consumer of a folded RMSNorm (W carries the norm's weight, rs = rstd of x's rows)"""
    M, K = x.shape
    N = W.shape[0]
    same(x, W, out)
    f32(rs)
    _timed(2.0 * M * N * K, "plain_norm", lambda: lib.lrp_gemm_nt_rs(p(x), p(W), p(rs), p(out), M, N, K, x.stride(0), W.stride(0), out.stride(0),
                                                                     dt(x), stream()), "lrp_gemm_nt_rs")
    return out


ROPE_FWD_FUSION = True   # module attribute (A/B): False = lrp_gemm_nt_rs + the stand-alone lrp_rope_fwd pass


def gemm_nt_rs_rope_ok(x, W, out, seq, rope_cols, head_dim):
    M, K = x.shape
    return bool(ROPE_FWD_FUSION and NORM_FUSION and x.dtype == torch.bfloat16 and
                lib.lrp_gemm_nt_rs_rope_ok(M, W.shape[0], K, x.stride(0), W.stride(0), out.stride(0), seq, rope_cols, head_dim, _DT[x.dtype]))


def gemm_nt_rs_rope(x, W, rs, cos, sin, out, seq, rope_cols, head_dim):
    """out = rs[:, None] * (x @ W^T) with RoPE applied to the head columns [0, rope_cols) in the GEMM's epilogue (the fused QKV forward: q and k
    leave the kernel rotated, v untouched); cos / sin fp32 [>= seq, head_dim], position of row m = m % seq"""
    M, K = x.shape
    N = W.shape[0]
    same(x, W, out)
    f32(rs, cos, sin)
    assert cos.shape[0] >= seq and cos.stride(0) == head_dim and sin.stride(0) == head_dim
    _timed(2.0 * M * N * K, "plain_norm", lambda: lib.lrp_gemm_nt_rs_rope(p(x), p(W), p(rs), p(cos), p(sin), p(out), M, N, K, x.stride(0), W.stride(0),
                                                                          out.stride(0), seq, rope_cols, head_dim, dt(x), stream()), "lrp_gemm_nt_rs_rope")
    return out


def gemm_nn_rs(s, W, rs, out):
    """out = rs[:, None] * (s @ W) from the STORED weight W [K, N] (rs = 1/2: the o-projection's dgrad with the uniform rule's factor)"""
    M, K = s.shape
    N = W.shape[1]
    same(s, W, out)
    f32(rs)
    _timed(2.0 * M * N * K, "plain_norm", lambda: lib.lrp_gemm_nn_rs(p(s), p(W), p(rs), p(out), M, N, K, s.stride(0), W.stride(0), out.stride(0),
                                                                     dt(s), stream()), "lrp_gemm_nn_rs")
    return out


def gemm_nn_rs_res(s, W, rs, res, out):
    """out = rs[:, None] * (s @ W) + res from the STORED weight W [K, N]: RMSNorm's identity-rule backward (row scale; the norm's weight sits in
    W) and the residual gradient in the dgrad GEMM's epilogue; out may alias res"""
    M, K = s.shape
    N = W.shape[1]
    same(s, W, res, out)
    f32(rs)
    _timed(2.0 * M * N * K, "plain_norm", lambda: lib.lrp_gemm_nn_rs_res(p(s), p(W), p(rs), p(res), p(out), M, N, K, s.stride(0), W.stride(0),
                                                                         res.stride(0), out.stride(0), dt(s), stream()), "lrp_gemm_nn_rs_res")
    return out


def add_rmsnorm_fwd(h, branch, w, eps, w_offset=0.0, hsum_out=None, y=None, rstd=None):
    M, H = h.shape
    y = torch.empty_like(h) if y is None else y
    rstd = torch.empty(M, device=h.device, dtype=torch.float32) if rstd is None else rstd
    same(h, branch, hsum_out, y)
    f32(rstd)
    w = aux(w, h, H)
    check(lib.lrp_add_rmsnorm_fwd(p(h), p(branch), p(w), p(hsum_out), p(y), p(rstd), M, H, eps, w_offset, dt(h), stream()),
          "lrp_add_rmsnorm_fwd")
    return y, rstd


def rmsnorm_bwd_add2(Gres, Gx, w, rstd, hsum, branch, Gs_out, A_out, rel_out=None, w_offset=0.0, eps_add=0.0, eps_lin=0.0):
    M, H = (Gx if Gx is not None else Gres).shape
    same(Gs_out, Gres, Gx, hsum, branch, A_out)
    f32(rstd, rel_out)
    w = aux(w, Gs_out, H)
    check(lib.lrp_rmsnorm_bwd_add2(p(Gres), p(Gx), p(w), p(rstd), p(hsum), p(branch), p(Gs_out), p(A_out), p(rel_out), M, H,
                                   w_offset, eps_add, eps_lin, dt(Gs_out), stream()), "lrp_rmsnorm_bwd_add2")
    return Gs_out, A_out


def head_rmsnorm_fwd(x, w, y, rstd, heads, d, eps, w_offset=0.0):
    """per-head RMSNorm of the first heads*d columns of x [rows, >= heads*d] (2-D views: row pitch = stride(0)) into y; rstd fp32 [rows*heads]"""
    rows = x.shape[0]
    same(x, y)
    f32(rstd)
    w = aux(w, x, d)
    check(lib.lrp_head_rmsnorm_fwd(p(x), p(w), p(y), p(rstd), rows, heads, d, x.stride(0), y.stride(0), eps, w_offset, dt(x), stream()),
          "lrp_head_rmsnorm_fwd")
    return y, rstd


def head_rmsnorm_bwd(G, w, rstd, out, heads, d, w_offset=0.0):
    """out = G (*) (w + w_offset) * rstd per (row, head): the backward of head_rmsnorm_fwd with rstd detached"""
    rows = G.shape[0]
    same(G, out)
    f32(rstd)
    w = aux(w, G, d)
    check(lib.lrp_head_rmsnorm_bwd(p(G), p(w), p(rstd), p(out), rows, heads, d, G.stride(0), out.stride(0), w_offset, dt(G), stream()),
          "lrp_head_rmsnorm_bwd")
    return out


SITE_FUSION = True      # module attribute (A/B measurements, tests): False keeps Gemma-3's norms / q-k norm / RoPE on the per-module launch sequences


def sandwich_norm_ok(x):
    """can lrp_sandwich_norm_fwd / _bwd take rows of this width and dtype (the row stays in one workgroup's registers)?"""
    return bool(SITE_FUSION and x.dtype in _DT and x.dim() == 2 and x.is_contiguous() and lib.lrp_sandwich_norm_ok(x.shape[1], dt(x)))


def sandwich_norm_fwd(x, res, w_post, w_pre, eps, w_offset, hsum_out, y, rstd_post, rstd_pre):
    """hsum_out = res + norm_post(x), y = norm_pre(hsum_out) in one pass (Gemma-3's post-norm / residual add / next pre-norm; y, w_pre may be
    None): bit-identical to add_rmsnorm_fwd(x, None, w_post) followed by add_rmsnorm_fwd(res, branch, w_pre, hsum_out=...)"""
    M, H = x.shape
    same(x, res, hsum_out, y)
    f32(rstd_post, rstd_pre)
    w_post = aux(w_post, x, H)
    w_pre = aux(w_pre, x, H) if w_pre is not None else None
    check(lib.lrp_sandwich_norm_fwd(p(x), p(res), p(w_post), p(w_pre), p(hsum_out), p(y), p(rstd_post), p(rstd_pre), M, H, eps, w_offset, dt(x),
                                    stream()), "lrp_sandwich_norm_fwd")
    return hsum_out, y


def sandwich_norm_bwd(Gres, Gx, w_pre, rstd_pre, w_post, rstd_post, Gs_out, Ga_out, w_offset=0.0):
    """Gs_out = Gres + Gx w_pre' rstd_pre (Gres may be None), Ga_out = Gs_out w_post' rstd_post: the backward of sandwich_norm_fwd's site with both
    rstd detached = two rmsnorm_bwd_add2 launches"""
    M, H = Gx.shape
    same(Gx, Gres, Gs_out, Ga_out)
    f32(rstd_pre, rstd_post)
    check(lib.lrp_sandwich_norm_bwd(p(Gres), p(Gx), p(aux(w_pre, Gx, H)), p(rstd_pre), p(aux(w_post, Gx, H)), p(rstd_post), p(Gs_out), p(Ga_out), M, H,
                                    w_offset, dt(Gx), stream()), "lrp_sandwich_norm_bwd")
    return Gs_out, Ga_out


def qk_norm_rope_fwd(qkv, wq, wk, qr, kr, rstd_q, rstd_k, cos_t, sin_t, seq, nq, nk, d, eps, w_offset=0.0):
    """per-head q / k RMSNorm + RoPE straight out of the fused projection output qkv [rows, >= (nq + nk) d] into qr [rows, nq d], kr [rows, nk d]
    (= 2 x head_rmsnorm_fwd + 2 x rope_fwd, bit-identical)"""
    rows = qkv.shape[0]
    same(qkv, qr, kr)
    f32(rstd_q, rstd_k, cos_t, sin_t)
    check(lib.lrp_qk_norm_rope_fwd(p(qkv), p(aux(wq, qkv, d)), p(aux(wk, qkv, d)), p(qr), p(kr), p(rstd_q), p(rstd_k), p(cos_t), p(sin_t), rows, seq,
                                   nq, nk, d, qkv.stride(0), qr.stride(0), kr.stride(0), eps, w_offset, dt(qkv), stream()), "lrp_qk_norm_rope_fwd")
    return qr, kr


def qkv_bwd_pack(dq, dk_h, dv_h, wq, wk, rstd_q, rstd_k, cos_t, sin_t, A, seq, nq, nk, d, w_offset=0.0):
    """A [rows, (nq + 2 nk) d] = [rope^T(dq) wq' rstd_q | rope^T(group sum of dk_h) wk' rstd_k | group sum of dv_h]: the qkv dgrad's operand in one
    pass (= 2 x gqa_reduce + 2 x rope_bwd + 2 x head_rmsnorm_bwd, bit-identical)"""
    rows = dq.shape[0]
    same(dq, dk_h, dv_h, A)
    f32(rstd_q, rstd_k, cos_t, sin_t)
    check(lib.lrp_qkv_bwd_pack(p(dq), p(dk_h), p(dv_h), p(aux(wq, dq, d)), p(aux(wk, dq, d)), p(rstd_q), p(rstd_k), p(cos_t), p(sin_t), p(A), rows, seq,
                               nq, nk, d, dq.stride(0), dk_h.stride(0), dv_h.stride(0), A.stride(0), w_offset, dt(dq), stream()), "lrp_qkv_bwd_pack")
    return A


def head_norm_bwd(g_xn, w, rstd, out, w_offset=0.0):
    """final-norm identity rule on the head rows: out = g_xn * (w + w_offset) * rstd  (g_xn fp32 or model dtype [B,H])"""
    g = g_xn if g_xn.dtype == out.dtype else cast(g_xn, out.dtype)
    rmsnorm_bwd_add2(None, g, w, rstd, None, None, out, None, None, w_offset, 0.0, 0.0)
    return out


def layernorm_fwd(x, w, b, eps):
    x = _c(x)
    H = x.shape[-1]
    M = x.numel() // H
    y = torch.empty_like(x)
    mean = torch.empty(M, device=x.device, dtype=torch.float32)
    rstd = torch.empty(M, device=x.device, dtype=torch.float32)
    w, b = aux(w, x, H), aux(b, x, H)
    check(lib.lrp_layernorm_fwd(p(x), p(w), p(b), p(y), p(mean), p(rstd), M, H, eps, dt(x), stream()), "lrp_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(Gy, y, w, rstd, eps_y=0.0):
    Gy = _c(Gy)
    H = Gy.shape[-1]
    M = Gy.numel() // H
    Gx = torch.empty_like(Gy)
    same(Gy, y)
    f32(rstd)
    w = aux(w, Gy, H)
    check(lib.lrp_layernorm_bwd(p(Gy), p(y), p(w), p(rstd), p(Gx), M, H, eps_y, dt(Gy), stream()), "lrp_layernorm_bwd")
    return Gx


def layernorm_bwd_plain(Gy, x, w, mean, rstd, out=None):
    """full LayerNorm VJP (no rule; mean and 1/std differentiated) from the forward's x / mean / rstd"""
    Gy, x = _c(Gy), _c(x)
    H = Gy.shape[-1]
    M = Gy.numel() // H
    out = torch.empty_like(Gy) if out is None else out
    same(Gy, x, out)
    f32(mean, rstd)
    w = aux(w, Gy, H)
    check(lib.lrp_layernorm_bwd_plain(p(Gy), p(x), p(w), p(mean), p(rstd), p(out), M, H, dt(Gy), stream()), "lrp_layernorm_bwd_plain")
    return out


def softmax_fwd(x, inv_temp=1.0):
    x = _c(x)
    n = x.shape[-1]
    out = torch.empty_like(x)
    check(lib.lrp_softmax_fwd(p(x), p(out), x.numel() // n, n, inv_temp, dt(x), stream()), "lrp_softmax_fwd")
    return out


def softmax_rule_bwd(x, pr, Rp, inv_temp=1.0):
    x, pr, Rp = _c(x), _c(pr), _c(Rp)
    n = x.shape[-1]
    Rx = torch.empty_like(x)
    same(x, pr, Rp)
    check(lib.lrp_softmax_rule_bwd(p(x), p(pr), p(Rp), p(Rx), x.numel() // n, n, inv_temp, dt(x), stream()),
          "lrp_softmax_rule_bwd")
    return Rx


def readout(emb, G, out=None):
    M, H = emb.shape
    out = torch.empty(M, device=emb.device, dtype=torch.float32) if out is None else out
    same(emb, G)
    f32(out)
    check(lib.lrp_readout(p(emb), p(G), p(out), M, H, dt(emb), stream()), "lrp_readout")
    return out


def argmax_rows(logits):
    B, V = logits.shape
    idx = torch.empty(B, device=logits.device, dtype=torch.int32)
    val = torch.empty(B, device=logits.device, dtype=torch.float32)
    check(lib.lrp_argmax_rows(p(logits), p(idx), p(val), B, V, logits.stride(0), stream()), "lrp_argmax_rows")
    return idx, val


def head_seed(W_lm, logits, idx, w_norm, rstd_last, out, w_offset=0.0, eps_lin=0.0):
    B, V = logits.shape
    H = W_lm.shape[1]
    same(W_lm, out)
    f32(logits, rstd_last)
    w_norm = aux(w_norm, W_lm, H)
    if idx.dtype != torch.int32:
        raise TypeError("head_seed: idx must be int32")
    check(lib.lrp_head_seed(p(W_lm), p(logits), p(idx), p(w_norm), p(rstd_last), p(out), B, V, H, logits.stride(0), w_offset,
                            eps_lin, dt(W_lm), stream()), "lrp_head_seed")
    return out


SMALLM_MAX = 16          # rows the W-streaming small-M kernels serve (above: the skinny split-K path, then the MFMA GEMM)


def _smallm_ws(M, N, K, ref):
    """fp32 scratch of the small-M kernels: the per-(device, stream) workspace (capture-safe, see workspace())"""
    return workspace(4 * lib.lrp_linear_smallm_ws(M, N, K, dt(ref)), ref)


def smallm_ok(M, W, x=None):
    """can the W-streaming small-M kernels serve this call?  (M <= 16 rows, W contiguous [N,K] with K a multiple of 16 bytes,
    16-byte aligned operands -- anything else goes to the GEMM path, which pads)"""
    e = epc(W)
    ok = (0 < M <= SMALLM_MAX and W.dim() == 2 and W.is_contiguous() and W.shape[1] % e == 0 and W.data_ptr() % 16 == 0
          and W.dtype in _DT)
    if ok and x is not None:
        ok = x.dim() == 2 and x.stride(1) == 1 and x.stride(0) % e == 0 and x.data_ptr() % 16 == 0 and x.dtype == W.dtype
    return ok


def linear_smallm_fwd(x, W, bias=None, out=None, out_dtype=None):
    """z [M,N] = x [M,K] @ W[N,K]^T (+ bias), M <= 16, W streamed once (no MFMA tile padding to 128 rows)"""
    M, K = x.shape
    N = W.shape[0]
    same(x, W)
    odt = out_dtype or x.dtype
    if out is None:
        out = torch.empty(M, N, device=x.device, dtype=odt)
    check(lib.lrp_linear_smallm_fwd(p(x), p(W), p(aux(bias, x, N)), p(out), p(_smallm_ws(M, N, K, x)), M, N, K, x.stride(0),
                                    out.stride(0), dt(x), _DT[out.dtype], stream()), "lrp_linear_smallm_fwd")
    return out


def linear_smallm_dgrad(g, W, z=None, x=None, eps=0.0, relevance_in=False, relevance_out=False, out=None, out_dtype=None):
    """out [M,K] = s @ W[N,K] (* x) with s = g, g*z/(z+eps) or g/(z+eps): the eps-rule backward of a Linear for M <= 16 rows
    from the STORED weight layout (no W^T copy), W streamed once"""
    M, N = g.shape
    K = W.shape[1]
    same(W, g, z, x)
    odt = out_dtype or W.dtype
    if out is None:
        out = torch.empty(M, K, device=W.device, dtype=odt)
    if x is not None and not x.is_contiguous():
        x = x.contiguous()
    check(lib.lrp_linear_smallm_dgrad(p(g), p(z), p(W), p(x), p(out), p(_smallm_ws(M, N, K, W)), M, N, K, g.stride(0),
                                      z.stride(0) if z is not None else 0, eps, int(relevance_in), int(relevance_out), dt(W),
                                      _DT[out.dtype], stream()), "lrp_linear_smallm_dgrad")
    return out


TAIL_SPLIT = True        # module attribute (A/B measurements): False = the round-3 handling of 257 ... 511-tile GEMMs


def tail_split_cols(M, Nout, Kc):
    """GEMMs whose tile count (256 x 256 tiles) is not a whole number of rounds of the 256 CUs and whose LAST round is at most half full
    (Gemma-3-4B: 8192 x 2560 = 320 tiles = 1.25 rounds; SigLIP: 16384 x 4352 = 1088 tiles = 4.25 rounds: the last round keeps a quarter of the
    chip busy for a whole tile time).  -> the number of leading output columns that make whole rounds; the remaining tile columns (<= 128
    tiles) are issued as a second problem through the split-K path, which spreads them over all CUs (K split so that tail tiles x splits =
    256).  Two launches of n + ~0.25 tile times instead of n + 1 (or, for 257 ... 384 tiles and K >= 8192, instead of splitting the WHOLE problem
    in two with fp32 slabs of the whole output).  None when the shape does not qualify (a K loop too short to split: the tail's slab round trip
    would cost more than the partial round it replaces)."""
    if not TAIL_SPLIT:
        return None
    tm, tn = (M + 255) // 256, (Nout + 255) // 256
    if tm * tn <= 256 or tm > 256 or 256 % tm:
        return None
    cpr = 256 // tm                                         # tile columns per full round
    tail_cols = tn % cpr
    tail_tiles = tm * tail_cols
    if tail_cols == 0 or tail_tiles > 128 or Kc // 64 < 8 * (256 // tail_tiles):
        return None
    return (tn - tail_cols) * 256


STREAM_FWD = True        # module attribute (A/B measurements): False sends every M <= 256 forward to the split-K skinny path


STREAM_FWD_SPLITS = True    # module attribute (A/B measurements): False = narrow weights (N < 192 * 64) stay on the split-K skinny path


def linear_stream_ok(x2, W):
    """does the one-launch weight-streaming forward (lrp_linear_stream_fwd: narrow-N, full-K, no split-K slabs) serve z = x2 W^T?  bf16,
    M <= 256 rows, K % 512 == 0, contiguous 16-byte aligned rows, and ceil(N / 64) workgroups that fill the chip"""
    if not STREAM_FWD or x2.dtype != torch.bfloat16 or W.dtype != torch.bfloat16 or x2.dim() != 2 or W.dim() != 2:
        return False
    if x2.stride(1) != 1 or W.stride(1) != 1 or x2.data_ptr() % 16 or W.data_ptr() % 16:
        return False
    if not lib.lrp_linear_stream_ok(x2.shape[0], W.shape[0], x2.shape[1], x2.stride(0), W.stride(0)):
        return False
    return bool(STREAM_FWD_SPLITS or lib.lrp_linear_stream_fwd_splits(x2.shape[0], W.shape[0], x2.shape[1]) == 1)


def linear_stream_fwd(x2, W, bias=None, out=None, out_dtype=None):
    """z[M,N] = x2[M,K] @ W[N,K]^T (+ bias), M <= 256, in ONE launch: every workgroup streams 64 rows of W over the whole K range"""
    M, K = x2.shape
    N = W.shape[0]
    same(x2, W)
    if out is None:
        out = torch.empty(M, N, device=x2.device, dtype=out_dtype or x2.dtype)
    need = lib.lrp_linear_stream_fwd_ws(M, N, K) if STREAM_FWD_SPLITS else 0      # narrow weights: K splits, slabs summed in-kernel (tickets)
    ws = workspace(need, x2) if need else None
    ntk = lib.lrp_linear_stream_fwd_tickets(M, N, K) if need else 0
    check(lib.lrp_linear_stream_fwd_tk(p(x2), p(W), p(aux(bias, x2, N)), p(out), M, N, K, x2.stride(0), W.stride(0), out.stride(0), dt(x2),
                                       _DT[out.dtype], p(ws), p(tickets(ntk, x2) if ntk else None), stream()), "lrp_linear_stream_fwd_tk")
    return out


def linear_fwd(x2, W, bias=None, out=None, out_dtype=None):
    """z[M,N] = x2[M,K] @ W[N,K]^T (+ bias) on the kernel that fits M (ref: lxt/explicit/functional.py:351):
       M <= 256 rows, bf16, K % 512 == 0, N >= 192 * 64 : one-launch weight-streaming kernel (narrow N, full K: no slabs, no second launch)
       M <= 256 rows, bf16, K % 64 == 0 : split-K skinny path of the ping-pong GEMM (W streamed once by all CUs)
       M <= 16 otherwise                : W-streaming small-M kernels (fp32, odd K)
       else                             : the MFMA GEMM (lrp_gemm_nt)"""
    M, K = x2.shape
    N = W.shape[0]
    odt = out_dtype or (out.dtype if out is not None else x2.dtype)
    if M <= SKINNY_MAX and linear_stream_ok(x2, W) and (out is None or (out.stride(1) == 1 and out.dtype in _DT)):
        return linear_stream_fwd(x2, W, bias, out=out, out_dtype=odt)
    if gemm_nn_ok(x2, W) and M > SKINNY_MAX:
        main = tail_split_cols(M, N, K)
        if main is not None and (out is None or out.stride(1) == 1):
            if out is None:
                out = torch.empty(M, N, device=x2.device, dtype=odt)
            bias = aux(bias, x2, N)
            gemm_nt_2d(x2, W[:main], out[:, :main], None if bias is None else bias[:main])                  # one full round of the chip
            gemm_skinny(x2, W[main:], out[:, main:], nn=False, bias=None if bias is None else bias[main:])    # the tail, K-split over all CUs
            return out
    if splitk_ok(M, N, K) and gemm_nn_ok(x2, W):
        if out is None:
            out = torch.empty(M, N, device=x2.device, dtype=odt)
        return gemm_skinny(x2, W, out, nn=False, bias=bias)
    if smallm_ok(M, W, x2):
        return linear_smallm_fwd(x2, W, bias, out=out, out_dtype=odt)
    if out is not None and x2.stride(1) == 1 and W.stride(1) == 1 and x2.dtype == W.dtype and x2.stride(0) % epc(x2) == 0:
        return gemm_nt_2d(x2, W, out, bias)
    z = gemm_nt(x2, W, bias, out_dtype=odt)
    if out is not None:
        out.copy_(z)
        return out
    return z


STREAM_DGRAD = True       # module attribute (A/B measurements): False keeps every small-M dgrad on the small-M / split-K skinny kernels


def linear_stream_dgrad_ok(s2, W):
    """is lrp_linear_stream_dgrad (64-column workgroups, wave-private LDS rings, transpose-read W operand) the kernel for c = s2 W?  bf16,
    M <= 32, N % 128 == 0, Kout % 64 == 0, contiguous 16-byte aligned rows, and a split count that fills the chip with <= 320 workgroups"""
    if not STREAM_DGRAD or s2.dtype != torch.bfloat16 or W.dtype != torch.bfloat16 or s2.dim() != 2 or W.dim() != 2:
        return False
    if s2.stride(1) != 1 or W.stride(1) != 1 or s2.data_ptr() % 16 or W.data_ptr() % 16:
        return False
    return bool(lib.lrp_linear_stream_dgrad_ok(s2.shape[0], W.shape[0], W.shape[1], s2.stride(0), W.stride(0)))


def linear_stream_dgrad(s2, W, out=None, out_dtype=None, z=None, eps=0.0, relevance_in=False):
    """c[M,Kout] = s'[M,N] @ W[N,Kout] for M <= 64 from the stored weight, W streamed once; s' = s2, or with z (the Linear's forward output)
    the eps-rule's stabilised operand formed inside the kernel: s2 z / (z + eps), or s2 / (z + eps) when s2 is a relevance (relevance_in)"""
    M, N = s2.shape
    Kout = W.shape[1]
    same(s2, W, z)
    if z is not None and (z.stride(1) != 1 or tuple(z.shape) != (M, N)):
        raise ValueError("linear_stream_dgrad: z must be [M, N] with contiguous rows")
    if z is not None and eps == 0.0:
        if relevance_in:
            raise ValueError("linear_stream_dgrad: a relevance operand needs eps != 0 (s / (z + 0) has no defined value at z = 0)")
        z = None                    # eps = 0: the stabiliser z / (z + 0) is exactly 1 (lxt.efficient) -- never formed as g * z * rcp(z)
    if out is None:
        out = torch.empty(M, Kout, device=s2.device, dtype=out_dtype or s2.dtype)
    need = lib.lrp_linear_stream_dgrad_ws(M, N, Kout)
    ws = workspace(need, s2) if need else None
    ntk = lib.lrp_linear_stream_dgrad_tickets(M, N, Kout) if STREAM_DGRAD_INKERNEL_REDUCE else 0
    check(lib.lrp_linear_stream_dgrad_tk(p(s2), p(z), p(W), p(out), M, N, Kout, s2.stride(0), z.stride(0) if z is not None else 0, W.stride(0),
                                         out.stride(0), eps, int(relevance_in), dt(s2), _DT[out.dtype], p(ws), p(tickets(ntk, s2) if ntk else None),
                                         stream()), "lrp_linear_stream_dgrad_tk")
    return out


def linear_dgrad(s2, W, out=None, out_dtype=None):
    """c[M,K] = s2[M,N] @ W[N,K]: the redistribution half of the Linear eps-rule (ref: lxt/explicit/functional.py:355-364) from the
    STORED weight layout:
       M <= 32, bf16, N % 128 == 0, Kout % 64 == 0            : lrp_linear_stream_dgrad (64-column workgroups, wave-private LDS rings; round 5: also
                                                               for M <= 2 -- with the in-kernel slab reduction it is 1.5 ... 8 us ahead of the
                                                               lane-local kernel there, tools/dgrad_m12.py)
       M <= 2, or M <= 16 where the NN kernel does not apply : W-streaming small-M dgrad
       M <= 256, bf16, N % 64 == 0                           : split-K skinny path, NN form
       bf16 problems of >= 190 tiles of 256 x 256            : lrp_gemm_nn (no W^T copy)
       everything else (fp32 parity path, odd shapes)        : lrp_gemm_nt on a W^T copy cached on the weight (ops.weight_t)"""
    M, N = s2.shape
    K = W.shape[1]
    odt = out_dtype or (out.dtype if out is not None else W.dtype)
    if M <= 32 and linear_stream_dgrad_ok(s2, W) and (out is None or (out.stride(1) == 1 and out.dtype in _DT)):
        return linear_stream_dgrad(s2, W, out=out, out_dtype=odt)
    nn = gemm_nn_ok(s2, W)
    if (M <= 2 or (M <= SMALLM_MAX and not nn)) and N >= 16 and smallm_ok(M, W) and s2.stride(1) == 1 and s2.dtype == W.dtype:
        return linear_smallm_dgrad(s2, W, out=out, out_dtype=odt)
    if nn and M > SKINNY_MAX:
        main = tail_split_cols(M, K, N)
        if main is not None and (out is None or out.stride(1) == 1) and W.data_ptr() % 16 == 0:
            if out is None:
                out = torch.empty(M, K, device=s2.device, dtype=odt)
            gemm_nn_2d(s2, W[:, :main], out[:, :main])                           # one full round of the chip
            gemm_skinny(s2, W[:, main:], out[:, main:], nn=True)                  # the tail (<= 128 tiles), contraction split over all CUs
            return out
    if nn:
        split = splitk_ok(M, K, N)
        # the NN form exists only in the 256 x 256 ping-pong kernel: problems it cannot fill (fewer than 190 tiles and no split-K: BERT-sized
        # weights at a few thousand rows) run the 128 x 128 / 64 x 64 NT kernels on a cached W^T instead, as in round 2
        if split or ((M + 255) // 256) * ((K + 255) // 256) >= 190:
            if out is None:
                out = torch.empty(M, K, device=s2.device, dtype=odt)
            return gemm_skinny(s2, W, out, nn=True) if split else gemm_nn_2d(s2, W, out)
    wt = weight_t(W)
    if out is not None and s2.stride(1) == 1 and s2.dtype == wt.dtype and s2.stride(0) % epc(s2) == 0:
        return gemm_nt_2d(s2, wt, out)
    c = gemm_nt(s2, wt, out_dtype=odt)
    if out is not None:
        out.copy_(c)
        return out
    return c


def clear_weight_cache(*weights):
    """drop the cached W^T copies (fp32 / odd-shape dgrad path) of the given weight tensors -- call after writing to a weight through
    `.data` (optimizers, adapter / quantisation loaders), which does not bump `_version` and would leave a stale W^T behind"""
    for w in weights:
        _WT.pop(w.untyped_storage()._cdata, None)


# -------------------------------------------------------------------------------------- attention
def pad_to(n, m):
    return (n + m - 1) // m * m


def transpose_heads(x, B, S, H, d, out=None):
    """x: [B*S, >= H*d] 2-D view -> [B, H, d, ldt] with ldt = S padded to 64 (pad columns zero)"""
    ldt = pad_to(S, 64)
    if out is None:
        out = torch.zeros(B, H, d, ldt, device=x.device, dtype=x.dtype)
    check(lib.lrp_transpose_heads(p(x), p(out), B, S, H, d, x.stride(0), out.stride(2), dt(x), stream()), "lrp_transpose_heads")
    return out


def _iv(row_iv, B, S):
    """(row_lo, row_hi) int32 [B, S] device tensors -> two raw pointers (0, 0 when no intervals are given)"""
    if row_iv is None:
        return 0, 0
    lo, hi = row_iv
    for t in (lo, hi):
        if t.dtype != torch.int32 or not t.is_contiguous() or t.numel() != B * S or not t.is_cuda:
            raise ValueError("row intervals must be contiguous int32 CUDA tensors of B*S elements")
    return lo.data_ptr(), hi.data_ptr()


def attn_needs_transposed(t, d):
    """do the attention kernels serving t's dtype and head dim d read head-transposed "_t" copies (transpose_heads)?
    bf16 / d = 128 does not: its kernels take every operand from the token-major tensors (pass None for the _t operands)"""
    return bool(lib.lrp_attn_needs_transposed(dt(t), d))


def attn_fwd(q, k, v, v_t, o, lse, B, S, Hq, Hkv, d, scale, causal=True, window=0, q_begin=0, row_iv=None):
    same(q, k, v, v_t, o)
    f32(lse)
    check(lib.lrp_attn_fwd(p(q), p(k), p(v), p(v_t), p(o), p(lse), B, S, Hq, Hkv, d, q.stride(0), k.stride(0), v.stride(0),
                           v_t.stride(2) if v_t is not None else 0, o.stride(0), scale, int(causal), window, q_begin,
                           *_iv(row_iv, B, S), dt(q), stream()), "lrp_attn_fwd")
    return o, lse


def attn_bwd_prep(Go, o, Gho, D, B, S, Hq, d, eps_pv, factor=0.5):
    same(Go, o, Gho)
    f32(D)
    check(lib.lrp_attn_bwd_prep(p(Go), p(o), p(Gho), p(D), B, S, Hq, d, Go.stride(0), o.stride(0), Gho.stride(0), eps_pv,
                                factor, dt(Go), stream()), "lrp_attn_bwd_prep")
    return Gho, D


def attn_bwd_dq(q, k, v, k_t, Gho, lse, D, dq, B, S, Hq, Hkv, d, scale, eps_mask, eps_qk, causal=True, window=0, q_begin=0,
                row_iv=None):
    same(q, k, v, k_t, Gho, dq)
    f32(lse, D)
    check(lib.lrp_attn_bwd_dq(p(q), p(k), p(v), p(k_t), p(Gho), p(lse), p(D), p(dq), B, S, Hq, Hkv, d, q.stride(0),
                              k.stride(0), v.stride(0), k_t.stride(2) if k_t is not None else 0, Gho.stride(0), dq.stride(0),
                              scale, eps_mask, eps_qk,
                              int(causal), window, q_begin, *_iv(row_iv, B, S), dt(q), stream()), "lrp_attn_bwd_dq")
    return dq


PREP_FUSION = True       # module attribute: False = the stand-alone lrp_attn_bwd_prep pass (A/B measurements)


def attn_dq_d_ok(dtype, d):
    """lrp_attn_bwd_dq_d serves (dtype, d): the dQ kernel forms D = rowsum(Gho (*) o) itself (no attn_bwd_prep pass)"""
    return bool(PREP_FUSION and dtype in _DT and lib.lrp_attn_bwd_dq_d_ok(_DT[dtype], d))


def attn_bwd_dq_d(q, k, v, Gho, o, lse, D, dq, B, S, Hq, Hkv, d, scale, causal=True, window=0, row_iv=None, rope=None):
    """dQ of the lxt.efficient placement with D[b, h, s] = sum_d Gho o formed in the kernel and WRITTEN to D (for attn_bwd_dkv); rope = (cos, sin)
    fp32 [>= S, d]: RoPE's backward applied to dQ on its way out (then dq is the gradient w.r.t. the UN-rotated q)"""
    same(q, k, v, Gho, o, dq)
    f32(lse, D)
    cs, sn = rope if rope is not None else (None, None)
    f32(cs, sn)
    check(lib.lrp_attn_bwd_dq_d(p(q), p(k), p(v), p(Gho), p(o), p(lse), p(D), p(dq), B, S, Hq, Hkv, d, q.stride(0), k.stride(0), v.stride(0),
                                Gho.stride(0), o.stride(0), dq.stride(0), scale, int(causal), window, *_iv(row_iv, B, S), p(cs), p(sn), dt(q),
                                stream()), "lrp_attn_bwd_dq_d")
    return dq


ROPE_BWD_FUSION = True   # module attribute (A/B): False = the stand-alone lrp_rope_bwd pass


def gqa_reduce_rope(x, out, rows, seq, Hkv, rep, d, cos, sin):
    """out[r, hk, :] = RoPE^T(sum_j x[r, hk rep + j, :]) -- the GQA group sum of dK and RoPE's backward in one pass"""
    f32(cos, sin)
    check(lib.lrp_gqa_reduce_rope(p(x), p(out), rows, seq, Hkv, rep, d, x.stride(0), out.stride(0), p(cos), p(sin), dt(x), stream()),
          "lrp_gqa_reduce_rope")
    return out


def attn_bwd_dkv(q, k, v, q_t, Gho, Gho_t, lse, D, dk_h, dv_h, B, S, Hq, Hkv, d, scale, eps_mask, eps_qk, causal=True,
                 window=0, q_begin=0, row_iv=None):
    same(q, k, v, q_t, Gho, Gho_t, dk_h, dv_h)
    f32(lse, D)
    check(lib.lrp_attn_bwd_dkv(p(q), p(k), p(v), p(q_t), p(Gho), p(Gho_t), p(lse), p(D), p(dk_h), p(dv_h), B, S, Hq, Hkv, d,
                               q.stride(0), k.stride(0), v.stride(0), q_t.stride(2) if q_t is not None else 0, Gho.stride(0),
                               dk_h.stride(0),
                               dv_h.stride(0), scale, eps_mask, eps_qk, int(causal), window, q_begin, *_iv(row_iv, B, S), dt(q), stream()),
          "lrp_attn_bwd_dkv")
    return dk_h, dv_h


def gqa_reduce(x, out, rows, Hkv, rep, d):
    check(lib.lrp_gqa_reduce(p(x), p(out), rows, Hkv, rep, d, x.stride(0), out.stride(0), dt(x), stream()), "lrp_gqa_reduce")
    return out
