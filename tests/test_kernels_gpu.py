"""GPU parity of each HIP kernel (through the C ABI) against the oracle's closed forms /
a plain PyTorch fp64 restatement of the same op, on seeded inputs.  fp32 kernels: normalised
max error <= 2e-5 (fp32 MFMA is an exact fma chain; the slack is accumulation order);
bf16 kernels: <= 2e-2 (bf16 storage rounding of inputs/outputs)."""
import math

import pytest
import torch
import torch.nn.functional as F

from tests.util import nmax, load, t

pytestmark = pytest.mark.gpu

TOL = {torch.float32: 2e-5, torch.bfloat16: 2e-2}
DTYPES = [torch.float32, torch.bfloat16]


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    import lxt_amd.ops as o
    return o


def rnd(*shape, dtype=torch.float32, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).cuda()


def f64(x):
    return x.double()


# ----------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(1, 768, 768), (128, 128, 64), (200, 136, 264), (257, 129, 72), (2048, 512, 1024), (64, 1000, 4096),
                                   (128, 768, 768), (128, 3072, 768), (128, 768, 3072), (100, 70, 128), (300, 200, 256), (33, 31, 64)])
def test_gemm_nt(ops, dtype, M, N, K):
    a, b = rnd(M, K, dtype=dtype, seed=1), rnd(N, K, dtype=dtype, seed=2, scale=K ** -0.5)
    bias = rnd(N, dtype=dtype, seed=3)
    out = ops.gemm_nt(a, b, bias)
    ref = f64(a) @ f64(b).T + f64(bias)
    assert out.shape == (M, N) and nmax(out, ref) < TOL[dtype]
    # transpose detection: asymmetric operands, A = shifted identity
    if M == N == 128:
        eye = torch.zeros(M, K, dtype=dtype, device="cuda")
        eye[torch.arange(K), torch.arange(K)] = 1
        out = ops.gemm_nt(eye, b)
        assert nmax(out[:K], f64(b).T[:K]) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(3301, 3700, 192), (4096, 3100, 128), (3600, 3584, 64), (3333, 3841, 320)])
def test_gemm_nt_big_tile_path(ops, dtype, M, N, K):
    """>= 190 tiles of 256x256: the 8-wave ping-pong kernel (ragged M / N: rows past the operand read as zero, odd and minimal
    numbers of K tiles; one K tile falls back to the plain form), bias, row strides larger than K"""
    g = torch.Generator().manual_seed(5)
    a = torch.randn(M, K + 64, generator=g).to(dtype).cuda()[:, :K]
    b = (torch.randn(N, K + 64, generator=g) * K ** -0.5).to(dtype).cuda()[:, :K]
    bias = torch.randn(N, generator=g).to(dtype).cuda()
    out = torch.empty(M, N + 8, dtype=dtype, device="cuda")[:, :N]
    ops.gemm_nt_2d(a, b, out, bias)
    ref = f64(a) @ f64(b).T + f64(bias)
    assert nmax(out, ref) < TOL[dtype]
    # every tile, not only the largest entries: block-wise normalised error
    blk = (out.double() - ref).abs().reshape(-1)[: (M * N // 4096) * 4096].reshape(-1, 4096).max(1).values
    assert float(blk.max()) < 4 * TOL[dtype] * float(ref.abs().max())


@pytest.mark.parametrize("M,N,K,odt", [(4096, 4096, 4096, torch.bfloat16), (3301, 3700, 192, torch.bfloat16), (4096, 3100, 128, torch.float32),
                                       (3333, 3841, 320, torch.bfloat16), (2048, 6144, 1024, torch.float32), (300, 520, 256, torch.bfloat16)])
def test_gemm_nn_from_stored_weight(ops, M, N, K, odt):
    """lrp_gemm_nn: C = A[M,K] @ W[K,N] with W row-major -- the eps-rule redistribution c = s W from the STORED weight [out,in] (no W^T
    copy; ref lxt/explicit/functional.py:355-364): the transposed MFMA operand is gathered from a row-major LDS tile by
    ds_read_b64_tr_b16.  Ragged M / N / odd K-tile counts, row pitches larger than the logical width, bf16 and fp32 outputs; every tile
    is checked (block-wise normalised error), against fp64 on the same bf16 operands."""
    g = torch.Generator().manual_seed(11)
    a = torch.randn(M, K + 64, generator=g).bfloat16().cuda()[:, :K]
    w = (torch.randn(K, (N + 71) // 8 * 8, generator=g) * K ** -0.5).bfloat16().cuda()[:, :N]
    out = torch.full((M, N + 8), float("nan"), dtype=odt, device="cuda")[:, :N]
    assert ops.gemm_nn_ok(a, w)
    ops.gemm_nn_2d(a, w, out)
    ref = f64(a) @ f64(w)
    tol = TOL[torch.bfloat16] if odt == torch.bfloat16 else 2e-5
    assert not torch.isnan(out).any() and nmax(out, ref) < tol
    blk = (out.double() - ref).abs().reshape(-1)[: (M * N // 4096) * 4096].reshape(-1, 4096).max(1).values
    assert float(blk.max()) < 4 * tol * float(ref.abs().max())


@pytest.mark.parametrize("nn", [False, True])
@pytest.mark.parametrize("M", [1, 4, 16, 17, 33, 100, 160, 256])
def test_gemm_skinny_split_k(ops, M, nn):
    """lrp_gemm_skinny: 1 <= M <= 256 rows, split-K over the CUs + fp32 slab reduction (+ bias, cast); forward z = x W^T (nn = 0) and
    redistribution c = s W from the stored weight (nn = 1); vs fp64 on the same bf16 operands"""
    g = torch.Generator().manual_seed(M + 7 * nn)
    N, K = (1536, 1024) if M % 2 else (1000, 2048 + 128)
    a = torch.randn(M, K, generator=g).bfloat16().cuda()
    if nn:
        b = (torch.randn(K, (N + 15) // 8 * 8, generator=g) * K ** -0.5).bfloat16().cuda()[:, :N]
        ref = f64(a) @ f64(b)
    else:
        b = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16().cuda()
        ref = f64(a) @ f64(b).T
    bias = torch.randn(N, generator=g).bfloat16().cuda() if M % 3 == 0 else None
    if bias is not None:
        ref = ref + f64(bias)
    for odt, tol in ((torch.float32, 2e-5), (torch.bfloat16, TOL[torch.bfloat16])):
        out = torch.full((M, N), float("nan"), dtype=odt, device="cuda")
        ops.gemm_skinny(a, b, out, nn=nn, bias=bias)
        assert not torch.isnan(out).any() and nmax(out, ref) < tol, (M, nn, odt)


@pytest.mark.parametrize("M", [1, 5, 16, 17, 33, 64, 100, 128])
def test_linear_stream_fwd(ops, M):
    """lrp_linear_stream_fwd (round 4: the Linear forward in its HBM-bound regime as ONE launch -- narrow N, full K, W streamed once through
    wave-private LDS rings, x through a shared LDS ring, 16-row blocks past M not multiplied): every M bucket (2 / 4 / 8 row blocks), ragged N (not a
    multiple of 64), K = 512 (prologue + peeled last block only) and 4096 (steady state), strided x, bias, bf16 and fp32 outputs; vs fp64 on
    the same bf16 operands, and bit-identical row by row to the call with fewer rows (a row's result does not depend on its neighbours)"""
    g = torch.Generator().manual_seed(100 + M)
    for (N, K) in ((12352 + 24, 512), (14336, 4096), (12288 + 2, 1536)):
        xs = torch.randn(M, K + 64, generator=g).bfloat16().cuda()
        x = xs[:, :K]                                                        # row pitch K + 64
        W = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16().cuda()
        bias = torch.randn(N, generator=g).bfloat16().cuda() if (M + K // 512) % 2 else None
        assert ops.linear_stream_ok(x, W)
        ref = f64(x) @ f64(W).T + (f64(bias) if bias is not None else 0.0)
        for odt, tol in ((torch.float32, 2e-5), (torch.bfloat16, TOL[torch.bfloat16])):
            out = torch.full((M, N), float("nan"), dtype=odt, device="cuda")
            ops.linear_stream_fwd(x, W, bias, out=out)
            assert not torch.isnan(out).any() and nmax(out, ref) < tol, (M, N, K, odt, nmax(out, ref))
        # through the dispatcher, and row independence
        z = ops.linear_fwd(x, W, bias)
        assert torch.equal(z, out)
        if M > 1:
            z1 = ops.linear_stream_fwd(x[: M - 1], W, bias)
            assert torch.equal(z1, z[: M - 1])
    # shapes the kernel refuses go to the split-K path (more than 128 rows; K not a multiple of 512; too few 64-row workgroups)
    assert not ops.linear_stream_ok(torch.empty(160, 4096, dtype=torch.bfloat16, device="cuda"), torch.empty(14336, 4096, dtype=torch.bfloat16, device="cuda"))
    assert not ops.linear_stream_ok(torch.empty(4, 4096 + 64, dtype=torch.bfloat16, device="cuda"), torch.empty(14336, 4096 + 64, dtype=torch.bfloat16, device="cuda"))
    assert not ops.linear_stream_ok(torch.empty(4, 4096, dtype=torch.bfloat16, device="cuda"), torch.empty(1024, 4096, dtype=torch.bfloat16, device="cuda"))
    # round 5 -- narrow weights: the same kernel with K splits, the fp32 slabs summed in-kernel by the last arriver of a 64-row block (still one
    # launch; lrp_linear_stream_fwd_tk).  [4096, 14336] (the down projection: 4 splits of 56 K tiles), [4096, 4096] (4 x 16), [8192, 2048] (2 x 16);
    # repeated calls (the ticket words must come back to zero), bias, both output dtypes, row independence, and the split-K skinny path beside it
    for (N, K) in ((4096, 14336), (4096, 4096), (8192, 2048)):
        x = torch.randn(M, K, generator=g).bfloat16().cuda()
        W = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16().cuda()
        bias = torch.randn(N, generator=g).bfloat16().cuda() if (M + N // 4096) % 2 else None
        assert ops.linear_stream_ok(x, W) and ops.lib.lrp_linear_stream_fwd_splits(M, N, K) in (2, 4)
        ref = f64(x) @ f64(W).T + (f64(bias) if bias is not None else 0.0)
        for odt, tol in ((torch.float32, 2e-5), (torch.bfloat16, TOL[torch.bfloat16])):
            outs = []
            for rep in range(3):
                out = torch.full((M, N), float("nan"), dtype=odt, device="cuda")
                ops.linear_stream_fwd(x, W, bias, out=out)
                outs.append(out)
            assert not torch.isnan(outs[0]).any() and nmax(outs[0], ref) < tol, (M, N, K, odt, nmax(outs[0], ref))
            assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
        z = ops.linear_fwd(x, W, bias)
        assert torch.equal(z, outs[0])
        if M > 1:
            assert torch.equal(ops.linear_stream_fwd(x[: M - 1], W, bias), z[: M - 1])
        keep = ops.STREAM_FWD_SPLITS
        ops.STREAM_FWD_SPLITS = False
        try:
            assert not ops.linear_stream_ok(x, W)
            z2 = ops.linear_fwd(x, W, bias)                                    # split-K skinny path of the ping-pong GEMM: two launches
        finally:
            ops.STREAM_FWD_SPLITS = keep
        assert nmax(z2, z) < 1e-2


@pytest.mark.parametrize("M", [1, 3, 16, 17, 33, 64])
def test_linear_stream_dgrad(ops, M):
    """lrp_linear_stream_dgrad (round 4): c = s W from the stored weight for M <= 64 -- 64-column workgroups, contraction range split over
    workgroups (fp32 slabs + ordered reduce) or not (direct store), wave-private LDS rings, transpose-read W operand: both M buckets, shapes with
    4 / 1 / 3 splits, a strided s, bf16 and fp32 outputs; vs fp64 on the same bf16 operands; deterministic (two runs bit-equal) and row-independent"""
    g = torch.Generator().manual_seed(200 + M)
    for (N, Kout) in ((14336, 4096), (4096, 14336), (1536, 4800), (768 * 128, 4096)):
        ss = torch.randn(M, N + 64, generator=g).bfloat16().cuda()
        s_ = ss[:, :N]
        W = (torch.randn(N, Kout, generator=g) * N ** -0.5).bfloat16().cuda()
        assert ops.linear_stream_dgrad_ok(s_, W) == (M <= 32), (N, Kout)     # dispatch policy: M <= 32 (and <= 320 workgroups: true for these shapes)
        ref = f64(s_) @ f64(W)
        for odt, tol in ((torch.float32, 2e-5), (torch.bfloat16, TOL[torch.bfloat16])):
            out = torch.full((M, Kout), float("nan"), dtype=odt, device="cuda")
            ops.linear_stream_dgrad(s_, W, out=out)
            assert not torch.isnan(out).any() and nmax(out, ref) < tol, (M, N, Kout, odt, nmax(out, ref))
            out2 = torch.empty_like(out)
            ops.linear_stream_dgrad(s_, W, out=out2)
            assert torch.equal(out, out2)
            # contraction splits (Kout = 4096: 4): the in-kernel last-arriver reduction (round 5) sums the fp32 slabs in slab order like the
            # reduce launch it replaces -- same bits, whichever workgroup arrives last, call after call
            ops.STREAM_DGRAD_INKERNEL_REDUCE = False
            try:
                two_launch = ops.linear_stream_dgrad(s_, W, out=torch.empty_like(out))
            finally:
                ops.STREAM_DGRAD_INKERNEL_REDUCE = True
            assert torch.equal(out, two_launch)
            for _ in range(20):
                out2.fill_(float("nan"))
                ops.linear_stream_dgrad(s_, W, out=out2)
                assert torch.equal(out, out2)
        if M > 1:
            assert torch.equal(ops.linear_stream_dgrad(s_[: M - 1], W), out[: M - 1])
        if M <= 32:
            # the eps-rule's stabiliser formed inside the kernel (gradient and relevance forms) vs fp64 on the same bf16 operands; the
            # operand s' is rounded to bf16 inside the kernel exactly where the unfused pair (lrp_eps_scale + dgrad) stores it
            zz = (torch.randn(M, N, generator=g) * 0.5).bfloat16().cuda()
            for rel_in, eps in ((False, 1e-6), (True, 1e-6), (True, 1e-2)):
                f = (1.0 if rel_in else f64(zz)) / (f64(zz) + eps)
                sref = (f64(s_) * f).bfloat16()                                  # the rounding point of the operand
                o32 = ops.linear_stream_dgrad(s_, W, z=zz, eps=eps, relevance_in=rel_in, out_dtype=torch.float32)
                ref_s = f64(sref) @ f64(W)
                assert nmax(o32, ref_s) < 2e-2, (M, N, Kout, rel_in, eps, nmax(o32, ref_s))
                unf = ops.linear_stream_dgrad(ops.eps_scale(s_.contiguous(), zz, 1.0, eps, relevance=rel_in), W, out_dtype=torch.float32)
                assert nmax(o32, unf) < 2e-2
        if 2 < M <= 32:
            assert torch.equal(ops.linear_dgrad(s_, W), out)                 # the dispatcher takes this kernel
    assert not ops.linear_stream_dgrad_ok(torch.empty(33, 14336, dtype=torch.bfloat16, device="cuda"), torch.empty(14336, 4096, dtype=torch.bfloat16, device="cuda"))
    assert not ops.linear_stream_dgrad_ok(torch.empty(8, 14336 + 64, dtype=torch.bfloat16, device="cuda"), torch.empty(14336 + 64, 4096, dtype=torch.bfloat16, device="cuda"))
    # the 128256-row LM head would need 6 contraction splits (384 workgroups): stays on the split-K skinny path
    assert not bool(ops.lib.lrp_linear_stream_dgrad_ok(8, 128256, 4096, 128256, 4096))


def test_gemm_big_m_row_chunks(ops):
    """activations beyond 2^30 elements (32-bit buffer offsets of the ping-pong kernel): the library issues the launch in row chunks --
    same kernel, same layout, no W^T fallback (B >= 19 prompts at S = 2048 on the gate/up dgrad operand).  NN form on a
    [K = 28672 + pad] operand with 37632 rows (two chunks: 37120 + 512): equal to the two halves computed separately, spot-checked vs fp64"""
    g = torch.Generator(device="cuda").manual_seed(3)
    M, K, N = 37632, 28672, 1024
    a_full = torch.randn(M, K + 64, generator=g, device="cuda", dtype=torch.float32).bfloat16()
    a = a_full[:, :K]
    assert M * a.stride(0) >= 2 ** 30
    w = (torch.randn(K, N, generator=g, device="cuda") * K ** -0.5).bfloat16()
    assert ops.gemm_nn_ok(a, w)
    out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
    ops.gemm_nn_2d(a, w, out)
    assert not torch.isnan(out).any()
    lo = torch.empty(18816, N, dtype=torch.bfloat16, device="cuda")
    hi = torch.empty(M - 18816, N, dtype=torch.bfloat16, device="cuda")
    ops.gemm_nn_2d(a[:18816], w, lo)
    ops.gemm_nn_2d(a[18816:], w, hi)
    assert torch.equal(out[:18816], lo) and torch.equal(out[18816:], hi)
    rows = torch.tensor([0, 1, 37119, 37120, 37121, M - 1], device="cuda")
    assert nmax(out[rows], f64(a[rows]) @ f64(w)) < TOL[torch.bfloat16]
    # NT form through lrp_gemm_nt
    wt = (torch.randn(512, K, generator=g, device="cuda") * K ** -0.5).bfloat16()
    o2 = torch.full((M, 512), float("nan"), dtype=torch.bfloat16, device="cuda")
    ops.gemm_nt_2d(a, wt, o2)
    assert not torch.isnan(o2).any() and nmax(o2[rows], f64(a[rows]) @ f64(wt).T) < TOL[torch.bfloat16]
    del a_full, a, out, o2
    torch.cuda.empty_cache()


def test_gemm_tail_split(ops):
    """GEMMs whose last round of 256 x 256 tiles is at most half full (Gemma-3-4B: 8192 x 2560 = 320 tiles = 1.25 rounds; SigLIP: 1088 tiles =
    4.25 rounds) are issued as whole rounds on the plain kernel + the remaining columns K-split over all CUs (ops.tail_split_cols).  Both operand forms through the dispatchers, with and without bias, into
    padded outputs: equal to fp64 within bf16 rounding, and the main part bit-equal to the undivided launch (same kernel, same tiles)"""
    g = torch.Generator().manual_seed(17)
    assert ops.tail_split_cols(8192, 2560, 2048) == 2048 and ops.tail_split_cols(8192, 2560, 20480) == 2048
    assert ops.tail_split_cols(8192, 4096, 4096) is None and ops.tail_split_cols(4096, 2560, 4096) is None
    assert ops.tail_split_cols(8192, 2560, 1024) is None          # K loop too short for 4 splits of >= 8 tiles
    assert ops.tail_split_cols(2048, 8192 + 2048, 4096) == 8192    # 8 x 40 tiles
    assert ops.tail_split_cols(16384, 4352, 2560) == 4096 and ops.tail_split_cols(16384, 4352, 1152) is None     # 4.25 rounds; K too short
    for (M, N, K) in [(8192, 2560, 2048), (8000, 2560 - 8, 4096), (2048, 10240, 2560), (16384, 4352, 2560)]:
        a = torch.randn(M, K, generator=g).bfloat16().cuda()
        w = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16().cuda()
        bias = torch.randn(N, generator=g).bfloat16().cuda()
        rows = torch.tensor([0, 1, 255, 256, M // 2 + 3, M - 1], device="cuda")
        for b_ in (None, bias):
            buf = torch.full((M, N + 64), float("nan"), dtype=torch.bfloat16, device="cuda")
            out = ops.linear_fwd(a, w, b_, out=buf[:, :N])
            assert out.data_ptr() == buf.data_ptr() and not torch.isnan(out).any() and torch.isnan(buf[:, N:]).all()
            ref = f64(a[rows]) @ f64(w).T + (0 if b_ is None else f64(b_))
            assert nmax(out[rows], ref) < TOL[torch.bfloat16]
            ops.TAIL_SPLIT = False
            try:
                whole = ops.linear_fwd(a, w, b_)
            finally:
                ops.TAIL_SPLIT = True
            main = ops.tail_split_cols(M, N, K)
            if not ops.splitk_ok(M, N, K):
                assert torch.equal(out[:, :main], whole[:, :main])
            assert nmax(out, whole.double()) < 2 * TOL[torch.bfloat16]
        # NN form: c = s W with W [Kc = contraction, N = output columns]
        s_ = torch.randn(M, K, generator=g).bfloat16().cuda()
        wn = (torch.randn(K, N, generator=g) * K ** -0.5).bfloat16().cuda()
        if N % 8 == 0:
            c = ops.linear_dgrad(s_, wn)
            assert nmax(c[rows], f64(s_[rows]) @ f64(wn)) < TOL[torch.bfloat16]
            ops.TAIL_SPLIT = False
            try:
                whole = ops.linear_dgrad(s_, wn)
            finally:
                ops.TAIL_SPLIT = True
            assert nmax(c, whole.double()) < 2 * TOL[torch.bfloat16]
        del a, w, s_, wn
    torch.cuda.empty_cache()


def test_gemm_skinny_single_split_and_big_n(ops):
    """tile counts that fill the chip alone (one split: the plain kernel writes the output, no slabs) and the many-tile split case"""
    g = torch.Generator().manual_seed(5)
    for (M, N, K) in [(4, 70000, 256), (8, 16384, 512), (160, 4096, 4096), (2, 4096, 2560), (5, 2048, 10240 + 64), (2048, 4096, 4096), (1000, 2048, 6144)]:
        a = torch.randn(M, K, generator=g).bfloat16().cuda()
        b = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16().cuda()
        out = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
        ops.gemm_skinny(a, b, out)
        assert not torch.isnan(out).any() and nmax(out, f64(a) @ f64(b).T) < 2e-5
    # more than one row of tiles (M = 2048: the one-prompt-per-step GEMMs against 4096-row weights), both operand forms, through the dispatchers
    a = torch.randn(2048, 4096, generator=g).bfloat16().cuda()
    w = (torch.randn(4096, 4096, generator=g) * 4096 ** -0.5).bfloat16().cuda()
    assert ops.splitk_ok(2048, 4096, 4096) and not ops.splitk_ok(8192, 4096, 4096) and not ops.splitk_ok(2048, 6144, 4096)
    assert nmax(ops.linear_fwd(a, w), f64(a) @ f64(w).T) < TOL[torch.bfloat16]
    assert nmax(ops.linear_dgrad(a, w), f64(a) @ f64(w)) < TOL[torch.bfloat16]
    # 320 tiles (1.25 rounds of the chip) with a long K loop: two K splits
    a = torch.randn(8192, 8192, generator=g).bfloat16().cuda()
    w = (torch.randn(2560, 8192, generator=g) * 8192 ** -0.5).bfloat16().cuda()
    assert ops.splitk_ok(8192, 2560, 8192) and nmax(ops.linear_fwd(a, w)[:512], f64(a[:512]) @ f64(w).T) < TOL[torch.bfloat16]


def _act64(x, act):
    if act == "silu":
        return x * torch.sigmoid(x)
    return torch.nn.functional.gelu(x, approximate="tanh")


@pytest.mark.parametrize("M,I,K,act", [(300, 512, 128, "silu"), (40, 64, 256, "silu"), (520, 1024, 256, "gelu_tanh")])
def test_gemm_gated_pair(ops, M, I, K, act):
    """ops.gemm_gated_fwd / _bwd: the gate/up Linear + the element-wise gated-MLP rule kernels (lrp_gated_act_fwd_il / _bwd_il) on the STORED gate/up output (small M; ref
    lxt/efficient/patches.py:145-157, lxt/explicit/models/llama.py:84-86,273-281), interleaved layout (ops.interleave_gate_up), vs fp64."""
    g_ = torch.Generator().manual_seed(M + I)
    bf = torch.bfloat16
    x = torch.randn(M, K, generator=g_).to(bf).cuda()
    wg, wu = ((torch.randn(I, K, generator=g_) * K ** -0.5).to(bf).cuda() for _ in range(2))
    Wgu = ops.interleave_gate_up(wg, wu)
    v = Wgu.view(I // 32, 2, 32, K)
    assert torch.equal(v[:, 0].reshape(I, K), wg) and torch.equal(v[:, 1].reshape(I, K), wu)
    gu, m = torch.full((M, 2 * I), float("nan"), dtype=bf, device="cuda"), torch.full((M, I), float("nan"), dtype=bf, device="cuda")
    ops.gemm_gated_fwd(x, Wgu, gu, m, act)
    gv = gu.view(M, I // 32, 2, 32)
    g, u = gv[:, :, 0].reshape(M, I), gv[:, :, 1].reshape(M, I)
    g64, u64 = f64(x) @ f64(wg).T, f64(x) @ f64(wu).T
    assert nmax(g, g64) < 2e-2 and nmax(u, u64) < 2e-2
    m_ref = _act64(f64(g), act).to(bf).double() * f64(u)          # from the STORED g, u: the rule itself is exact up to one rounding
    assert not torch.isnan(m).any() and nmax(m, m_ref) < 1e-2
    Adn = torch.randn(M, K, generator=g_).to(bf).cuda()
    Wd = (torch.randn(K, I, generator=g_) * K ** -0.5).to(bf).cuda()
    for eps_g, eps_lin in ((1e-10, 0.0), (1e-8, 1e-8)):
        Agu = torch.full((M, 2 * I), float("nan"), dtype=bf, device="cuda")
        ops.gemm_gated_bwd(Adn, Wd, gu, Agu, eps_g, eps_lin, act)
        Gm = ops.linear_dgrad(Adn, Wd)
        av = Agu.view(M, I // 32, 2, 32)
        Ag, Au = av[:, :, 0].reshape(M, I), av[:, :, 1].reshape(M, I)
        y = _act64(f64(g), act).to(bf).double()
        half = 0.5 * f64(Gm)
        Ag_ref = half * f64(u) * (y / (f64(g) + eps_g))
        Au_ref = half * y * (f64(u) / (f64(u) + eps_lin) if eps_lin else 1.0)
        assert not torch.isnan(Agu).any() and nmax(Ag, Ag_ref) < 2e-2 and nmax(Au, Au_ref) < 2e-2
        assert nmax(Gm, f64(Adn) @ f64(Wd)) < 2e-2


def _coef_split(coef, I):
    """the stash of lrp_gemm_gated_fwd_coef (include/lrp_hip.h): columns 8 t .. 8 t + 7 of a row = { cg x 4 | cu x 4 } of intermediate indices 4 t .. 4 t + 3"""
    v = coef.view(coef.shape[0], I // 4, 2, 4)
    return v[:, :, 0].reshape(-1, I), v[:, :, 1].reshape(-1, I)


@pytest.mark.parametrize("M,I,K,act,scaled", [(4096, 4096, 1024, "silu", False), (3000, 7168, 256, "gelu_tanh", False), (3000, 7168, 128, "silu", True),
                                              (8192, 2048, 512, "silu", True)])
def test_gemm_gated_coef_epilogues(ops, M, I, K, act, scaled):
    """lrp_gemm_gated_fwd_coef / _bwd_coef (round 6): the gated-MLP rules (ref lxt/efficient/patches.py:145-157, lxt/efficient/rules.py:88-100,
    lxt/explicit/models/llama.py:84-86,273-281) inside the epilogues of the gate/up forward GEMM and of the down-projection dgrad (NN form).  The
    forward leaves m = act(g) u and the COEFFICIENTS cg = 1/2 u act(g) / (g + eps_g), cu = 1/2 act(g) u / (u + eps_lin); the backward is
    Agu = Gm (*) (cg | cu).  Full and ragged tiles, both rule placements, both activations, with and without the K1n row scale -- against fp64 on
    the same bf16 operands, and against the GEMM + element-wise pair to bf16 rounding."""
    g_ = torch.Generator().manual_seed(M + I)
    bf = torch.bfloat16
    x = torch.randn(M, K, generator=g_).to(bf).cuda()
    wg, wu = ((torch.randn(I, K, generator=g_) * K ** -0.5).to(bf).cuda() for _ in range(2))
    Wgu = ops.interleave_gate_up(wg, wu)
    rs = (torch.rand(M, generator=g_) + 0.5).cuda() if scaled else None
    Adn = torch.randn(M, K, generator=g_).to(bf).cuda()
    Wd = (torch.randn(K, I, generator=g_) * K ** -0.5).to(bf).cuda()
    assert ops.gated_coef_ok(M, I, K, K, K, K, I, act, bf)
    assert not ops.gated_coef_ok(300, I, K, K, K, K, I, act, bf) and not ops.gated_coef_ok(M, I, K, K, K, K, I, act, torch.float32)
    g64, u64 = f64(x) @ f64(wg).T, f64(x) @ f64(wu).T
    if scaled:
        g64, u64 = g64 * rs.double()[:, None], u64 * rs.double()[:, None]
    y64 = _act64(g64, act)
    Gm64 = f64(Adn) @ f64(Wd)
    live = g64.abs() > 1e-3                      # (the stabilised ratios are compared away from their poles at g, u = -eps)
    live_u = u64.abs() > 1e-3
    # unfused pair for the cross-check
    gu2, m2 = torch.empty(M, 2 * I, dtype=bf, device="cuda"), torch.empty(M, I, dtype=bf, device="cuda")
    if not scaled:
        ops.gemm_gated_fwd(x, Wgu, gu2, m2, act)
    for eps_g, eps_lin in ((1e-10, 0.0), (1e-8, 1e-8)):
        coef, m = torch.full((M, 2 * I), float("nan"), dtype=bf, device="cuda"), torch.full((M, I), float("nan"), dtype=bf, device="cuda")
        ops.gemm_gated_fwd_coef(x, Wgu, coef, m, eps_g, eps_lin, act, rs=rs)
        assert not torch.isnan(m).any() and not torch.isnan(coef).any()
        assert nmax(m, y64 * u64) < 1e-2
        cg, cu = _coef_split(coef, I)
        cg_ref = torch.where(live, 0.5 * u64 * y64 / (g64 + eps_g), torch.zeros_like(g64))
        cu_ref = 0.5 * y64 * (torch.where(live_u, u64 / (u64 + eps_lin), torch.ones_like(u64)) if eps_lin else 1.0)
        assert nmax(torch.where(live, cg.double(), torch.zeros_like(g64)), cg_ref) < 1e-2
        assert nmax(torch.where(live_u, cu.double(), cu_ref), cu_ref) < 1e-2
        Agu = torch.full((M, 2 * I), float("nan"), dtype=bf, device="cuda")
        ops.gemm_gated_bwd_coef(Adn, Wd, coef, Agu)
        assert not torch.isnan(Agu).any()
        av = Agu.view(M, I // 32, 2, 32)
        Ag, Au = av[:, :, 0].reshape(M, I), av[:, :, 1].reshape(M, I)
        # exactly the rule on the STORED coefficients (one product, one rounding) ...
        assert nmax(Ag, Gm64 * cg.double()) < 1e-2 and nmax(Au, Gm64 * cu.double()) < 1e-2
        # ... and the reference's rule in fp64
        assert nmax(torch.where(live, Ag.double(), torch.zeros_like(g64)), Gm64 * cg_ref) < 2e-2
        assert nmax(torch.where(live_u, Au.double(), Gm64 * cu_ref), Gm64 * cu_ref) < 2e-2
        if not scaled:
            assert nmax(m, m2.double()) < 1e-2
            Agu2 = torch.empty_like(Agu)
            ops.gemm_gated_bwd(Adn, Wd, gu2, Agu2, eps_g, eps_lin, act)
            a2 = Agu2.view(M, I // 32, 2, 32)
            Ag2, Au2 = a2[:, :, 0].reshape(M, I).double(), a2[:, :, 1].reshape(M, I).double()
            z = torch.zeros_like(g64)
            assert nmax(torch.where(live, Ag.double(), z), torch.where(live, Ag2, z)) < 2e-2
            assert nmax(torch.where(live_u, Au.double(), z), torch.where(live_u, Au2, z)) < 2e-2


@pytest.mark.parametrize("M,N,K,seq,rope_cols", [(2048, 6144, 256, 512, 5120), (2304, 5632, 192, 192, 4608), (8192, 6144, 128, 2048, 5120)])
def test_gemm_nt_rs_rope(ops, M, N, K, seq, rope_cols):
    """lrp_gemm_nt_rs_rope (round 6): the fused QKV forward with RoPE in the epilogue -- HF's apply_rotary_pos_emb (q cos + rotate_half(q) sin) on
    the head columns [0, rope_cols), the K1n row scale everywhere, v columns untouched; positions = row % seq (prompts that end inside a 256-row
    tile included).  Against fp64 on the same bf16 operands, and against the two-launch form lrp_gemm_nt_rs + lrp_rope_fwd to bf16 rounding (the
    fused form rounds once)."""
    g_ = torch.Generator().manual_seed(M + N + seq)
    bf, d = torch.bfloat16, 128
    x = torch.randn(M, K, generator=g_).to(bf).cuda()
    W = (torch.randn(N, K, generator=g_) * K ** -0.5).to(bf).cuda()
    rs = (torch.rand(M, generator=g_) + 0.5).cuda()
    inv = 1.0 / (10000.0 ** (torch.arange(0, d, 2, dtype=torch.float32) / d))
    fr = torch.arange(seq + 7, dtype=torch.float32)[:, None] * inv[None]
    emb = torch.cat((fr, fr), -1)
    cos, sin = emb.cos().to(bf).float().cuda().contiguous(), emb.sin().to(bf).float().cuda().contiguous()
    out = torch.full((M, N), float("nan"), dtype=bf, device="cuda")
    assert ops.gemm_nt_rs_rope_ok(x, W, out, seq, rope_cols, d)
    assert not ops.gemm_nt_rs_rope_ok(x, W, out, seq, rope_cols, 64) and not ops.gemm_nt_rs_rope_ok(x[:300], W, out[:300], seq, rope_cols, d)
    assert not ops.gemm_nt_rs_rope_ok(x, W, out, seq + 8, rope_cols, d)                      # seq must be a multiple of 16
    ops.gemm_nt_rs_rope(x, W, rs, cos, sin, out, seq, rope_cols, d)
    z = rs.double()[:, None] * (f64(x) @ f64(W).T)
    zr = z[:, :rope_cols].view(M, rope_cols // d, d)
    pos = torch.arange(M, device="cuda") % seq
    c, s_ = cos.double()[pos][:, None, :], sin.double()[pos][:, None, :]
    rot = torch.cat((-zr[..., d // 2:], zr[..., : d // 2]), -1)
    ref = torch.cat(((zr * c + rot * s_).view(M, rope_cols), z[:, rope_cols:]), 1)
    assert not torch.isnan(out).any() and nmax(out, ref) < 1e-2
    assert (out.double() - ref).abs().max() <= ref.abs().max() * 2.0 ** -8                     # ONE bf16 rounding of the fp32 result
    two = torch.empty(M, N, dtype=bf, device="cuda")
    ops.gemm_nt_rs(x, W, rs, two)
    two_r = ops.rope_fwd(two, torch.empty(M, rope_cols, dtype=bf, device="cuda"), cos, sin, seq, rope_cols // d, d)
    assert nmax(out[:, :rope_cols], two_r.double()) < 1.5e-2 and torch.equal(out[:, rope_cols:], two[:, rope_cols:])


@pytest.mark.parametrize("M,H,K,I", [(2304, 5632, 512, 2816), (2100, 5632, 256, 3072)])
def test_gemm_norm_fused_epilogues(ops, M, H, K, I):
    """K1n (include/lrp_hip.h): Llama-type RMSNorm (ref lxt/efficient/patches.py:111-123, variance detached = identity rule) and the residual
    sums around it folded into the epilogues of the ping-pong GEMM -- full and ragged row tiles.  Each entry point against fp64 on the same
    bf16 operands; where a stand-alone kernel pair computes the same thing, against that pair to one bf16 rounding."""
    g_ = torch.Generator().manual_seed(M + H)
    bf = torch.bfloat16
    mk = lambda *sh, sc=1.0: (torch.randn(*sh, generator=g_) * sc).to(bf).cuda()      # noqa: E731
    x, res = mk(M, K), mk(M, H)
    W = mk(H, K, sc=K ** -0.5)
    assert ops.norm_fused_ok(M, H, K, K, K, False, bf) and not ops.norm_fused_ok(300, H, K, K, K, False, bf)
    assert not ops.norm_fused_ok(M, H + 64, K, K, K, False, bf) and not ops.norm_fused_ok(M, H, K, K, K, False, torch.float32)
    # ---- out = res + x W^T, partial sums of squares of the rounded rows, rstd from the partials
    out = torch.full((M, H), float("nan"), dtype=bf, device="cuda")
    ssq = torch.full((H // 64, M), float("nan"), dtype=torch.float32, device="cuda")
    ops.gemm_res_ssq(x, W, res, out, ssq)
    ref = f64(res) + f64(x) @ f64(W).T
    assert not torch.isnan(out).any() and not torch.isnan(ssq).any()
    assert nmax(out, ref) < 1e-2
    assert (out.double() - ref).abs().max() <= (ref.abs().max() * 2.0 ** -8)          # one bf16 rounding of the fp32 sum
    ssq_ref = (out.double() ** 2).view(M, H // 64, 64).sum(-1).T                       # from the STORED rows: exact up to fp32 summation
    assert torch.allclose(ssq.double(), ssq_ref, rtol=1e-5, atol=1e-6)
    # the explicit placement's form: the Linear's own output is kept beside the sum (its stabiliser divides by it); everything else unchanged
    out_x, ssq_x, raw = torch.empty_like(out), torch.empty_like(ssq), torch.full((M, H), float("nan"), dtype=bf, device="cuda")
    ops.gemm_res_ssq(x, W, res, out_x, ssq_x, raw=raw)
    assert torch.equal(out_x, out) and torch.equal(ssq_x, ssq) and not torch.isnan(raw).any()
    assert nmax(raw, f64(x) @ f64(W).T) < 1e-2 and torch.equal(raw, ops.linear_fwd(x, W))
    eps = 1e-5
    rstd = ops.rms_rstd(ssq, M, H, eps, torch.empty(M, dtype=torch.float32, device="cuda"))
    rstd_ref = torch.rsqrt((out.double() ** 2).mean(-1) + eps)
    assert torch.allclose(rstd.double(), rstd_ref, rtol=1e-5)
    _, rstd_k = ops.add_rmsnorm_fwd(out, None, torch.ones(H, dtype=bf, device="cuda"), eps)      # the stand-alone kernel's statistic
    assert torch.allclose(rstd, rstd_k, rtol=1e-5)
    # ---- consumer: rows scaled after the product, (rstd x) W^T = rstd (x W^T)
    N2 = 2 * I
    W2 = mk(N2, H, sc=H ** -0.5)
    y = torch.full((M, N2), float("nan"), dtype=bf, device="cuda")
    ops.gemm_nt_rs(out, W2, rstd, y)
    y_ref = rstd.double()[:, None] * (f64(out) @ f64(W2).T)
    assert not torch.isnan(y).any() and nmax(y, y_ref) < 1e-2
    # (the gate/up consumer with the gated rule behind the row scale: test_gemm_gated_coef_epilogues, scaled cases)
    # ---- backward: G_h = rstd (.) (A W) + G_res from the stored weight (NN), also in place on the residual gradient
    A = mk(M, N2)
    Gres = mk(M, H)
    Gh = torch.full((M, H), float("nan"), dtype=bf, device="cuda")
    ops.gemm_nn_rs_res(A, W2, rstd, Gres, Gh)
    Gh_ref = rstd.double()[:, None] * (f64(A) @ f64(W2)) + f64(Gres)
    assert not torch.isnan(Gh).any() and nmax(Gh, Gh_ref) < 1e-2
    half = torch.full((M,), 0.5, device="cuda")
    Gq = torch.full((M, H), float("nan"), dtype=bf, device="cuda")
    ops.gemm_nn_rs(A, W2, half, Gq)                                                   # plain row scale: 1/2 x the dgrad, exact
    assert torch.equal(Gq.float(), 0.5 * ops.linear_dgrad(A, W2).float())
    Gx = ops.linear_dgrad(A, W2)                                                      # stand-alone pair: dgrad, then the norm's backward kernel
    Gh2 = torch.empty_like(Gh)
    ops.rmsnorm_bwd_add2(Gres, Gx, torch.ones(H, dtype=bf, device="cuda"), rstd, None, None, Gh2, None, None, 0.0, 0.0, 0.0)
    assert nmax(Gh, Gh2) < 1e-2                                                        # (the pair rounds Gx to bf16 first: one rounding apart)
    inpl = Gres.clone()
    ops.gemm_nn_rs_res(A, W2, rstd, inpl, inpl)
    assert torch.equal(inpl, Gh)


def test_gemm_batched_and_f32_out(ops):
    a, b = rnd(3, 70, 96, seed=4), rnd(3, 50, 96, seed=5)
    assert nmax(ops.gemm_nt(a, b), f64(a) @ f64(b).transpose(1, 2)) < 2e-5
    a16, b16 = a.bfloat16(), b[0].bfloat16()
    out = ops.gemm_nt(a16, b16, out_dtype=torch.float32)
    assert out.dtype == torch.float32 and nmax(out, f64(a16) @ f64(b16).T) < 1e-5
    # ragged K (padding path of the wrapper)
    a, b = rnd(16, 10, seed=6), rnd(5, 10, seed=7)
    assert nmax(ops.gemm_nt(a, b), f64(a) @ f64(b).T) < 2e-5


@pytest.mark.parametrize("dtype", DTYPES)
def test_transpose_cast(ops, dtype):
    x = rnd(3, 70, 130, dtype=dtype)
    assert torch.equal(ops.transpose(x), x.transpose(1, 2).contiguous())
    y = rnd(1000, 33)
    assert torch.equal(ops.cast(y, torch.bfloat16), y.bfloat16())
    assert torch.equal(ops.cast(y.bfloat16(), torch.float32), y.bfloat16().float())


# ------------------------------------------------------------------------------- Linear eps rule (K1)
def test_linear_eps_c1_golden(ops):
    """BASELINE config 1 on the device: Linear(768->768), M=1, against the reference's own output."""
    fx = load("rules.npz")
    for tag in ("c1", "toy", "mid"):
        x, W, b, g = (t(fx[f"lin_{tag}_{k}"]).cuda() for k in "xWbg")
        for eps_tag, eps in (("f", 1e-6), ("r", 1e-8)):
            z = ops.gemm_nt(x, W, b)
            # relevance form with the forward's own z (what lf.linear_epsilon saves): s = R/(z+eps)
            R_out = ops.mul(z, g)
            s = ops.eps_scale(R_out, z, 1.0, eps, relevance=True)
            R_in = ops.mul(ops.gemm_nt(s, ops.transpose(W)), x)
            assert nmax(R_in, fx[f"lin_{tag}_{eps_tag}_Rin"]) < 2e-5, (tag, eps_tag)
            # gradient form: G = g  ->  R_in = x * ((g * z/(z+eps)) W)
            A = ops.eps_scale(g, z, 1.0, eps)
            assert nmax(ops.mul(ops.gemm_nt(A, ops.transpose(W)), x), fx[f"lin_{tag}_{eps_tag}_Rin"]) < 2e-5
            if ops.smallm_ok(x.shape[0], W, x):  # W-streaming forward + dgrad pair (what linear_epsilon / the engine's top rows run)
                z2 = ops.linear_smallm_fwd(x, W, b)
                assert nmax(z2, z) < 1e-5
                # (the relevance R_out = z (*) g was formed from the GEMM's z: divide by THAT z -- z/(z2+eps) is O(1)-sensitive
                # to the last-bit differences between two fp32 evaluations wherever |z| ~ eps)
                R2 = ops.linear_smallm_dgrad(R_out, W, z=z, x=x, eps=eps, relevance_in=True, relevance_out=True, out_dtype=torch.float32)
                R3 = ops.linear_smallm_dgrad(g, W, z=z2, x=x, eps=eps, relevance_out=True, out_dtype=torch.float32)
                assert nmax(R2, fx[f"lin_{tag}_{eps_tag}_Rin"]) < 2e-5 and nmax(R3, fx[f"lin_{tag}_{eps_tag}_Rin"]) < 2e-5, (tag, eps_tag)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(1, 768, 768), (3, 1000, 520), (4, 14336, 4096), (16, 4096, 14336), (8, 300, 4096), (5, 33, 72),
                                   (2, 128256, 1024)])
def test_linear_smallm_fwd_dgrad(ops, dtype, M, N, K):
    """the W-streaming small-M kernels (M <= 16, any N, K a multiple of 16 bytes) against fp64: forward with bias, dgrad in its
    three input forms (plain gradient, gradient x z/(z+eps), relevance / (z+eps)), both output forms, strided g / z rows"""
    e = 16 // torch.empty(0, dtype=dtype).element_size()
    K = -(-K // e) * e
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    x, W = rnd(M, K, dtype=dtype, seed=1), rnd(N, K, dtype=dtype, seed=2, scale=K ** -0.5)
    b = rnd(N, dtype=dtype, seed=4)
    z = ops.linear_smallm_fwd(x, W, b)
    zf = ops.linear_smallm_fwd(x, W, None, out_dtype=torch.float32)
    zref = f64(x) @ f64(W).T
    assert nmax(z, zref + f64(b)) < tol and nmax(zf, zref) < (2e-5 if dtype == torch.float32 else 2e-3)
    assert z.dtype == dtype and zf.dtype == torch.float32
    g2 = rnd(M, 2 * N, dtype=dtype, seed=3)
    g = g2[:, N:] if N % e == 0 else g2[:, N:].contiguous()        # a strided view where the alignment allows it
    zs = z.double()
    for eps in (0.0, 1e-6):
        ref_plain = f64(g) @ f64(W)
        assert nmax(ops.linear_smallm_dgrad(g, W), ref_plain) < tol
        ratio = zs / (zs + eps) if eps else 1.0
        out = ops.linear_smallm_dgrad(g, W, z=z, eps=eps, out_dtype=torch.float32)
        assert out.dtype == torch.float32 and nmax(out, (f64(g) * ratio) @ f64(W)) < tol
        if eps:
            outR = ops.linear_smallm_dgrad(g, W, z=z, x=x, eps=eps, relevance_in=True, relevance_out=True, out_dtype=torch.float32)
            assert nmax(outR, ((f64(g) / (zs + eps)) @ f64(W)) * f64(x)) < tol


# ----------------------------------------------------------------------------------- element-wise
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n", [1, 37, 4096, 100003])
def test_eps_scale_mul_add2(ops, dtype, n):
    g, z = rnd(n, dtype=dtype, seed=1), rnd(n, dtype=dtype, seed=2)
    for c, eps in ((1.0, 1e-6), (2.0, 1e-8), (1.0, 0.0)):
        ref = f64(g) * (f64(z) / (c * f64(z) + eps) if eps else 1.0 / c)
        assert nmax(ops.eps_scale(g, z, c, eps), ref) < TOL[dtype]
    assert nmax(ops.eps_scale(g, z, 2.0, 1e-3, relevance=True), f64(g) / (2 * f64(z) + 1e-3)) < TOL[dtype]
    assert nmax(ops.mul(g, z), f64(g) * f64(z)) < TOL[dtype]
    a, b = rnd(n, dtype=dtype, seed=3), rnd(n, dtype=dtype, seed=4)
    Ra, Rb = ops.add2_rule_bwd(a, b, g, 1e-6)
    s = f64(g) / (f64(a) + f64(b) + 1e-6)
    assert nmax(Ra, s * f64(a)) < TOL[dtype] * 50 and nmax(Rb, s * f64(b)) < TOL[dtype] * 50


def test_rule_goldens_elementwise(ops):
    fx = load("rules.npz")
    a, b, g = (t(fx[k]).cuda() for k in ("add_a", "add_b", "add_g"))
    Ra, Rb = ops.add2_rule_bwd(a, b, (a + b) * g, 1e-8)
    assert nmax(Ra, fx["add_Ra"]) < 2e-5 and nmax(Rb, fx["add_Rb"]) < 2e-5
    x, G = t(fx["act_x"]).cuda(), t(fx["act_G"]).cuda()
    assert nmax(ops.act_fwd(x, "silu"), fx["act_silu_y"]) < 1e-6
    assert nmax(ops.act_bwd(G, x, "silu", 1e-10), fx["act_silu_Gin"]) < 1e-5
    assert nmax(ops.act_fwd(x, "gelu_tanh"), fx["act_gelut_y"]) < 1e-6
    assert nmax(ops.act_bwd(G, x, "gelu_tanh", 1e-10), fx["act_gelut_Gin"]) < 1e-5
    # softmax rule incl. -inf mask entries
    xs, gs = t(fx["sm_x"]).cuda(), t(fx["sm_g"]).cuda()
    p = ops.softmax_fwd(xs)
    assert nmax(p, fx["sm_p"]) < 1e-6
    Rx = ops.softmax_rule_bwd(xs, p, p * gs)
    assert torch.isfinite(Rx).all() and nmax(Rx, fx["sm_Rx"]) < 2e-5
    # layer norm
    x, w, bb, g = (t(fx[k]).cuda() for k in ("ln_x", "ln_w", "ln_b", "ln_g"))
    y, mean, rstd = ops.layernorm_fwd(x, w, bb, 1e-12)
    assert nmax(y, fx["ln_y"]) < 1e-5
    Gx = ops.layernorm_bwd(g, y, w, rstd, 1e-6)      # G_y = R_out / y = g
    assert nmax(ops.mul(Gx, x), fx["ln_Rin"]) < 5e-5
    # rms norm forward
    x, w = t(fx["rms_x"]).cuda().reshape(-1, 64), t(fx["rms_w"]).cuda()
    y, _ = ops.add_rmsnorm_fwd(x, None, w, 1e-5)
    assert nmax(y, t(fx["rms_y"]).reshape(-1, 64)) < 1e-6


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,I", [(5, 24), (130, 1024)])
def test_gated_act(ops, dtype, M, I):
    gu = rnd(M, 2 * I, dtype=dtype, seed=1)
    g, u = gu[:, :I], gu[:, I:]
    m = ops.gated_act_fwd(g, u)
    act = F.silu(f64(g)).to(dtype).double()
    assert nmax(m, act * f64(u)) < TOL[dtype]
    Gm = rnd(M, I, dtype=dtype, seed=2)
    Agu = torch.empty(M, 2 * I, dtype=dtype, device="cuda")
    for eps_g, eps_lin in ((1e-10, 0.0), (1e-8, 1e-8)):
        ops.gated_act_bwd(Gm, g, u, Agu[:, :I], Agu[:, I:], eps_g, eps_lin)
        Ag = 0.5 * f64(Gm) * f64(u) * act / (f64(g) + eps_g)
        Au = 0.5 * f64(Gm) * act * (f64(u) / (f64(u) + eps_lin) if eps_lin else 1.0)
        assert nmax(Agu[:, :I], Ag) < TOL[dtype] * 20 and nmax(Agu[:, I:], Au) < TOL[dtype] * 20


def _rope_ref(x, cos, sin):
    d = x.shape[-1]
    rot = torch.cat((-x[..., d // 2:], x[..., : d // 2]), -1)
    return x * cos + rot * sin


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("d", [16, 128])
def test_rope(ops, dtype, d):
    B, S, nh = 2, 37, 3
    x = rnd(B * S, nh * d + 8, dtype=dtype, seed=1)[:, : nh * d]      # strided rows
    pos = torch.arange(S, dtype=torch.float64)
    inv = 1.0 / (10000.0 ** (torch.arange(0, d, 2, dtype=torch.float64) / d))
    emb = torch.cat((pos[:, None] * inv, pos[:, None] * inv), -1)
    cos, sin = emb.cos().float().cuda(), emb.sin().float().cuda()
    xr = ops.rope_fwd(x, torch.empty(B * S, nh * d, dtype=dtype, device="cuda"), cos, sin, S, nh, d)
    x4 = f64(x).reshape(B, S, nh, d)
    ref = _rope_ref(x4, f64(cos)[None, :, None], f64(sin)[None, :, None])
    assert nmax(xr, ref.reshape(B * S, -1)) < TOL[dtype]
    G = rnd(B * S, nh * d, dtype=dtype, seed=2)
    for er, el in ((0.0, 0.0), (1e-8, 1e-8)):
        A = ops.rope_bwd(G, xr if er else None, x if el else None, torch.empty_like(G), cos, sin, S, nh, d, er, el)
        xg = x4.clone().requires_grad_()
        yr = _rope_ref(xg, f64(cos)[None, :, None], f64(sin)[None, :, None])
        Gp = f64(G).reshape(B, S, nh, d)
        if er:
            xrd = f64(xr).reshape(B, S, nh, d)
            Gp = Gp * xrd / (xrd + er)
        gx, = torch.autograd.grad(yr, xg, Gp)
        if el:
            gx = gx * x4 / (x4 + el)
        assert nmax(A, gx.reshape(B * S, -1)) < TOL[dtype] * 5


# ---------------------------------------------------------------------------------------- row ops
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,H", [(3, 64), (129, 4096), (7, 100)])
def test_rmsnorm_fwd_bwd(ops, dtype, M, H):
    h, br, w = rnd(M, H, dtype=dtype, seed=1), rnd(M, H, dtype=dtype, seed=2), rnd(H, dtype=dtype, seed=3)
    hs = torch.empty_like(h)
    y, rstd = ops.add_rmsnorm_fwd(h, br, w, 1e-5, hsum_out=hs)
    hsum = (f64(h) + f64(br)).to(dtype).double()
    rs = torch.rsqrt(hsum.pow(2).mean(-1, keepdim=True) + 1e-5)
    assert nmax(hs, hsum) < 1e-6 and nmax(rstd, rs[:, 0]) < 1e-5
    assert nmax(y, f64(w) * (hsum * rs).to(dtype).double()) < TOL[dtype]
    y1, _ = ops.add_rmsnorm_fwd(h, None, w, 1e-5, w_offset=1.0)
    rs1 = torch.rsqrt(f64(h).pow(2).mean(-1, keepdim=True) + 1e-5)
    assert nmax(y1, f64(h) * rs1 * (1 + f64(w))) < TOL[dtype]
    # backward
    Gres, Gx = rnd(M, H, dtype=dtype, seed=4), rnd(M, H, dtype=dtype, seed=5)
    Gs, A = torch.empty_like(h), torch.empty_like(h)
    rel = torch.empty(M, device="cuda")
    for ea, el in ((0.0, 0.0), (1e-8, 1e-8)):
        ops.rmsnorm_bwd_add2(Gres, Gx, w, rstd, hs, br, Gs, A, rel, 0.0, ea, el)
        Gh = f64(Gres) + f64(Gx) * f64(w) * rstd.double()[:, None]
        Gs_ref = Gh * (f64(hs) / (f64(hs) + ea) if ea else 1.0)
        A_ref = Gs_ref * (f64(br) / (f64(br) + el) if el else 1.0)
        assert nmax(Gs, Gs_ref) < TOL[dtype] * 5 and nmax(A, A_ref) < TOL[dtype] * 5
        assert nmax(rel, (f64(hs) * Gh).sum(-1)) < TOL[dtype] * 5
    ops.rmsnorm_bwd_add2(Gres, None, None, None, hs, br, Gs, A, None, 0.0, 1e-8, 1e-8)
    assert nmax(Gs, f64(Gres) * f64(hs) / (f64(hs) + 1e-8)) < TOL[dtype] * 5
    ops.rmsnorm_bwd_add2(None, Gx, w, rstd, None, None, Gs, None, None, 0.0, 0.0, 0.0)
    assert nmax(Gs, f64(Gx) * f64(w) * rstd.double()[:, None]) < TOL[dtype] * 5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("heads,d", [(4, 32), (8, 256), (3, 128), (2, 64)])
def test_head_rmsnorm_fwd_bwd(ops, dtype, heads, d):
    """per-head RMSNorm on strided slices of a fused projection output (Gemma-3 q_norm / k_norm; ref lxt/efficient/models/gemma3.py:11-12:
    rstd detached): forward y = (1 + w) x rstd and backward G (1 + w) rstd vs fp64, written into a strided slice as well"""
    rows, lead, tail = 77, 16, 24
    g = torch.Generator().manual_seed(heads * d)
    big = torch.randn(rows, lead + heads * d + tail, generator=g).to(dtype).cuda()
    x = big[:, lead: lead + heads * d]
    w = (torch.randn(d, generator=g) * 0.1).to(dtype).cuda()
    y = torch.empty(rows, heads * d, dtype=dtype, device="cuda")
    rstd = torch.empty(rows * heads, dtype=torch.float32, device="cuda")
    ops.head_rmsnorm_fwd(x, w, y, rstd, heads, d, 1e-6, 1.0)
    x64 = f64(x).view(rows, heads, d)
    r64 = torch.rsqrt(x64.pow(2).mean(-1, keepdim=True) + 1e-6)
    assert nmax(y.view(rows, heads, d), x64 * r64 * (1 + f64(w))) < TOL[dtype] and nmax(rstd.view(rows, heads, 1), r64) < 1e-5
    G = torch.randn(rows, heads * d, generator=g).to(dtype).cuda()
    outb = torch.zeros(rows, lead + heads * d + tail, dtype=dtype, device="cuda")
    ops.head_rmsnorm_bwd(G, w, rstd, outb[:, lead: lead + heads * d], heads, d, 1.0)
    assert nmax(outb[:, lead: lead + heads * d].reshape(rows, heads, d), f64(G).view(rows, heads, d) * (1 + f64(w)) * r64) < TOL[dtype]
    assert float(outb[:, :lead].abs().max()) == 0.0 and float(outb[:, lead + heads * d:].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,H,w_off", [(5, 256, 1.0), (131, 2560, 1.0), (67, 4096, 0.0), (9, 1152, 1.0)])
def test_sandwich_norm_site_kernels_bit_identical(ops, dtype, M, H, w_off):
    """Gemma-3's post-norm / residual add / next pre-norm as ONE pass (lrp_sandwich_norm_fwd / _bwd; ref HF Gemma3DecoderLayer, identity rules of
    lxt/efficient/models/gemma3.py:11-19): bit-identical to the two add_rmsnorm_fwd / two rmsnorm_bwd_add2 launches it replaces (which are
    tested against fp64 above), both weight conventions, rows of 1 / 2 / 4 chunks per thread"""
    x, res = rnd(M, H, dtype=dtype, seed=1), rnd(M, H, dtype=dtype, seed=2)
    w1, w2 = rnd(H, dtype=dtype, seed=3) * 0.1, rnd(H, dtype=dtype, seed=4) * 0.1
    if not ops.sandwich_norm_ok(x):
        pytest.skip("row too wide for the in-register kernel")
    e = lambda: torch.empty(M, H, dtype=dtype, device="cuda")      # noqa: E731
    r = lambda: torch.empty(M, dtype=torch.float32, device="cuda")      # noqa: E731
    pa, r1 = ops.add_rmsnorm_fwd(x, None, w1, 1e-6, w_off)
    h_ref, y_ref, r2 = e(), e(), r()
    ops.add_rmsnorm_fwd(res, pa, w2, 1e-6, w_off, hsum_out=h_ref, y=y_ref, rstd=r2)
    h1, y, q1, q2 = e(), e(), r(), r()
    ops.sandwich_norm_fwd(x, res, w1, w2, 1e-6, w_off, h1, y, q1, q2)
    assert torch.equal(h1, h_ref) and torch.equal(y, y_ref) and torch.equal(q1, r1) and torch.equal(q2, r2)
    h1b, q1b, q2b = e(), r(), r()
    ops.sandwich_norm_fwd(x, res, w1, None, 1e-6, w_off, h1b, None, q1b, q2b)                # without the pre-norm output
    assert torch.equal(h1b, h_ref) and torch.equal(q2b, r2)
    # backward: Gs = Gres + Gx w_pre' rstd_pre, Ga = Gs w_post' rstd_post
    Gres, Gx = rnd(M, H, dtype=dtype, seed=5), rnd(M, H, dtype=dtype, seed=6)
    for gr in (Gres, None):
        Gs_ref, Ga_ref, Gs, Ga = e(), e(), e(), e()
        ops.rmsnorm_bwd_add2(gr, Gx, w2, r2, None, None, Gs_ref, None, None, w_off, 0.0, 0.0)
        ops.rmsnorm_bwd_add2(None, Gs_ref, w1, r1, None, None, Ga_ref, None, None, w_off, 0.0, 0.0)
        ops.sandwich_norm_bwd(gr, Gx, w2, r2, w1, r1, Gs, Ga, w_off)
        assert torch.equal(Gs, Gs_ref) and torch.equal(Ga, Ga_ref)
        gs64 = (f64(gr) if gr is not None else 0.0) + f64(Gx) * (f64(w2) + w_off) * f64(r2)[:, None]
        assert nmax(Gs, gs64) < TOL[dtype] * 5 and nmax(Ga, gs64 * (f64(w1) + w_off) * f64(r1)[:, None]) < TOL[dtype] * 5


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("nq,nk,d", [(8, 4, 256), (4, 1, 128), (6, 2, 64)])
def test_qk_norm_rope_site_kernels_bit_identical(ops, dtype, nq, nk, d):
    """per-head q / k RMSNorm + RoPE on the fused projection output in one pass, and the qkv dgrad's operand (group sums of dK / dV, RoPE's
    backward, the norms' scale) in one pass: bit-identical to the 4- and 6-launch sequences of stand-alone kernels (each tested against fp64
    above); strided rows on both sides"""
    B, S = 2, 53
    rows = B * S
    g = torch.Generator().manual_seed(nq * d)
    qkv = torch.randn(rows, (nq + 2 * nk) * d + 16, generator=g).to(dtype).cuda()[:, : (nq + 2 * nk) * d]
    wq, wk = (torch.randn(d, generator=g) * 0.1).to(dtype).cuda(), (torch.randn(d, generator=g) * 0.1).to(dtype).cuda()
    pos = torch.arange(S, dtype=torch.float64)
    inv = 1.0 / (10000.0 ** (torch.arange(0, d, 2, dtype=torch.float64) / d))
    emb = torch.cat((pos[:, None] * inv, pos[:, None] * inv), -1)
    cos, sin = emb.cos().float().cuda(), emb.sin().float().cuda()
    e = lambda n: torch.empty(rows, n, dtype=dtype, device="cuda")      # noqa: E731
    rq, rk = torch.empty(rows * nq, device="cuda"), torch.empty(rows * nk, device="cuda")
    qn, kn = e(nq * d), e(nk * d)
    ops.head_rmsnorm_fwd(qkv[:, : nq * d], wq, qn, rq, nq, d, 1e-6, 1.0)
    ops.head_rmsnorm_fwd(qkv[:, nq * d: (nq + nk) * d], wk, kn, rk, nk, d, 1e-6, 1.0)
    qr_ref, kr_ref = ops.rope_fwd(qn, e(nq * d), cos, sin, S, nq, d), ops.rope_fwd(kn, e(nk * d), cos, sin, S, nk, d)
    rq2, rk2 = torch.empty_like(rq), torch.empty_like(rk)
    qr, kr = ops.qk_norm_rope_fwd(qkv, wq, wk, e(nq * d), e(nk * d), rq2, rk2, cos, sin, S, nq, nk, d, 1e-6, 1.0)
    assert torch.equal(qr, qr_ref) and torch.equal(kr, kr_ref) and torch.equal(rq, rq2) and torch.equal(rk, rk2)
    # backward
    rep = nq // nk
    dq, dk_h, dv_h = (torch.randn(rows, nq * d, generator=g).to(dtype).cuda() for _ in range(3))
    A_ref = torch.zeros(rows, (nq + 2 * nk) * d + 8, dtype=dtype, device="cuda")[:, : (nq + 2 * nk) * d]
    dk = ops.gqa_reduce(dk_h, e(nk * d), rows, nk, rep, d)
    ops.gqa_reduce(dv_h, A_ref[:, (nq + nk) * d:], rows, nk, rep, d)
    Gqn = ops.rope_bwd(dq, None, None, e(nq * d), cos, sin, S, nq, d, 0.0, 0.0)
    Gkn = ops.rope_bwd(dk, None, None, e(nk * d), cos, sin, S, nk, d, 0.0, 0.0)
    ops.head_rmsnorm_bwd(Gqn, wq, rq, A_ref[:, : nq * d], nq, d, 1.0)
    ops.head_rmsnorm_bwd(Gkn, wk, rk, A_ref[:, nq * d: (nq + nk) * d], nk, d, 1.0)
    A = torch.zeros(rows, (nq + 2 * nk) * d + 8, dtype=dtype, device="cuda")[:, : (nq + 2 * nk) * d]
    ops.qkv_bwd_pack(dq, dk_h, dv_h, wq, wk, rq, rk, cos, sin, A, S, nq, nk, d, 1.0)
    assert torch.equal(A, A_ref)


def test_readout_argmax_headseed(ops):
    e, G = rnd(50, 264, seed=1), rnd(50, 264, seed=2)
    assert nmax(ops.readout(e, G), (f64(e) * f64(G)).sum(-1)) < 1e-5
    logits = rnd(3, 1000, seed=3)
    logits[1, 77] = 50.0
    idx, val = ops.argmax_rows(logits)
    assert torch.equal(idx.long().cpu(), logits.argmax(-1).cpu()) and torch.equal(val, logits.max(-1).values)
    # vocabulary-sized rows (16-byte loads, unrolled body + remainders), an unaligned row pitch, ties -> lowest index, maximum in the tail
    for V, ld in ((128256, 128256), (262208, 262208), (50257, 50257 + 3), (4099, 4099)):
        big = torch.randn(4, ld, device="cuda")[:, :V]
        big[0, V - 1] = 60.0
        big[1, 5] = big[1, V // 2] = 55.0
        big[2, 4 * (V // 4) - 1] = 70.0
        i2, v2 = ops.argmax_rows(big)
        assert i2.tolist() == [V - 1, 5, 4 * (V // 4) - 1, int(big[3].argmax())] and torch.equal(v2, big.max(-1).values)
    W, wn, rstd = rnd(1000, 264, seed=4), rnd(264, seed=5), rnd(3, seed=6).abs()
    out = ops.head_seed(W, logits, idx, wn, rstd, torch.empty(3, 264, device="cuda"), 0.0, 1e-8)
    z = val.double()
    ref = (z / (z + 1e-8) * rstd.double())[:, None] * f64(W)[idx.long()] * f64(wn)
    assert nmax(out, ref) < 1e-5


# -------------------------------------------------------------------------------------- attention
def _attn_ref(q, k, v, scale, causal, window):
    """fp64 eager attention with the LRP gradient modifiers; q [B,Hq,S,d], k/v [B,Hkv,S,d]"""
    B, Hq, S, d = q.shape
    rep = Hq // k.shape[1]
    kx, vx = k.repeat_interleave(rep, 1), v.repeat_interleave(rep, 1)
    s = q @ kx.transpose(-1, -2)
    i = torch.arange(S, device=q.device)
    vis = torch.ones(S, S, dtype=torch.bool, device=q.device)
    if causal:
        vis &= i[None, :] <= i[:, None]
    if window > 0:
        vis &= i[None, :] > i[:, None] - window
    s3 = (s * scale).masked_fill(~vis, float("-inf"))
    p = torch.softmax(s3, -1)
    return s, p, p @ vx, vis, rep


def _tm(x):      # [B,H,S,d] -> token-major [B*S, H*d]
    B, H, S, d = x.shape
    return x.permute(0, 2, 1, 3).reshape(B * S, H * d).contiguous()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,S,Hq,Hkv,d,causal,window", [
    (1, 16, 4, 2, 16, True, 0), (2, 100, 4, 2, 32, True, 0), (1, 192, 4, 1, 128, True, 0),
    (1, 130, 2, 2, 64, False, 0), (1, 200, 2, 1, 64, True, 48), (1, 300, 8, 2, 128, True, 0),
    (1, 150, 2, 1, 256, True, 0), (1, 150, 2, 1, 256, True, 64), (1, 300, 6, 2, 96, True, 0), (2, 260, 3, 3, 96, False, 0), (1, 400, 2, 1, 96, True, 90),
    (2, 520, 8, 2, 64, True, 0), (1, 333, 4, 4, 64, False, 0),
    (1, 500, 4, 2, 128, True, 100), (1, 260, 4, 2, 128, False, 0), (2, 700, 8, 2, 128, True, 0), (1, 257, 4, 4, 128, True, 0)])
@pytest.mark.parametrize("mode", ["efficient", "explicit"])
def test_attention(ops, dtype, B, S, Hq, Hkv, d, causal, window, mode):
    if dtype == torch.bfloat16 and d < 32:
        pytest.skip("bf16 needs head_dim >= 32 (one 64-byte MFMA K chunk)")
    if dtype == torch.float32 and d == 96:
        pytest.skip("head_dim 96 exists on the bf16 32 x 32 kernels only (SigLIP's 72 padded)")
    tol = 3e-5 if dtype == torch.float32 else 3e-2
    q, k, v = rnd(B, Hq, S, d, dtype=dtype, seed=1), rnd(B, Hkv, S, d, dtype=dtype, seed=2), rnd(B, Hkv, S, d, dtype=dtype, seed=3)
    scale = d ** -0.5
    qt, kt, vt = _tm(q), _tm(k), _tm(v)
    v_t = ops.transpose_heads(vt, B, S, Hkv, d)
    assert torch.equal(v_t[..., :S], v.transpose(-1, -2))
    assert ops.attn_needs_transposed(qt, d) == (not (dtype == torch.bfloat16 and d in (64, 96, 128, 256)))
    o = torch.empty(B * S, Hq * d, dtype=dtype, device="cuda")
    lse = torch.empty(B, Hq, S, device="cuda")
    ops.attn_fwd(qt, kt, vt, v_t, o, lse, B, S, Hq, Hkv, d, scale, causal, window)
    s, p, o_ref, vis, rep = _attn_ref(f64(q), f64(k), f64(v), scale, causal, window)
    assert nmax(o, _tm(o_ref)) < tol
    lse_ref = torch.logsumexp((s * scale).masked_fill(~vis, float("-inf")), -1)
    assert nmax(lse, lse_ref) < (1e-5 if dtype == torch.float32 else 1e-2)

    E = dict(pv=1e-6, mask=1e-8, qk=1e-8) if mode == "explicit" else dict(pv=0.0, mask=0.0, qk=0.0)
    Go = rnd(B * S, Hq * d, dtype=dtype, seed=4)
    Gho, D = torch.empty_like(Go), torch.empty(B, Hq, S, device="cuda")
    ops.attn_bwd_prep(Go, o, Gho, D, B, S, Hq, d, E["pv"], 0.5)
    od = f64(o).reshape(B, S, Hq, d).permute(0, 2, 1, 3)
    God = f64(Go).reshape(B, S, Hq, d).permute(0, 2, 1, 3)
    Gho_ref = 0.5 * God * (od / (od + E["pv"]) if E["pv"] else 1.0)
    assert nmax(Gho, _tm(Gho_ref)) < tol
    # reference backward from the kernel's own (rounded) Gho so only the attention math is compared
    Gh = f64(Gho).reshape(B, S, Hq, d).permute(0, 2, 1, 3)
    assert nmax(D, (Gh * od).sum(-1)) < tol
    vx, kx = f64(v).repeat_interleave(rep, 1), f64(k).repeat_interleave(rep, 1)
    dP = Gh @ vx.transpose(-1, -2)
    dS3 = p * (dP - (dP * p).sum(-1, keepdim=True))
    s2 = s * scale
    f = torch.ones_like(s)
    if E["mask"]:
        f = f * s2 / (s2 + E["mask"])
    f = f * (s / (2 * s + E["qk"]) if E["qk"] else 0.5)
    Ghs = torch.where(vis, dS3 * scale * f, torch.zeros_like(s))
    dQ = Ghs @ kx
    dK = (Ghs.transpose(-1, -2) @ f64(q)).reshape(B, Hkv, rep, S, d).sum(2)
    dV = (p.transpose(-1, -2) @ Gh).reshape(B, Hkv, rep, S, d).sum(2)
    k_t, q_t, Gho_t = ops.transpose_heads(kt, B, S, Hkv, d), ops.transpose_heads(qt, B, S, Hq, d), ops.transpose_heads(Gho, B, S, Hq, d)
    dq = torch.empty_like(qt)
    ops.attn_bwd_dq(qt, kt, vt, k_t, Gho, lse, D, dq, B, S, Hq, Hkv, d, scale, E["mask"], E["qk"], causal, window)
    assert nmax(dq, _tm(dQ)) < tol * 3
    if mode == "efficient" and ops.attn_dq_d_ok(dtype, d):
        # lrp_attn_bwd_dq_d: the same dQ with D = rowsum(Gho (*) o) formed inside the kernel (no prep pass) and handed on to dK / dV
        D2, dq2 = torch.full_like(D, float("nan")), torch.full_like(dq, float("nan"))
        ops.attn_bwd_dq_d(qt, kt, vt, Gho, o, lse, D2, dq2, B, S, Hq, Hkv, d, scale, causal, window)
        assert not torch.isnan(D2).any() and torch.allclose(D2, D, rtol=1e-4, atol=1e-5 * float(D.abs().max()))
        assert not torch.isnan(dq2).any() and nmax(dq2, dq) < 1e-2 and nmax(dq2, _tm(dQ)) < tol * 3
        if d in (64, 128):
            # RoPE's backward on the way out of the dQ kernel / inside dK's group sum == rope_bwd on the stored gradients (ref: HF
            # apply_rotary_pos_emb's VJP; lxt/explicit/models/llama.py:226-260 with eps = 0), to one bf16 rounding
            inv = 1.0 / (10000.0 ** (torch.arange(0, d, 2, dtype=torch.float32) / d))
            fr = torch.arange(S, dtype=torch.float32)[:, None] * inv[None, :]
            cs, sn = torch.cat((fr, fr), -1).cos().cuda().contiguous(), torch.cat((fr, fr), -1).sin().cuda().contiguous()
            dq3 = torch.full_like(dq, float("nan"))
            ops.attn_bwd_dq_d(qt, kt, vt, Gho, o, lse, D2, dq3, B, S, Hq, Hkv, d, scale, causal, window, rope=(cs, sn))
            dq_rot = torch.empty_like(dq2)
            ops.rope_bwd(dq2, None, None, dq_rot, cs, sn, S, Hq, d, 0.0, 0.0)
            assert not torch.isnan(dq3).any() and nmax(dq3, dq_rot) < 1e-2
            xh = rnd(B * S, Hq * d, dtype=dtype, seed=9)                        # per-query-head dK stand-in
            red, red_rot, fused = torch.empty(B * S, Hkv * d, dtype=dtype, device="cuda"), torch.empty(B * S, Hkv * d, dtype=dtype, device="cuda"), \
                torch.full((B * S, Hkv * d), float("nan"), dtype=dtype, device="cuda")
            ops.gqa_reduce(xh, red, B * S, Hkv, rep, d)
            ops.rope_bwd(red, None, None, red_rot, cs, sn, S, Hkv, d, 0.0, 0.0)
            ops.gqa_reduce_rope(xh, fused, B * S, S, Hkv, rep, d, cs, sn)
            assert not torch.isnan(fused).any() and nmax(fused, red_rot) < 1e-2
    dk_h, dv_h = torch.empty_like(qt), torch.empty_like(qt)
    ops.attn_bwd_dkv(qt, kt, vt, q_t, Gho, Gho_t, lse, D, dk_h, dv_h, B, S, Hq, Hkv, d, scale, E["mask"], E["qk"], causal, window)
    dk, dv = torch.empty_like(kt), torch.empty_like(vt)
    ops.gqa_reduce(dk_h, dk, B * S, Hkv, rep, d)
    ops.gqa_reduce(dv_h, dv, B * S, Hkv, rep, d)
    assert nmax(dk, _tm(dK)) < tol * 3 and nmax(dv, _tm(dV)) < tol * 3


def _intervals(kind, B, S):
    """(lo, hi, causal_flag) int32 [B, S] for the mask families HF produces"""
    i = torch.arange(S)
    lo, hi = torch.zeros(B, S, dtype=torch.int32), torch.zeros(B, S, dtype=torch.int32)
    causal = True
    for b in range(B):
        if kind == "left_pad":           # causal + the first n_pad keys (and query rows) are padding
            n_pad = (7 + 13 * b) % (S // 2)
            lo[b], hi[b] = n_pad, i + 1
            hi[b, :n_pad] = 0             # padded query rows: empty interval
            lo[b, :n_pad] = 0
        elif kind == "right_pad":
            n = S - (5 + 11 * b) % (S // 2)
            lo[b], hi[b] = 0, torch.minimum(i + 1, torch.tensor(n))
        elif kind == "packed":           # block-diagonal causal (packed sequences)
            cuts = [0, S // 3 + b, (2 * S) // 3, S]
            for a, e in zip(cuts[:-1], cuts[1:]):
                lo[b, a:e] = a
            hi[b] = i + 1
        elif kind == "blocks_full":      # packed sequences WITHOUT a causal flag: bidirectional block-diagonal, the whole mask lives in the intervals
            causal = False               # (the kernels derive their tile ranges from the intervals: no tile outside a block is visited)
            cuts = [0, S // 3 + b, (2 * S) // 3, S]
            for a, e in zip(cuts[:-1], cuts[1:]):
                lo[b, a:e], hi[b, a:e] = a, e
        elif kind == "random":           # arbitrary, NON-monotone intervals (some empty): nothing about the interval structure may be assumed
            causal = False
            g = torch.Generator().manual_seed(77 + b)
            a = torch.randint(0, S, (S,), generator=g)
            e = torch.randint(0, S + 1, (S,), generator=g)
            lo[b], hi[b] = torch.minimum(a, e).int(), torch.maximum(a, e).int()
            hi[b, ::7] = lo[b, ::7]       # every 7th row sees nothing
        elif kind == "image_block":      # Gemma-3: causal text, bidirectional inside an image block -> not causal-bounded
            causal = False
            a, e = S // 4, S // 4 + S // 3
            lo[b], hi[b] = 0, i + 1
            hi[b, a:e] = e
    return lo, hi, causal


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("kind", ["left_pad", "right_pad", "packed", "image_block", "blocks_full", "random"])
@pytest.mark.parametrize("B,S,Hq,Hkv,d,window", [(2, 150, 4, 2, 64, 0), (2, 200, 2, 1, 128, 0), (1, 140, 2, 1, 256, 0), (2, 160, 2, 2, 64, 40), (2, 210, 2, 1, 96, 0),
                                                 (2, 330, 4, 2, 128, 0), (2, 300, 2, 1, 128, 70)])
def test_attention_row_intervals(ops, dtype, kind, B, S, Hq, Hkv, d, window):
    """per-row key intervals (padding / packed sequences / bidirectional blocks) against an fp64 eager attention with
    the same boolean mask; rows with an empty interval must come out as exact zeros and stay NaN-free"""
    if kind in ("image_block", "blocks_full", "random") and window:
        pytest.skip("bidirectional blocks are used with global layers")
    if dtype == torch.float32 and d == 96:
        pytest.skip("head_dim 96 exists on the bf16 32 x 32 kernels only")
    tol = 3e-5 if dtype == torch.float32 else 3e-2
    lo, hi, causal = _intervals(kind, B, S)
    row_iv = (lo.cuda().contiguous(), hi.cuda().contiguous())
    q, k, v = rnd(B, Hq, S, d, dtype=dtype, seed=1), rnd(B, Hkv, S, d, dtype=dtype, seed=2), rnd(B, Hkv, S, d, dtype=dtype, seed=3)
    scale = d ** -0.5
    qt, kt, vt = _tm(q), _tm(k), _tm(v)
    v_t = ops.transpose_heads(vt, B, S, Hkv, d)
    o = torch.empty(B * S, Hq * d, dtype=dtype, device="cuda")
    lse = torch.empty(B, Hq, S, device="cuda")
    ops.attn_fwd(qt, kt, vt, v_t, o, lse, B, S, Hq, Hkv, d, scale, causal, window, row_iv=row_iv)
    j = torch.arange(S, device="cuda")
    vis = (j[None, None, :] >= row_iv[0][:, :, None]) & (j[None, None, :] < row_iv[1][:, :, None])      # [B,S,S]
    if causal:
        vis &= (j[None, :] <= j[:, None])[None]
    if window:
        vis &= (j[None, :] > j[:, None] - window)[None]
    vis = vis[:, None]                                                                                      # [B,1,S,S]
    rep = Hq // Hkv
    qd, kx, vx = f64(q), f64(k).repeat_interleave(rep, 1), f64(v).repeat_interleave(rep, 1)
    s = qd @ kx.transpose(-1, -2)
    pr = torch.nan_to_num(torch.softmax((s * scale).masked_fill(~vis, float("-inf")), -1), nan=0.0)
    o_ref = pr @ vx
    assert torch.isfinite(o.float()).all()
    assert nmax(o, _tm(o_ref)) < tol
    empty = ~vis.any(-1).expand(B, Hq, S)
    if empty.any():
        assert (o.float().reshape(B, S, Hq, d).permute(0, 2, 1, 3)[empty] == 0).all()

    Go = rnd(B * S, Hq * d, dtype=dtype, seed=4)
    Gho, D = torch.empty_like(Go), torch.empty(B, Hq, S, device="cuda")
    ops.attn_bwd_prep(Go, o, Gho, D, B, S, Hq, d, 0.0, 0.5)
    Gh = f64(Gho).reshape(B, S, Hq, d).permute(0, 2, 1, 3)
    dP = Gh @ vx.transpose(-1, -2)
    dS3 = pr * (dP - (dP * pr).sum(-1, keepdim=True))
    Ghs = torch.where(vis, dS3 * scale * 0.5, torch.zeros_like(s))
    dQ = Ghs @ kx
    dK = (Ghs.transpose(-1, -2) @ qd).reshape(B, Hkv, rep, S, d).sum(2)
    dV = (pr.transpose(-1, -2) @ Gh).reshape(B, Hkv, rep, S, d).sum(2)
    k_t, q_t, Gho_t = ops.transpose_heads(kt, B, S, Hkv, d), ops.transpose_heads(qt, B, S, Hq, d), ops.transpose_heads(Gho, B, S, Hq, d)
    dq = torch.empty_like(qt)
    ops.attn_bwd_dq(qt, kt, vt, k_t, Gho, lse, D, dq, B, S, Hq, Hkv, d, scale, 0.0, 0.0, causal, window, row_iv=row_iv)
    dk_h, dv_h = torch.empty_like(qt), torch.empty_like(qt)
    ops.attn_bwd_dkv(qt, kt, vt, q_t, Gho, Gho_t, lse, D, dk_h, dv_h, B, S, Hq, Hkv, d, scale, 0.0, 0.0, causal, window, row_iv=row_iv)
    dk, dv = torch.empty_like(kt), torch.empty_like(vt)
    ops.gqa_reduce(dk_h, dk, B * S, Hkv, rep, d)
    ops.gqa_reduce(dv_h, dv, B * S, Hkv, rep, d)
    for t in (dq, dk, dv):
        assert torch.isfinite(t.float()).all()
    assert nmax(dq, _tm(dQ)) < tol * 3
    assert nmax(dk, _tm(dK)) < tol * 3 and nmax(dv, _tm(dV)) < tol * 3


def test_rule_goldens_attention_pieces(ops):
    """matmul rule (Prop 3.3) and uniform-eps P.V rule against the reference's outputs, composed
    from the C-ABI GEMM + eps-scale exactly as the explicit API layer does."""
    fx = load("rules.npz")
    a, b, g = t(fx["mm_a"]).cuda(), t(fx["mm_b"]).cuda(), t(fx["mm_g"]).cuda()
    bt = ops.transpose(b)
    o = ops.gemm_nt(a, bt)
    s = ops.eps_scale(ops.mul(o, g), o, 2.0, 1e-8, relevance=True)
    Ra = ops.mul(ops.gemm_nt(s, b), a)
    Rb = ops.mul(ops.gemm_nt(ops.transpose(a), ops.transpose(s)), b)
    assert nmax(Ra, fx["mm_Ra"]) < 2e-5 and nmax(Rb, fx["mm_Rb"]) < 2e-5
    p, v, g = t(fx["pv_p"]).cuda(), t(fx["pv_v"]).cuda(), t(fx["pv_g"]).cuda()
    o = ops.gemm_nt(p, ops.transpose(v))
    s = ops.eps_scale(ops.mul(o, g), o, 1.0, 1e-6, relevance=True)
    s = ops.mul(s, torch.full_like(s, 0.5))
    Rp = ops.mul(ops.gemm_nt(s, v), p)
    Rv = ops.mul(ops.gemm_nt(ops.transpose(p), ops.transpose(s)), v)
    assert nmax(Rp, fx["pv_Rp"]) < 2e-5 and nmax(Rv, fx["pv_Rv"]) < 2e-5
