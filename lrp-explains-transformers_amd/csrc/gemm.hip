// gemm.hip -- NT GEMM on MFMA for the Linear eps-rule (K1) and every other contraction that
// is not attention.   C[M,N] = A[M,K] . B[N,K]^T (+bias), fp32 accumulate.
//
// Kernels (gfx950, wave64), chosen by launch_fast():
//   * gemm_pp.hip: 256 x 256 tile, 8 waves in two ping-pong groups, buffer_load .. lds staging -- the big bf16 problems (M = B*S rows);
//   * gemm_nt_glds_kernel (below): TBM x TBN tile (128x128 / 64x64 / 32x32 / 256x256), direct-to-LDS staging (global_load_lds_dwordx4),
//     K step = 128 BYTES per row for both dtypes (64 bf16 / 32 fp32), two LDS stages, one barrier per K step; fp32 operands accumulate
//     in blocks (an fp32 MFMA chain is a sequential fmaf chain);
//   * gemm_nt_kernel (below): 128 x 128, register-staged, for K that is not a whole number of 128-byte steps (ragged K).
//   Common: operands swapped (mma(Bfrag, Afrag)) so a lane owns 4 CONSECUTIVE output columns of one row; LDS rows are 128 B with
//   the 16-byte chunk index XOR-swizzled with (row & 7): every ds_read_b128 lane group touches 16 distinct 16-byte bank slots;
//   XCD-aware block remap + grouped tile order keep an operand panel in one XCD's L2.
#include "common.hpp"

// gemm_pp.hip: 8-wave ping-pong kernel for the big bf16 problems (>= 190 tiles of 256 x 256; 32-bit buffer offsets)
int lrp_launch_gemm_pp(const void* A, const void* B, void* C, const void* bias, int M, int N, int K, int64_t lda, int64_t ldb,
                       int64_t ldc, int out_dtype, int nn, int splits, int kt_per_split, int64_t slab_stride, hipStream_t st);

int lrp_launch_gemm_pp_gated_fwd(const void* x, const void* Wgu, const float* rs, void* coef, void* m, int M, int I, int K, int64_t ldx,
                                 int64_t ldw, int64_t ldcoef, int64_t ldm, float eps_g, float eps_lin, int act, hipStream_t st);
int lrp_launch_gemm_pp_gated_bwd(const void* Adn, const void* Wdn, const void* coef, void* Agu, int M, int I, int K, int64_t lda,
                                 int64_t ldw, int64_t ldcoef, int64_t ldagu, hipStream_t st);

int lrp_launch_gemm_pp_res_ssq(const void* x, const void* W, const void* res, void* out, float* ssq, int M, int N, int K, int64_t ldx,
                               int64_t ldw, int64_t ldres, int64_t ldout, int64_t ldssq, void* raw, int64_t ldraw, hipStream_t st);
int lrp_launch_gemm_pp_nt_rs(const void* x, const void* W, const float* rs, void* out, int M, int N, int K, int64_t ldx, int64_t ldw,
                             int64_t ldout, hipStream_t st);
int lrp_launch_gemm_pp_nn_rs(const void* s, const void* W, const float* rs, void* out, int M, int N, int K, int64_t lds_, int64_t ldw,
                             int64_t ldout, hipStream_t st);
int lrp_launch_gemm_pp_nt_rs_rope(const void* x, const void* W, const float* rs, const float* cos, const float* sin, void* out, int M, int N, int K,
                                  int64_t ldx, int64_t ldw, int64_t ldout, int seq, int rope_cols, hipStream_t st);
int lrp_launch_gemm_pp_nn_rs_res(const void* s, const void* W, const float* rs, const void* res, void* out, int M, int N, int K, int64_t lds_,
                                 int64_t ldw, int64_t ldres, int64_t ldout, hipStream_t st);

namespace {

constexpr int BM = 128, BN = 128, KB = 128;     // KB: bytes of K per stage and per row
constexpr int NT = 256;

template <typename T, typename TO>
__global__ __launch_bounds__(NT, 2) void gemm_nt_kernel(
    const T* __restrict__ A, const T* __restrict__ B, TO* __restrict__ C, const T* __restrict__ bias,
    int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, int64_t sA, int64_t sB, int64_t sC,
    int tiles_m, int tiles_n) {
    constexpr int EPC = 16 / sizeof(T);          // elements per 16-byte chunk
    constexpr int KE = KB / sizeof(T);           // elements of K per stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sAb = smem;                            // [2][BM][KB]
    char* sBb = smem + 2 * BM * KB;              // [2][BN][KB]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntile = tiles_m * tiles_n;
    const int t = xcd_remap(blockIdx.x, ntile);
    const int tm = t % tiles_m, tn = t / tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const int bz = blockIdx.y;
    A += (int64_t)bz * sA;
    B += (int64_t)bz * sB;
    C += (int64_t)bz * sC;

    // staging geometry: thread -> (row = tid>>3 + 32*p, chunk = tid&7), p = 0..3
    const int srow = tid >> 3, schunk = tid & 7;
    const int nkt = (K + KE - 1) / KE;

    u32x4 ra[4], rb[4];
    auto gload = [&](int kt) {
        const int kbase = kt * KE + schunk * EPC;
        const bool kok = kbase < K;              // K % EPC == 0 is a precondition (whole chunks)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int r = srow + 32 * p;
            const int gm = m0 + r, gn = n0 + r;
            ra[p] = (kok && gm < M) ? *reinterpret_cast<const u32x4*>(A + (int64_t)gm * lda + kbase)
                                    : u32x4{0, 0, 0, 0};
            rb[p] = (kok && gn < N) ? *reinterpret_cast<const u32x4*>(B + (int64_t)gn * ldb + kbase)
                                    : u32x4{0, 0, 0, 0};
        }
    };
    auto swrite = [&](int buf) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int r = srow + 32 * p;
            const int off = r * KB + ((schunk ^ (r & 7)) << 4);
            *reinterpret_cast<u32x4*>(sAb + buf * BM * KB + off) = ra[p];
            *reinterpret_cast<u32x4*>(sBb + buf * BN * KB + off) = rb[p];
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    gload(0);
    swrite(0);
    __syncthreads();

    const int frow = lane & 15, fq = lane >> 4;
    typedef typename Mma16<T>::frag frag_t;
    int cur = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) gload(kt + 1);
        const char* pa = sAb + cur * BM * KB + (wm * 64) * KB;
        const char* pb = sBb + cur * BN * KB + (wn * 64) * KB;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            frag_t fa[4], fb[4];
            const int c = kk * 4 + fq;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = i * 16 + frow;     // (r & 7) == (frow & 7)
                const int off = r * KB + ((c ^ (frow & 7)) << 4);
                fa[i] = *reinterpret_cast<const frag_t*>(pa + off);
                fb[i] = *reinterpret_cast<const frag_t*>(pb + off);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = Mma16<T>::mma(fb[j], fa[i], acc[i][j]);
        }
        if (kt + 1 < nkt) swrite(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // epilogue: lane owns C[m = i*16 + (l&15)][n = j*16 + (l>>4)*4 .. +4]
    const bool vec_ok = ((ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int gm = m0 + wm * 64 + i * 16 + frow;
        if (gm >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gn = n0 + wn * 64 + j * 16 + fq * 4;
            if (gn >= N) continue;
            f32x4 v = acc[i][j];
            if (bias) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (gn + r < N) v[r] += to_f32(bias[gn + r]);
            }
            TO* dst = C + (int64_t)gm * ldc + gn;
            if (vec_ok && gn + 3 < N) {
                if constexpr (sizeof(TO) == 4) {
                    *reinterpret_cast<f32x4*>(dst) = v;
                } else {
                    bf16x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (bf16_t)v[r];
                    *reinterpret_cast<bf16x4*>(dst) = o;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (gn + r < N) dst[r] = from_f32<TO>(v[r]);
            }
        }
    }
}


// =================================================================================================
// Fast path: direct-to-LDS staging (global_load_lds_dwordx4, no VGPR round trip, no ds_write pass).
// One wave instruction deposits 1 KiB = 8 tile rows x 128 B, lane l -> LDS byte l*16, i.e. row
// (l>>3), slot (l&7).  The XOR swizzle therefore goes on the SOURCE address: lane l fetches global
// chunk (l&7)^(l>>3) of its row, and the fragment reads apply the same involution (slot = c ^ (row&7)).
// Tile BM x BN x (128 B of K), WM x WN waves, two LDS stages, one barrier per K step: the loads of
// step t+1 are in flight under the MFMAs of step t (__syncthreads drains the LDS-DMA queue).
// Requires K to be a multiple of the 128-byte K step; M, N arbitrary (row indices are clamped, the
// duplicated rows only feed outputs that are never stored).
// =================================================================================================
template <typename T, typename TO, int TBM, int TBN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN >= 16 ? 4 : 2)) void gemm_nt_glds_kernel(
    const T* __restrict__ A, const T* __restrict__ B, TO* __restrict__ C, const T* __restrict__ bias,
    int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, int64_t sA, int64_t sB, int64_t sC,
    int tiles_m, int tiles_n) {
    constexpr int NW = WM * WN;
    constexpr int EPC = 16 / sizeof(T), KE = KB / sizeof(T);
    constexpr int SM = TBM / WM, SN = TBN / WN;      // wave sub-tile
    constexpr int FM = SM / 16, FN = SN / 16;        // 16x16 MFMA tiles per wave
    constexpr int GA = TBM / 8 / NW, GB = TBN / 8 / NW;   // 1-KiB row groups per wave per operand
    static_assert(TBM % (8 * NW) == 0 && TBN % (8 * NW) == 0, "tile rows must split over the waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STAGE = (TBM + TBN) * KB;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int ntile = tiles_m * tiles_n;
    const int t = xcd_remap(blockIdx.x, ntile);
    int tm, tn;
    grouped_tile(t, tiles_m, tiles_n, tm, tn);
    const int m0 = tm * TBM, n0 = tn * TBN;
    const int bz = blockIdx.y;
    A += (int64_t)bz * sA;
    B += (int64_t)bz * sB;
    C += (int64_t)bz * sC;
    const int nkt = K / KE;

    // per-lane source pointers for the row groups this wave stages (row clamp = bounds handling)
    const int lrow = lane >> 3, lchunk = (lane & 7) ^ (lane >> 3);
    const T* pa[GA];
    const T* pb[GB];
#pragma unroll
    for (int i = 0; i < GA; ++i) {
        int r = m0 + (wave * GA + i) * 8 + lrow;
        r = r < M ? r : M - 1;
        pa[i] = A + (int64_t)r * lda + lchunk * EPC;
    }
#pragma unroll
    for (int i = 0; i < GB; ++i) {
        int r = n0 + (wave * GB + i) * 8 + lrow;
        r = r < N ? r : N - 1;
        pb[i] = B + (int64_t)r * ldb + lchunk * EPC;
    }
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef const __attribute__((address_space(1))) void* glb_ptr_t;
    auto stage = [&](int kt, int buf) {
        char* sa = smem + buf * STAGE + (wave * GA) * 1024;
        char* sb = smem + buf * STAGE + TBM * KB + (wave * GB) * 1024;
#pragma unroll
        for (int i = 0; i < GA; ++i)
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(pa[i] + (int64_t)kt * KE), (lds_ptr_t)(sa + i * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < GB; ++i)
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(pb[i] + (int64_t)kt * KE), (lds_ptr_t)(sb + i * 1024), 16, 0, 0);
    };

    // fp32 operands (the parity path): BLOCKED accumulation.  An fp32 MFMA chain is a k-ordered fmaf chain (one rounding per
    // product, no wider internal accumulator), so a plain K loop is a sequential sum of K terms -- ~3x the rounding error of a
    // CPU BLAS (which keeps ~100 partial sums), and the explicit rules' z/(z+eps) poles amplify exactly that error
    // (tools/explicit_forward_error.py).  Here two K steps (64 terms) go into a short accumulator that is folded into the long
    // one: error ~ eps (sqrt(64) + sqrt(K/64)) instead of eps sqrt(K).  bf16 operands keep the single accumulator.
    constexpr bool BLOCKED = (sizeof(T) == 4) && (FM * FN <= 16);
    f32x4 acc[FM][FN], lo[BLOCKED ? FM : 1][BLOCKED ? FN : 1];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (BLOCKED) lo[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }

    stage(0, 0);
    __syncthreads();

    const int frow = lane & 15, fq = lane >> 4;
    typedef typename Mma16<T>::frag frag_t;
    int cur = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) stage(kt + 1, cur ^ 1);
        const char* pas = smem + cur * STAGE + (wm * SM) * KB;
        const char* pbs = smem + cur * STAGE + TBM * KB + (wn * SN) * KB;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            frag_t fa[FM], fb[FN];
            const int off = ((kk * 4 + fq) ^ (frow & 7)) << 4;
#pragma unroll
            for (int i = 0; i < FM; ++i) fa[i] = *reinterpret_cast<const frag_t*>(pas + (i * 16 + frow) * KB + off);
#pragma unroll
            for (int j = 0; j < FN; ++j) fb[j] = *reinterpret_cast<const frag_t*>(pbs + (j * 16 + frow) * KB + off);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    if constexpr (BLOCKED) lo[i][j] = Mma16<T>::mma(fb[j], fa[i], lo[i][j]);
                    else acc[i][j] = Mma16<T>::mma(fb[j], fa[i], acc[i][j]);
                }
        }
        if constexpr (BLOCKED) {
            if ((kt & 1) == 1 || kt + 1 == nkt) {
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j) { acc[i][j] += lo[i][j]; lo[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            }
        }
        __syncthreads();        // drains the LDS-DMA queue: tile kt+1 has landed, stage `cur` is free
        cur ^= 1;
    }

    const bool vec_ok = ((ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0);
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int gm = m0 + wm * SM + i * 16 + frow;
        if (gm >= M) continue;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int gn = n0 + wn * SN + j * 16 + fq * 4;
            if (gn >= N) continue;
            f32x4 v = acc[i][j];
            if (bias) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (gn + r < N) v[r] += to_f32(bias[gn + r]);
            }
            TO* dst = C + (int64_t)gm * ldc + gn;
            if (vec_ok && gn + 3 < N) {
                if constexpr (sizeof(TO) == 4) {
                    *reinterpret_cast<f32x4*>(dst) = v;
                } else {
                    bf16x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (bf16_t)v[r];
                    *reinterpret_cast<bf16x4*>(dst) = o;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (gn + r < N) dst[r] = from_f32<TO>(v[r]);
            }
        }
    }
}

template <typename T, typename TO, int TBM, int TBN, int WM, int WN>
int launch_glds(const void* A, const void* B, void* C, const void* bias, int M, int N, int K, int64_t lda, int64_t ldb,
                int64_t ldc, int batch, int64_t sA, int64_t sB, int64_t sC, hipStream_t st) {
    const int tiles_m = (M + TBM - 1) / TBM, tiles_n = (N + TBN - 1) / TBN;
    dim3 grid(tiles_m * tiles_n, batch), block(64 * WM * WN);
    const size_t lds = 2 * (size_t)(TBM + TBN) * KB;
    auto kern = gemm_nt_glds_kernel<T, TO, TBM, TBN, WM, WN>;
    LRP_SET_MAX_LDS(kern, lds);
    hipLaunchKernelGGL(kern, grid, block, lds, st, (const T*)A, (const T*)B, (TO*)C, (const T*)bias, M, N, K, lda, ldb, ldc,
                       sA, sB, sC, tiles_m, tiles_n);
    return lrp_check_launch();
}


// Tile selection (measured on MI355X, profiles/r01_gemm_tiles.txt, r01..r03_gemm_experiments.txt): the 256x256 8-wave ping-pong
// kernel of gemm_pp.hip once the problem yields >= ~190 tiles (every CU busy), the 128x128 4-wave kernel (2 workgroups per
// CU) below that.  Everything else that was tried is logged in profiles/ (the code of the losing variants is not kept).
// rows of A one launch of the ping-pong kernel can address (32-bit buffer byte offsets: rows * lda < 2^30 elements), a multiple of 256;
// problems with more rows are issued as several launches over row chunks (rows are independent) -- no silent change of kernel or layout
inline int pp_row_chunk(int64_t lda) {
    const int64_t r = (((1ll << 30) - 1) / (lda > 0 ? lda : 1)) / 256 * 256;
    return (int)(r < 256 ? 0 : (r > (1 << 30) ? (1 << 30) : r));
}
template <typename T, typename TO>
int launch_fast(const void* A, const void* B, void* C, const void* bias, int M, int N, int K, int64_t lda, int64_t ldb,
                int64_t ldc, int batch, int64_t sA, int64_t sB, int64_t sC, hipStream_t st) {
    const int64_t tiles256 = (int64_t)((M + 255) / 256) * ((N + 255) / 256) * batch;
    constexpr int KE = KB / (int)sizeof(T);
    // fp32 is the parity path, not the throughput path (1/16 of the bf16 MFMA rate): always the 128x128 kernel, whose fp32
    // instantiation accumulates in blocks
    if (sizeof(T) == 2 && tiles256 >= 190) {
        if (batch == 1 && K / KE >= 2 && pp_row_chunk(lda) > 0 && (int64_t)N * ldb < (1ll << 30)) {
            const int chunk = pp_row_chunk(lda);
            for (int m0 = 0; m0 < M; m0 += chunk) {
                const int rc = lrp_launch_gemm_pp((const T*)A + (int64_t)m0 * lda, B, (TO*)C + (int64_t)m0 * ldc, bias, M - m0 < chunk ? M - m0 : chunk,
                                                  N, K, lda, ldb, ldc, sizeof(TO) == 4 ? LRP_F32 : LRP_BF16, 0, 1, K / KE, 0, st);
                if (rc != LRP_OK) return rc;
            }
            return LRP_OK;
        }
        return launch_glds<T, TO, 256, 256, 4, 4>(A, B, C, bias, M, N, K, lda, ldb, ldc, batch, sA, sB, sC, st);
    }
    // small problems (BERT-sized M = 128: 6..24 tiles of 128x128 on 256 CUs, each walking the whole K alone): 64x64 or 32x32
    // tiles give 4x / 16x the workgroups; same K order per output element, so the results do not change
    const int64_t tiles128 = (int64_t)((M + 127) / 128) * ((N + 127) / 128) * batch;
    if (tiles128 < 96) {
        const int64_t tiles64 = (int64_t)((M + 63) / 64) * ((N + 63) / 64) * batch;
        if (tiles64 < 96) return launch_glds<T, TO, 32, 32, 1, 1>(A, B, C, bias, M, N, K, lda, ldb, ldc, batch, sA, sB, sC, st);
        return launch_glds<T, TO, 64, 64, 2, 2>(A, B, C, bias, M, N, K, lda, ldb, ldc, batch, sA, sB, sC, st);
    }
    return launch_glds<T, TO, 128, 128, 2, 2>(A, B, C, bias, M, N, K, lda, ldb, ldc, batch, sA, sB, sC, st);
}

template <typename T, typename TO>
int launch_gemm(const void* A, const void* B, void* C, const void* bias, int M, int N, int K,
                int64_t lda, int64_t ldb, int64_t ldc, int batch, int64_t sA, int64_t sB,
                int64_t sC, hipStream_t st) {
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    dim3 grid(tiles_m * tiles_n, batch), block(NT);
    const size_t lds = 2 * (BM + BN) * KB;
    LRP_SET_MAX_LDS((&gemm_nt_kernel<T, TO>), lds);
    hipLaunchKernelGGL((gemm_nt_kernel<T, TO>), grid, block, lds, st, (const T*)A, (const T*)B,
                       (TO*)C, (const T*)bias, M, N, K, lda, ldb, ldc, sA, sB, sC, tiles_m, tiles_n);
    return lrp_check_launch();
}

}  // namespace

extern "C" int lrp_gemm_nt(const void* A, const void* B, void* C, const void* bias, int M, int N,
                           int K, int64_t lda, int64_t ldb, int64_t ldc, int batch, int64_t sA,
                           int64_t sB, int64_t sC, int dtype, int out_dtype, void* stream) {
    if (!A || !B || !C || M < 0 || N < 0 || K < 0 || batch < 1) return LRP_EINVAL;
    if (M == 0 || N == 0) return LRP_OK;
    const int epc = (dtype == LRP_F32) ? 4 : 8;
    if (dtype != LRP_F32 && dtype != LRP_BF16) return LRP_EINVAL;
    if ((K % epc) || (lda % epc) || (ldb % epc) || (sA % epc) || (sB % epc)) return LRP_EALIGN;
    if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(B) & 15)) return LRP_EALIGN;
    if (batch > 65535) return LRP_ESHAPE;
    hipStream_t st = (hipStream_t)stream;
    // fast path (direct-to-LDS staging) when K is a whole number of 128-byte steps, else the register-staged kernel
    const int ke = (dtype == LRP_F32) ? 32 : 64;
    if ((K % ke) == 0 && M >= 1 && N >= 1) {
        if (dtype == LRP_F32) {
            if (out_dtype != LRP_F32) return LRP_EINVAL;
            return launch_fast<float, float>(A, B, C, bias, M, N, K, lda, ldb, ldc, batch, sA, sB, sC, st);
        }
        if (out_dtype == LRP_F32) return launch_fast<bf16_t, float>(A, B, C, bias, M, N, K, lda, ldb, ldc, batch, sA, sB, sC, st);
        if (out_dtype == LRP_BF16) return launch_fast<bf16_t, bf16_t>(A, B, C, bias, M, N, K, lda, ldb, ldc, batch, sA, sB, sC, st);
        return LRP_EINVAL;
    }
    if (dtype == LRP_F32) {
        if (out_dtype != LRP_F32) return LRP_EINVAL;
        return launch_gemm<float, float>(A, B, C, bias, M, N, K, lda, ldb, ldc, batch, sA, sB, sC, st);
    }
    if (out_dtype == LRP_F32)
        return launch_gemm<bf16_t, float>(A, B, C, bias, M, N, K, lda, ldb, ldc, batch, sA, sB, sC, st);
    if (out_dtype == LRP_BF16)
        return launch_gemm<bf16_t, bf16_t>(A, B, C, bias, M, N, K, lda, ldb, ldc, batch, sA, sB, sC, st);
    return LRP_EINVAL;
}

// =================================================================================================
// NN form and the skinny (split-K) dispatch of the ping-pong kernel (gemm_pp.hip)
// =================================================================================================
namespace {

// out[m][n] = sum_s slab[s][m][n] (+ bias[n]), cast to TO; slabs are [M][ldw] fp32, ldw % 4 == 0
template <typename TO>
__global__ void splitk_reduce_kernel(const float* __restrict__ ws, const bf16_t* __restrict__ bias, TO* __restrict__ out, int M, int N,
                                     int64_t ldw, int64_t ldo, int splits, int64_t slab) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // one float4 of one row
    const int nq = (N + 3) / 4;
    if (q >= (int64_t)M * nq) return;
    const int m = (int)(q / nq), n = (int)(q % nq) * 4;
    const float* p = ws + (int64_t)m * ldw + n;
    // eight slab loads in flight per lane (the kernel is pure latency: a few MB out of L2 / Infinity Cache right behind the GEMM that wrote
    // them), summed in slab order: the result does not depend on the unroll
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int s_ = 0;
    for (; s_ + 8 <= splits; s_ += 8) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(p + (int64_t)(s_ + u) * slab);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; s_ + 2 <= splits; s_ += 2) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(p + (int64_t)s_ * slab), v1 = *reinterpret_cast<const f32x4*>(p + (int64_t)(s_ + 1) * slab);
        acc += v0;
        acc += v1;
    }
    if (s_ < splits) acc += *reinterpret_cast<const f32x4*>(p + (int64_t)s_ * slab);
    if (n + 3 < N && (ldo & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
        if constexpr (sizeof(TO) == 2) {
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (bf16_t)(acc[e] + (bias ? to_f32(bias[n + e]) : 0.f));
            *reinterpret_cast<bf16x4*>(out + (int64_t)m * ldo + n) = o;
        } else {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = acc[e] + (bias ? to_f32(bias[n + e]) : 0.f);
            *reinterpret_cast<f32x4*>(out + (int64_t)m * ldo + n) = o;
        }
        return;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (n + e < N) out[(int64_t)m * ldo + n + e] = from_f32<TO>(acc[e] + (bias ? to_f32(bias[n + e]) : 0.f));
}

bool pp_ok(int M, int N, int K, int64_t lda, int64_t ldb, int nn) {
    const int64_t brows = nn ? K : N;
    return (K % 64) == 0 && K >= 128 && pp_row_chunk(lda) > 0 && brows * ldb < (1ll << 30);
}

// split policy of the split-K path: ONE round of workgroups (one workgroup per CU: 128 KiB of LDS each) that covers as many of the 256 CUs
// as the tile count allows -- a second, partial round would double the time of a kernel that only streams the weight --, at least 2 K
// tiles per split.  M <= 256 rows: one row of tiles (the HBM-bound regime of the Linear eps-rule).  More rows (round 3: M = 2048, one
// prompt per step): problems of <= 128 tiles, which would leave half of the chip idle, are split over K the same way.
int skinny_splits(int M, int N, int K) {
    const int tiles = ((M + 255) / 256) * ((N + 255) / 256), nkt = K / 64;
    // 1 ... 1.5 rounds of the 256 CUs (Gemma-3-4B: 8192 x 2560 = 320 tiles): the second round would run on a quarter of the chip for a
    // full tile time; with the K range halved it is 2.5 half-length rounds.  Pays once a tile's K loop outweighs the slab round trip.
    if (tiles > 256 && tiles <= 384 && nkt >= 128) return 2;
    int s_ = 256 / tiles;
    if (s_ > nkt / 2) s_ = nkt / 2;
    return s_ < 1 ? 1 : s_;
}

}  // namespace

// the slab reduction alone (linear_stream.hip's dgrad splits its contraction range the same way)
int lrp_launch_splitk_reduce(const float* ws, void* out, int M, int N, int64_t ldw, int64_t ldo, int splits, int64_t slab, int out_dtype,
                             hipStream_t st) {
    const int64_t nq = (int64_t)M * ((N + 3) / 4);
    dim3 grid((unsigned)((nq + 255) / 256)), block(256);
    if (out_dtype == LRP_F32)
        hipLaunchKernelGGL((splitk_reduce_kernel<float>), grid, block, 0, st, ws, (const bf16_t*)nullptr, (float*)out, M, N, ldw, ldo, splits, slab);
    else
        hipLaunchKernelGGL((splitk_reduce_kernel<bf16_t>), grid, block, 0, st, ws, (const bf16_t*)nullptr, (bf16_t*)out, M, N, ldw, ldo, splits, slab);
    return lrp_check_launch();
}

extern "C" int lrp_gemm_nn(const void* A, const void* Bt, void* C, const void* bias, int M, int N, int K, int64_t lda, int64_t ldb,
                           int64_t ldc, int dtype, int out_dtype, void* stream) {
    if (!A || !Bt || !C || M < 0 || N < 0 || K < 0) return LRP_EINVAL;
    if (M == 0 || N == 0) return LRP_OK;
    if (dtype != LRP_BF16 || (out_dtype != LRP_BF16 && out_dtype != LRP_F32)) return LRP_ESHAPE;
    if ((lda % 8) || (ldb % 8) || (reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(Bt) & 15)) return LRP_EALIGN;
    if (!pp_ok(M, N, K, lda, ldb, 1)) return LRP_ESHAPE;
    const int chunk = pp_row_chunk(lda);
    const int64_t osz = (out_dtype == LRP_F32) ? 4 : 2;
    for (int m0 = 0; m0 < M; m0 += chunk) {
        const int rc = lrp_launch_gemm_pp((const char*)A + (int64_t)m0 * lda * 2, Bt, (char*)C + (int64_t)m0 * ldc * osz, bias,
                                          M - m0 < chunk ? M - m0 : chunk, N, K, lda, ldb, ldc, out_dtype, 1, 1, K / 64, 0, (hipStream_t)stream);
        if (rc != LRP_OK) return rc;
    }
    return LRP_OK;
}

extern "C" int lrp_gemm_skinny_splits(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K < 128) return 1;
    return skinny_splits(M, N, K);
}

extern "C" int64_t lrp_gemm_skinny_ws(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K < 128) return 0;
    const int64_t ldw = (N + 3) / 4 * 4;
    return (int64_t)skinny_splits(M, N, K) * M * ldw * 4;
}

extern "C" int lrp_gemm_skinny(const void* A, const void* B, void* C, const void* bias, int M, int N, int K, int64_t lda, int64_t ldb,
                               int64_t ldc, int nn, int dtype, int out_dtype, void* ws, void* stream) {
    if (!A || !B || !C || !ws || M <= 0 || N <= 0 || K <= 0) return LRP_EINVAL;
    if (dtype != LRP_BF16 || (out_dtype != LRP_BF16 && out_dtype != LRP_F32)) return LRP_ESHAPE;
    if ((lda % 8) || (ldb % 8) || (reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(B) & 15) ||
        (reinterpret_cast<uintptr_t>(ws) & 15)) return LRP_EALIGN;
    if (!pp_ok(M, N, K, lda, ldb, nn) || M > pp_row_chunk(lda)) return LRP_ESHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int splits = skinny_splits(M, N, K), nkt = K / 64;
    int per = (nkt + splits - 1) / splits;
    int used = (nkt + per - 1) / per;
    while (used > 1 && nkt - (used - 1) * per < 2) {                         // the kernel needs >= 2 K tiles in every split, the last one too
        ++per;                                                              // (K = 2560: 40 tiles in 16 splits of 3 would leave 1)
        used = (nkt + per - 1) / per;
    }
    const int64_t ldw = (N + 3) / 4 * 4, slab = (int64_t)M * ldw;
    // one split (the tile count alone fills the chip: the LM head): the plain kernel writes the output directly
    if (used == 1) return lrp_launch_gemm_pp(A, B, C, bias, M, N, K, lda, ldb, ldc, out_dtype, nn, 1, nkt, 0, st);
    int rc = lrp_launch_gemm_pp(A, B, ws, nullptr, M, N, K, lda, ldb, ldw, LRP_F32, nn, used, per, slab, st);
    if (rc != LRP_OK) return rc;
    const int64_t nq = (int64_t)M * ((N + 3) / 4);
    dim3 grid((unsigned)((nq + 255) / 256)), block(256);
    if (out_dtype == LRP_F32)
        hipLaunchKernelGGL((splitk_reduce_kernel<float>), grid, block, 0, st, (const float*)ws, (const bf16_t*)bias, (float*)C, M, N, ldw, ldc, used, slab);
    else
        hipLaunchKernelGGL((splitk_reduce_kernel<bf16_t>), grid, block, 0, st, (const float*)ws, (const bf16_t*)bias, (bf16_t*)C, M, N, ldw, ldc, used, slab);
    return lrp_check_launch();
}

// =================================================================================================
// gated-MLP rules fused into the GEMMs around them (interleaved gate/up layout, include/lrp_hip.h)
// =================================================================================================
namespace {
bool gated_fused_ok(int M, int Ncols, int K, int I, int64_t lda, int64_t ldb, int nn, int act) {
    const int64_t tiles = (int64_t)((M + 255) / 256) * ((Ncols + 255) / 256);
    return tiles >= 190 && (I % LRP_GATED_IL) == 0 && (act == LRP_ACT_SILU || act == LRP_ACT_GELU_TANH) && pp_ok(M, Ncols, K, lda, ldb, nn);
}
}  // namespace

// ---- round 6: the fused form stashes the backward's COEFFICIENTS (include/lrp_hip.h).  Both launches of a layer must be problems the
// ping-pong kernel's fused epilogues take: the gate/up forward [M, 2 I] over K = hidden (NT) and the down-projection dgrad [M, I] over hidden (NN)
extern "C" int lrp_gemm_gated_coef_ok(int M, int I, int H, int64_t ldx, int64_t ldwgu, int64_t lda, int64_t ldwd, int act, int dtype) {
    if (dtype != LRP_BF16 || M <= 0 || I <= 0 || H <= 0 || (ldx % 8) || (ldwgu % 8) || (lda % 8) || (ldwd % 8)) return 0;
    return (gated_fused_ok(M, 2 * I, H, I, ldx, ldwgu, 0, act) && gated_fused_ok(M, I, H, I, lda, ldwd, 1, act)) ? 1 : 0;
}

extern "C" int lrp_gemm_gated_fwd_coef(const void* x, const void* Wgu, const float* rs, void* coef, void* m, int M, int I, int K, int64_t ldx,
                                       int64_t ldw, int64_t ldcoef, int64_t ldm, float eps_g, float eps_lin, int act, int dtype, void* stream) {
    if (!x || !Wgu || !coef || !m || M < 0 || I < 0 || K < 0 || act < 0 || act > 3 || eps_g < 0.f || eps_lin < 0.f) return LRP_EINVAL;
    if (M == 0 || I == 0) return LRP_OK;
    if (dtype != LRP_BF16 || (ldx % 8) || (ldw % 8) || !gated_fused_ok(M, 2 * I, K, I, ldx, ldw, 0, act)) return LRP_ESHAPE;
    if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(Wgu) & 15)) return LRP_EALIGN;
    const int chunk = pp_row_chunk(ldx);
    for (int m0 = 0; m0 < M; m0 += chunk) {
        const int rc = lrp_launch_gemm_pp_gated_fwd((const char*)x + (int64_t)m0 * ldx * 2, Wgu, rs ? rs + m0 : nullptr,
                                                    (char*)coef + (int64_t)m0 * ldcoef * 2, (char*)m + (int64_t)m0 * ldm * 2,
                                                    M - m0 < chunk ? M - m0 : chunk, I, K, ldx, ldw, ldcoef, ldm, eps_g, eps_lin, act,
                                                    (hipStream_t)stream);
        if (rc != LRP_OK) return rc;
    }
    return LRP_OK;
}

extern "C" int lrp_gemm_gated_bwd_coef(const void* Adn, const void* Wdn, const void* coef, void* Agu, int M, int I, int K, int64_t lda,
                                       int64_t ldw, int64_t ldcoef, int64_t ldagu, int dtype, void* stream) {
    if (!Adn || !Wdn || !coef || !Agu || M < 0 || I < 0 || K < 0) return LRP_EINVAL;
    if (M == 0 || I == 0) return LRP_OK;
    if (dtype != LRP_BF16 || (lda % 8) || (ldw % 8) || !gated_fused_ok(M, I, K, I, lda, ldw, 1, LRP_ACT_SILU)) return LRP_ESHAPE;
    if ((reinterpret_cast<uintptr_t>(Adn) & 15) || (reinterpret_cast<uintptr_t>(Wdn) & 15)) return LRP_EALIGN;
    const int chunk = pp_row_chunk(lda);
    for (int m0 = 0; m0 < M; m0 += chunk) {
        const int rc = lrp_launch_gemm_pp_gated_bwd((const char*)Adn + (int64_t)m0 * lda * 2, Wdn, (const char*)coef + (int64_t)m0 * ldcoef * 2,
                                                    (char*)Agu + (int64_t)m0 * ldagu * 2, M - m0 < chunk ? M - m0 : chunk, I, K, lda, ldw,
                                                    ldcoef, ldagu, (hipStream_t)stream);
        if (rc != LRP_OK) return rc;
    }
    return LRP_OK;
}


// =================================================================================================
// K1n: Llama-type RMSNorm folded into the GEMMs around it (include/lrp_hip.h)
// =================================================================================================
namespace {
bool a16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
}  // namespace

extern "C" int lrp_gemm_norm_fused_ok(int M, int N, int K, int64_t lda, int64_t ldb, int nn, int dtype) {
    if (dtype != LRP_BF16 || M <= 0 || N <= 0 || K <= 0 || (N % 256) || (lda % 8) || (ldb % 8)) return 0;
    const int64_t tiles = (int64_t)((M + 255) / 256) * (N / 256);
    return (tiles >= 190 && pp_ok(M, N, K, lda, ldb, nn)) ? 1 : 0;
}

extern "C" int lrp_gemm_res_ssq(const void* x, const void* W, const void* res, void* out, float* ssq, int M, int N, int K, int64_t ldx,
                                int64_t ldw, int64_t ldres, int64_t ldout, int64_t ldssq, void* raw, int64_t ldraw, int dtype, void* stream) {
    if (!x || !W || !res || !out || !ssq || M < 0 || N < 0 || K < 0 || ldssq < M) return LRP_EINVAL;
    if (M == 0 || N == 0) return LRP_OK;
    if (!lrp_gemm_norm_fused_ok(M, N, K, ldx, ldw, 0, dtype)) return LRP_ESHAPE;
    if (!a16(x) || !a16(W) || (ldout % 8) || (ldres % 8) || (raw && (!a16(raw) || (ldraw % 8)))) return LRP_EALIGN;
    const int chunk = pp_row_chunk(ldx);
    for (int m0 = 0; m0 < M; m0 += chunk) {
        const int rc = lrp_launch_gemm_pp_res_ssq((const char*)x + (int64_t)m0 * ldx * 2, W, (const char*)res + (int64_t)m0 * ldres * 2,
                                                  (char*)out + (int64_t)m0 * ldout * 2, ssq + m0, M - m0 < chunk ? M - m0 : chunk, N, K, ldx, ldw,
                                                  ldres, ldout, ldssq, raw ? (char*)raw + (int64_t)m0 * ldraw * 2 : nullptr, ldraw, (hipStream_t)stream);
        if (rc != LRP_OK) return rc;
    }
    return LRP_OK;
}

extern "C" int lrp_gemm_nt_rs(const void* x, const void* W, const float* rs, void* out, int M, int N, int K, int64_t ldx, int64_t ldw,
                              int64_t ldout, int dtype, void* stream) {
    if (!x || !W || !rs || !out || M < 0 || N < 0 || K < 0) return LRP_EINVAL;
    if (M == 0 || N == 0) return LRP_OK;
    if (!lrp_gemm_norm_fused_ok(M, N, K, ldx, ldw, 0, dtype)) return LRP_ESHAPE;
    if (!a16(x) || !a16(W) || (ldout % 8)) return LRP_EALIGN;
    const int chunk = pp_row_chunk(ldx);
    for (int m0 = 0; m0 < M; m0 += chunk) {
        const int rc = lrp_launch_gemm_pp_nt_rs((const char*)x + (int64_t)m0 * ldx * 2, W, rs + m0, (char*)out + (int64_t)m0 * ldout * 2,
                                                M - m0 < chunk ? M - m0 : chunk, N, K, ldx, ldw, ldout, (hipStream_t)stream);
        if (rc != LRP_OK) return rc;
    }
    return LRP_OK;
}

extern "C" int lrp_gemm_nt_rs_rope_ok(int M, int N, int K, int64_t ldx, int64_t ldw, int64_t ldout, int seq, int rope_cols, int head_dim, int dtype) {
    if (head_dim != 128 || seq <= 0 || (seq % 16) || rope_cols < 0 || rope_cols > N || (rope_cols % 128) || (M % 256) || (ldout % 8)) return 0;
    const int chunk = pp_row_chunk(ldx);
    if (M > chunk && (chunk % seq) && (seq % chunk)) return 0;                // (row chunks of a huge batch must start on a prompt boundary modulo seq)
    return lrp_gemm_norm_fused_ok(M, N, K, ldx, ldw, 0, dtype);
}

extern "C" int lrp_gemm_nt_rs_rope(const void* x, const void* W, const float* rs, const float* cos, const float* sin, void* out, int M, int N, int K,
                                   int64_t ldx, int64_t ldw, int64_t ldout, int seq, int rope_cols, int head_dim, int dtype, void* stream) {
    if (!x || !W || !rs || !cos || !sin || !out || M < 0 || N < 0 || K < 0) return LRP_EINVAL;
    if (M == 0 || N == 0) return LRP_OK;
    if (!lrp_gemm_nt_rs_rope_ok(M, N, K, ldx, ldw, ldout, seq, rope_cols, head_dim, dtype)) return LRP_ESHAPE;
    if (!a16(x) || !a16(W) || !a16(out) || !a16(cos) || !a16(sin)) return LRP_EALIGN;
    const int chunk = pp_row_chunk(ldx);
    for (int m0 = 0; m0 < M; m0 += chunk) {
        // (the kernel takes positions as row % seq: a chunk that starts inside a prompt is shifted by handing it tables that start at that position)
        const int p0 = m0 % seq;
        const int rc = lrp_launch_gemm_pp_nt_rs_rope((const char*)x + (int64_t)m0 * ldx * 2, W, rs + m0, cos + (int64_t)p0 * 128, sin + (int64_t)p0 * 128,
                                                     (char*)out + (int64_t)m0 * ldout * 2, M - m0 < chunk ? M - m0 : chunk, N, K, ldx, ldw, ldout,
                                                     seq, rope_cols, (hipStream_t)stream);
        if (rc != LRP_OK) return rc;
    }
    return LRP_OK;
}

extern "C" int lrp_gemm_nn_rs(const void* s, const void* W, const float* rs, void* out, int M, int N, int K, int64_t lds_, int64_t ldw,
                              int64_t ldout, int dtype, void* stream) {
    if (!s || !W || !rs || !out || M < 0 || N < 0 || K < 0) return LRP_EINVAL;
    if (M == 0 || N == 0) return LRP_OK;
    if (!lrp_gemm_norm_fused_ok(M, N, K, lds_, ldw, 1, dtype)) return LRP_ESHAPE;
    if (!a16(s) || !a16(W) || (ldout % 8)) return LRP_EALIGN;
    const int chunk = pp_row_chunk(lds_);
    for (int m0 = 0; m0 < M; m0 += chunk) {
        const int rc = lrp_launch_gemm_pp_nn_rs((const char*)s + (int64_t)m0 * lds_ * 2, W, rs + m0, (char*)out + (int64_t)m0 * ldout * 2,
                                                M - m0 < chunk ? M - m0 : chunk, N, K, lds_, ldw, ldout, (hipStream_t)stream);
        if (rc != LRP_OK) return rc;
    }
    return LRP_OK;
}

extern "C" int lrp_gemm_nn_rs_res(const void* s, const void* W, const float* rs, const void* res, void* out, int M, int N, int K, int64_t lds_,
                                  int64_t ldw, int64_t ldres, int64_t ldout, int dtype, void* stream) {
    if (!s || !W || !rs || !res || !out || M < 0 || N < 0 || K < 0) return LRP_EINVAL;
    if (M == 0 || N == 0) return LRP_OK;
    if (!lrp_gemm_norm_fused_ok(M, N, K, lds_, ldw, 1, dtype)) return LRP_ESHAPE;
    if (!a16(s) || !a16(W) || (ldout % 8) || (ldres % 8)) return LRP_EALIGN;
    const int chunk = pp_row_chunk(lds_);
    for (int m0 = 0; m0 < M; m0 += chunk) {
        const int rc = lrp_launch_gemm_pp_nn_rs_res((const char*)s + (int64_t)m0 * lds_ * 2, W, rs + m0, (const char*)res + (int64_t)m0 * ldres * 2,
                                                    (char*)out + (int64_t)m0 * ldout * 2, M - m0 < chunk ? M - m0 : chunk, N, K, lds_, ldw, ldres,
                                                    ldout, (hipStream_t)stream);
        if (rc != LRP_OK) return rc;
    }
    return LRP_OK;
}
