// gemm.hip -- NT GEMM on MFMA for the Linear eps-rule (K1) and every other contraction that
// is not attention.   C[M,N] = A[M,K] . B[N,K]^T (+bias), fp32 accumulate.
//
// Design (gfx950, wave64):
//   * tile BM x BN = 128 x 128, K step = 128 BYTES per row (64 bf16 / 32 fp32), so the LDS
//     image, the staging code and the bank-conflict analysis are dtype-independent;
//   * 256 threads = 4 waves in a 2x2 grid, each wave a 64x64 sub-tile = 4x4 MFMA 16x16 tiles,
//     operands swapped (mma(Bfrag, Afrag)) so a lane owns 4 CONSECUTIVE output columns of one
//     row -> one 8/16-byte store per tile instead of four scalars;
//   * LDS rows are 128 B; the 16-B chunk index is XOR-swizzled with (row & 7): ds_write_b128 of
//     8 consecutive lanes covers one full row, and every ds_read_b128 lane group touches 16
//     distinct 16-B bank slots (conflict-free, checked per lane group of MI355X_MICROARCH LDS table);
//   * global -> register -> LDS staging, double-buffered: the loads of tile t+1 are issued before
//     the MFMAs of tile t and written to the other buffer after them; one barrier per K step;
//   * 64 KiB LDS/block -> 2 blocks per CU; XCD-aware block remap keeps a B panel in one L2.
#include "common.hpp"

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

template <typename T, typename TO>
int launch_gemm(const void* A, const void* B, void* C, const void* bias, int M, int N, int K,
                int64_t lda, int64_t ldb, int64_t ldc, int batch, int64_t sA, int64_t sB,
                int64_t sC, hipStream_t st) {
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    dim3 grid(tiles_m * tiles_n, batch), block(NT);
    const size_t lds = 2 * (BM + BN) * KB;
    static bool attr_set = false;   // per instantiation; idempotent, so a race is harmless
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_kernel<T, TO>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
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
