// gemm32.hip -- bf16 NT GEMM, 256x256 tile, on v_mfma_f32_32x32x16_bf16 (half the MFMA instructions and half the
// VGPR operand reads per FLOP of the 16x16x32 form; DESIGN.md section 4.1: under the 1400 W cap the GEMM is bound by energy
// per FLOP, not by a pipeline bubble).  16 waves (4 x 4), 64 x 64 per wave = 2 x 2 MFMA blocks, K step 64 (128 B per row),
// direct-to-LDS staging, two stages.  LDS image: row-major [rows][128 B], 16-byte chunk c of row r at slot c ^ ((r >> 1) & 7):
// the 32-row fragment read (lane = row l & 31, 16-byte chunk 2 k16 + (l >> 5)) is conflict-free for every ds_read_b128
// lane group (8 even + 8 odd rows per group, the two parities sit in the two 128-byte halves of a 256-byte bank row).
// Operands are swapped in the MFMA (a = B rows, b = A rows) so that a lane owns 4 consecutive output columns.
#include "../common.hpp"

namespace gemm32 {

constexpr int KB = 128;                 // bytes of K per row and stage
constexpr int TBM = 256, TBN = 256, WN = 4, NW = 16;
constexpr int STAGE = (TBM + TBN) * KB;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

LRP_DEVICE f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

template <typename TO>
__global__ __launch_bounds__(1024, 4) void gemm_nt_m32_kernel(
    const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, TO* __restrict__ C, const bf16_t* __restrict__ bias,
    int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, int tiles_m, int tiles_n) {
    constexpr int KE = 64, GA = TBM / 8 / NW, GB = TBN / 8 / NW;        // 2 one-KiB row groups per wave and operand
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int nkt = K / KE;
    int tm, tn;
    grouped_tile(xcd_remap(blockIdx.x, tiles_m * tiles_n), tiles_m, tiles_n, tm, tn);
    const int m0 = tm * TBM, n0 = tn * TBN;

    // staging: lane l of a wave instruction -> LDS row (l >> 3) of its 8-row group, slot (l & 7); row r holds global
    // chunk c at slot c ^ ((r >> 1) & 7)  =>  the lane fetches chunk (l & 7) ^ ((row >> 1) & 7)
    const int lrow = lane >> 3;
    const bf16_t* pa[GA];
    const bf16_t* pb[GB];
#pragma unroll
    for (int i = 0; i < GA; ++i) {
        const int row = (wave * GA + i) * 8 + lrow;
        int r = m0 + row;
        r = r < M ? r : M - 1;
        pa[i] = A + (int64_t)r * lda + (((lane & 7) ^ ((row >> 1) & 7)) * 8);
    }
#pragma unroll
    for (int i = 0; i < GB; ++i) {
        const int row = (wave * GB + i) * 8 + lrow;
        int r = n0 + row;
        r = r < N ? r : N - 1;
        pb[i] = B + (int64_t)r * ldb + (((lane & 7) ^ ((row >> 1) & 7)) * 8);
    }
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
    // fragment offsets: row (blk * 32 + l31) of the wave's rows, chunk (2 k16 + hi) ^ ((l31 >> 1) & 7)
    uint32_t fo[4];
#pragma unroll
    for (int k16 = 0; k16 < 4; ++k16) fo[k16] = l31 * KB + (((2 * k16 + hi) ^ ((l31 >> 1) & 7)) << 4);
    const int offA = (wm * 64) * KB, offB = TBM * KB + (wn * 64) * KB;

    f32x16 acc[2][2];                                   // [m block][n block]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    stage(0, 0);
    __syncthreads();
    int cur = 0;
    // fragments double-buffered in registers: the reads of K sub-step k16+1 are issued before the MFMAs of sub-step k16
    bf16x8 fa[2][2], fb[2][2];
    auto rd = [&](int buf, int k16, int slot) {
        const char* sa = smem + buf * STAGE + offA;
        const char* sb = smem + buf * STAGE + offB;
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[slot][i] = *reinterpret_cast<const bf16x8*>(sa + i * 32 * KB + fo[k16]);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[slot][j] = *reinterpret_cast<const bf16x8*>(sb + j * 32 * KB + fo[k16]);
    };
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) stage(kt + 1, cur ^ 1);
        rd(cur, 0, 0);
#pragma unroll
        for (int k16 = 0; k16 < 4; ++k16) {
            if (k16 < 3) rd(cur, k16 + 1, (k16 + 1) & 1);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(fb[k16 & 1][j], fa[k16 & 1][i], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        cur ^= 1;
    }

    // D[i = n][j = m]: lane (m = l31, hi) holds n = 8 g + 4 hi + e (g = r >> 2, e = r & 3): 4 consecutive output columns
    const bool vec_ok = ((ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int gm = m0 + wm * 64 + i * 32 + l31;
        if (gm >= M) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int gn = n0 + wn * 64 + j * 32 + 8 * g + 4 * hi;
                if (gn >= N) continue;
                f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                if (bias) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (gn + e < N) v[e] += to_f32(bias[gn + e]);
                }
                TO* dst = C + (int64_t)gm * ldc + gn;
                if (vec_ok && gn + 3 < N) {
                    if constexpr (sizeof(TO) == 4) *reinterpret_cast<f32x4*>(dst) = v;
                    else {
                        bf16x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = (bf16_t)v[e];
                        *reinterpret_cast<bf16x4*>(dst) = o;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (gn + e < N) dst[e] = from_f32<TO>(v[e]);
                }
            }
    }
}

}  // namespace gemm32

template <typename TO>
static int launch_m32_t(const void* A, const void* B, void* C, const void* bias, int M, int N, int K, int64_t lda, int64_t ldb,
                        int64_t ldc, hipStream_t st) {
    using namespace gemm32;
    const int tiles_m = (M + TBM - 1) / TBM, tiles_n = (N + TBN - 1) / TBN;
    const size_t lds = 2 * (size_t)STAGE;
    auto kern = gemm_nt_m32_kernel<TO>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(1024), lds, st, (const bf16_t*)A, (const bf16_t*)B, (TO*)C,
                       (const bf16_t*)bias, M, N, K, lda, ldb, ldc, tiles_m, tiles_n);
    return lrp_check_launch();
}

// bf16 operands, K % 64 == 0, batch 1
int lrp_gemm_m32(const void* A, const void* B, void* C, const void* bias, int M, int N, int K, int64_t lda, int64_t ldb,
                 int64_t ldc, int out_f32, hipStream_t st) {
    if (out_f32) return launch_m32_t<float>(A, B, C, bias, M, N, K, lda, ldb, ldc, st);
    return launch_m32_t<bf16_t>(A, B, C, bias, M, N, K, lda, ldb, ldc, st);
}
