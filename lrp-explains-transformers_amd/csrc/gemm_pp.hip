// gemm_pp.hip -- the big bf16 NT GEMMs of the Linear eps-rule (K1 at M = B*S rows): C[M,N] = A[M,K] . B[N,K]^T (+bias),
// fp32 accumulate, as an 8-wave PING-PONG kernel on v_mfma_f32_32x32x16_bf16.
//
// Why this structure (profiles/r01_gemm_experiments.txt, r02_gemm_experiments.txt): every lock-step form (4, 8 or 16 waves that
// all read LDS, then all issue MFMAs) lands at 1.2-1.3 PFLOP/s because its MFMA-slot occupancy stops at ~63 %; the package runs at
// its power cap, so only occupancy -- not latency hiding -- moves the number.  Here the 8 waves of a workgroup form two groups
// of four (one wave of each group per SIMD) that run HALF A PHASE APART: while group X issues 16 back-to-back MFMAs (32x32x16:
// a single wave issues them at the pipe rate, 32 cycles each; 16x16x32 would be issue-limited at 18.25/16), group Y reads its
// next fragments out of LDS and issues the direct-to-LDS loads of a later K tile; then they swap.  A SIMD's matrix pipe always
// has one wave feeding it; LDS reads, address arithmetic, waits and LDS-DMA issue all sit in the other wave's shadow.
//
// Geometry: 256 x 256 tile, K tile = 64 elements (128 B per row), 512 threads.  Wave w: group g = w >> 2 (rows g*128 .. +127
// of the tile), column block wc = w & 3 (columns wc*64 .. +63): 128 x 64 per wave = 2 row halves (a) x {2 x 2 MFMA tiles of
// 32 x 32} = 128 accumulator registers.  One phase = one row half a of one K tile: L = 8 A-fragment reads (+ 8 B-fragment
// reads when a = 0; the B fragments are kept for a = 1), M = 16 MFMAs (4 k-steps x 2 x 2 tiles).  Registers: 128 acc +
// 32 (A) + 32 (B) fragments + 8 LDS addresses + 16 (source pointers).
//
// LDS (128 KiB): [A buf0][A buf1][B buf0][B buf1], 32 KiB each = 256 rows x 128 B.  Row r holds its eight 16-byte chunks at
// slot = chunk ^ ((r >> 1) & 7): a 32-row fragment read (lane l: row l & 31, chunk 2 ks + (l >> 5)) is conflict-free under the
// ds_read_b128 lane groups of the hardware (tests/test_layout_maps_cpu.py).  Direct-to-LDS loads write lane-linear 1-KiB pieces
// (8 rows), so the swizzle is applied to the per-lane SOURCE chunk.  Every fragment address is a per-lane register + immediate.
//
// Staging units per K tile t (unit order = issue order = consumption order):
//     V0(t) = A rows of row half 0 of both groups (128 rows, 2 pieces per wave)
//     V1(t) = all 256 B rows                           (4 pieces per wave)
//     V2(t) = A rows of row half 1                      (2 pieces per wave)
//   phase (t,0) reads V0(t), V1(t) and issues V2(t+1) into the other buffer (free since phase (t-1,1) of both groups);
//   phase (t,1) reads V2(t)        and issues V0(t+2), V1(t+2) into THIS buffer (free: both groups finished phase (t,0)'s
//   reads, which are waited for -- lgkmcnt(0) -- BEFORE the barrier that ends the L interval).
//   Loads are therefore 5-6 intervals (>= 2.5 K-tile times) ahead of their first read.  After its issue a wave waits
//   vmcnt(8): everything but the newest 8 pieces has landed = the unit the NEXT phase reads; the barrier publishes it.
// Barriers: L | barrier | M | barrier per phase; group 1 executes ONE extra barrier up front, which puts it half a phase behind
// for the whole kernel (and group 0 one at the very end to balance the count).
#include "common.hpp"

namespace {

constexpr int PP_KT = 64;                         // K elements per K tile
constexpr int PP_OPND = 256 * 128;                // one operand of one buffer (bytes)

typedef __attribute__((address_space(3))) void* pp_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* pp_glb_ptr_t;

#ifndef PP_SETPRIO
#define PP_SETPRIO 1
#endif

// PP_TIMELINE (dev builds only, tools/gemm_ab.py): waves 0 and 4 of a few workgroups stamp s_memtime at the start of every
// interval into the spare LDS above the 128 KiB of tiles and dump it through the `bias` pointer (which is then NOT a bias).
#ifdef PP_TIMELINE
#define PP_STAMP()                                                                                \
    if (tl_on) {                                                                                  \
        const uint64_t tt_ = __builtin_amdgcn_s_memtime();                                        \
        if (tl_idx < 192) asm volatile("ds_write_b64 %0, %1" ::"v"(tl_base + 8u * tl_idx), "v"(tt_) : "memory");   \
        ++tl_idx;                                                                                 \
    }
#else
#define PP_STAMP()
#endif

#define PP_DSRD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))

template <typename TO>
__global__ __launch_bounds__(512, 2) void gemm_nt_pp_kernel(
    const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, TO* __restrict__ C, const bf16_t* __restrict__ bias,
    int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave >> 2, wc = wave & 3;
    const int nkt = K / PP_KT;                                         // host guarantees nkt >= 2
    int tm, tn;
    grouped_tile(xcd_remap(blockIdx.x, tiles_m * tiles_n), tiles_m, tiles_n, tm, tn);
    const int m0 = tm * 256, n0 = tn * 256;

    // ---- staging sources: piece = 8 rows x 128 B; lane l -> row (l >> 3), LDS slot (l & 7), source chunk slot ^ ((row >> 1) & 7)
    // A pieces of this wave: rows g*128 + a*64 + (wc*16 + 8 p) .. + 7 (a = unit half, p = 0, 1); B pieces: rows wave*32 + 8 p (p = 0..3)
    const int prow = lane >> 3, pslot = lane & 7;
    const bf16_t* srcA[2][2];
    const bf16_t* srcB[4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int r = g * 128 + a * 64 + wc * 16 + 8 * p + prow;
            int gr = m0 + r;
            gr = gr < M ? gr : M - 1;
            srcA[a][p] = A + (int64_t)gr * lda + ((pslot ^ ((r >> 1) & 7)) << 3);
        }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int r = wave * 32 + 8 * p + prow;
        int gr = n0 + r;
        gr = gr < N ? gr : N - 1;
        srcB[p] = B + (int64_t)gr * ldb + ((pslot ^ ((r >> 1) & 7)) << 3);
    }
    char* const ldsA = smem + (g * 128 + wc * 16) * 128;               // + buf*PP_OPND + a*8192 + p*1024
    char* const ldsB = smem + 2 * PP_OPND + (wave * 32) * 128;         // + buf*PP_OPND + p*1024
    auto stage_A = [&](int a, int kt, int buf) {
#pragma unroll
        for (int p = 0; p < 2; ++p)
            __builtin_amdgcn_global_load_lds((pp_glb_ptr_t)(srcA[a][p] + (int64_t)kt * PP_KT),
                                             (pp_lds_ptr_t)(ldsA + buf * PP_OPND + a * 8192 + p * 1024), 16, 0, 0);
    };
    auto stage_B = [&](int kt, int buf) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
            __builtin_amdgcn_global_load_lds((pp_glb_ptr_t)(srcB[p] + (int64_t)kt * PP_KT),
                                             (pp_lds_ptr_t)(ldsB + buf * PP_OPND + p * 1024), 16, 0, 0);
    };

    // ---- fragment addresses: row (l & 31) of a 32-row block, chunk (2 ks + (l >> 5)) ^ ((l >> 1) & 7)
    const int hi = lane >> 5, sw = (lane >> 1) & 7;
    const unsigned rowA = (unsigned)((g * 128 + (lane & 31)) * 128), rowB = (unsigned)(2 * PP_OPND + (wc * 64 + (lane & 31)) * 128);
    unsigned cA[4], cB[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const unsigned c = (unsigned)(((2 * ks + hi) ^ sw) << 4);
        cA[ks] = rowA + c;
        cB[ks] = rowB + c;
    }

    f32x16 acc[2][2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][i][j][r] = 0.f;
    bf16x8 fa[2][4], fb[2][4];                                          // [32-row block][k-step]

#ifdef PP_TIMELINE
    const int tl_slot = (blockIdx.x == 0) ? 0 : (blockIdx.x == 101) ? 1 : (blockIdx.x == 257) ? 2 : -1;
    const bool tl_on = tl_slot >= 0 && (wave & 3) == 0;
    const unsigned tl_base = 4u * PP_OPND + (unsigned)g * 2048u;
    unsigned tl_idx = 0;
    PP_STAMP()
#endif
    // ---- prologue: V0(0) V1(0) V2(0) V0(1) V1(1); V0(0), V1(0) landed = all but the newest 8 pieces
    stage_A(0, 0, 0); stage_B(0, 0); stage_A(1, 0, 0); stage_A(0, 1, 1); stage_B(1, 1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (g == 1) __builtin_amdgcn_s_barrier();                          // the half-phase offset between the two groups

#define PP_READ_A(BUF, AH)                                                                        \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                            \
        PP_DSRD(fa[0][ks], cA[ks], (BUF) * PP_OPND + (AH) * 8192);                                \
        PP_DSRD(fa[1][ks], cA[ks], (BUF) * PP_OPND + (AH) * 8192 + 4096);                         \
    }
#define PP_READ_B(BUF)                                                                            \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                            \
        PP_DSRD(fb[0][ks], cB[ks], (BUF) * PP_OPND);                                              \
        PP_DSRD(fb[1][ks], cB[ks], (BUF) * PP_OPND + 4096);                                       \
    }
#define PP_WAIT_A() asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[0][2]), "+v"(fa[0][3]),   \
                                 "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fa[1][2]), "+v"(fa[1][3]) :: "memory")
#define PP_WAIT_B() asm volatile("" : "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fb[0][2]), "+v"(fb[0][3]),                        \
                                 "+v"(fb[1][0]), "+v"(fb[1][1]), "+v"(fb[1][2]), "+v"(fb[1][3]))
#define PP_MMA(AH)                                                                                \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                              \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                             \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                         \
                acc[AH][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j][ks], fa[i][ks], acc[AH][i][j], 0, 0, 0);
#define PP_FENCE() __builtin_amdgcn_sched_barrier(0)
#if PP_SETPRIO
#define PP_PRIO(n) __builtin_amdgcn_s_setprio(n)
#else
#define PP_PRIO(n)
#endif

    // one K tile out of buffer BUF (compile-time); t is the running K-tile index
#define PP_KTILE(BUF)                                                                             \
    {                                                                                             \
        /* ---- phase (t, 0): L */                                                                \
        PP_READ_B(BUF) PP_READ_A(BUF, 0)                                                          \
        if (t + 1 < nkt) { stage_A(1, t + 1, (BUF) ^ 1); asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }   \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                     \
        PP_WAIT_A(); PP_WAIT_B(); PP_FENCE();                                                     \
        __builtin_amdgcn_s_barrier(); PP_FENCE(); PP_STAMP()                                      \
        /* ---- M */                                                                              \
        PP_PRIO(1); PP_MMA(0) PP_PRIO(0); PP_FENCE();                                             \
        __builtin_amdgcn_s_barrier(); PP_FENCE(); PP_STAMP()                                      \
        /* ---- phase (t, 1): L */                                                                \
        PP_READ_A(BUF, 1)                                                                         \
        if (t + 2 < nkt) { stage_A(0, t + 2, BUF); stage_B(t + 2, BUF); asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }   \
        else if (t + 1 < nkt) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                    \
        PP_WAIT_A(); PP_FENCE();                                                                  \
        __builtin_amdgcn_s_barrier(); PP_FENCE(); PP_STAMP()                                      \
        /* ---- M */                                                                              \
        PP_PRIO(1); PP_MMA(1) PP_PRIO(0); PP_FENCE();                                             \
        __builtin_amdgcn_s_barrier(); PP_FENCE(); PP_STAMP()                                      \
    }

    int t = 0;
    for (; t + 1 < nkt; t += 2) {
        PP_KTILE(0)
        ++t;
        PP_KTILE(1)
        --t;
    }
    if (t < nkt) PP_KTILE(0)

#ifdef PP_TIMELINE
    uint64_t* const tl_out = (uint64_t*)bias;
    bias = nullptr;
#endif
    // ---- epilogue.  D = mfma(Bfrag, Afrag): lane l holds C[m = .. + (l & 31)][n = .. + 8 q + 4 (l >> 5) + e], acc register 4 q + e
    const int mrow = m0 + g * 128 + (lane & 31);
    const int ncol = n0 + wc * 64;
    const bool full = (m0 + 256 <= M) && (n0 + 256 <= N) && ((ldc & 7) == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int gm = mrow + a * 64 + i * 32;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x16 v = acc[a][i][j];
                const int nb = ncol + j * 32;
                if (bias) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int gn = nb + 8 * q + 4 * hi + e;
                            if (gn < N) v[4 * q + e] += to_f32(bias[gn]);
                        }
                }
                if constexpr (sizeof(TO) == 4) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int gn = nb + 8 * q + 4 * hi;
                        TO* dst = C + (int64_t)gm * ldc + gn;
                        if (full) {
                            *reinterpret_cast<f32x4*>(dst) = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
                        } else if (gm < M) {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (gn + e < N) dst[e] = v[4 * q + e];
                        }
                    }
                } else {
                    // pack to bf16, then pair the two lane halves (v_permlane32_swap) so that a lane stores 16 contiguous bytes:
                    // lower half n = 16 qq .. + 7, upper half n = 16 qq + 8 .. + 15
                    uint32_t w[4][2];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        bf16x2 p0 = {(bf16_t)v[4 * q], (bf16_t)v[4 * q + 1]}, p1 = {(bf16_t)v[4 * q + 2], (bf16_t)v[4 * q + 3]};
                        w[q][0] = __builtin_bit_cast(uint32_t, p0);
                        w[q][1] = __builtin_bit_cast(uint32_t, p1);
                    }
                    if (full) {
#pragma unroll
                        for (int qq = 0; qq < 2; ++qq) {
                            uint32_t x0 = w[2 * qq][0], x1 = w[2 * qq][1], y0 = w[2 * qq + 1][0], y1 = w[2 * qq + 1][1];
                            auto s0 = __builtin_amdgcn_permlane32_swap(x0, y0, false, false);
                            auto s1 = __builtin_amdgcn_permlane32_swap(x1, y1, false, false);
                            u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
                            *reinterpret_cast<u32x4*>(C + (int64_t)gm * ldc + nb + 16 * qq + 8 * hi) = o;
                        }
                    } else if (gm < M) {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int gn = nb + 8 * q + 4 * hi + e;
                                if (gn < N) C[(int64_t)gm * ldc + gn] = (bf16_t)v[4 * q + e];
                            }
                    }
                }
            }
        }
    if (g == 0) __builtin_amdgcn_s_barrier();                          // balances group 1's extra barrier
#ifdef PP_TIMELINE
    if (tl_on) {
        PP_STAMP()                                                      // end of the C stores' issue
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PP_STAMP()                                                      // stores retired
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (tl_out && lane < 3) {
            // lane 0..2 copy 64 stamps each
            for (int i2 = lane * 64; i2 < lane * 64 + 64; ++i2) {
                uint64_t v2;
                asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v2) : "v"(tl_base + 8u * (unsigned)i2) : "memory");
                tl_out[(tl_slot * 2 + g) * 192 + i2] = v2;
            }
        }
    }
#endif
#undef PP_KTILE
#undef PP_READ_A
#undef PP_READ_B
#undef PP_WAIT_A
#undef PP_WAIT_B
#undef PP_MMA
#undef PP_FENCE
#undef PP_PRIO
}

}  // namespace

// host entry used by lrp_gemm_nt's dispatcher (gemm.hip): bf16 operands, K a multiple of 64 with K >= 128, operands < 4 GiB apart
template <typename TO>
static int launch_pp_t(const void* A, const void* B, void* C, const void* bias, int M, int N, int K, int64_t lda, int64_t ldb,
                       int64_t ldc, hipStream_t st) {
    const int tiles_m = (M + 255) / 256, tiles_n = (N + 255) / 256;
    dim3 grid(tiles_m * tiles_n), block(512);
#ifdef PP_TIMELINE
    const size_t lds = 4 * (size_t)PP_OPND + 4096;
#else
    const size_t lds = 4 * (size_t)PP_OPND;
#endif
    auto kern = gemm_nt_pp_kernel<TO>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, grid, block, lds, st, (const bf16_t*)A, (const bf16_t*)B, (TO*)C, (const bf16_t*)bias, M, N, K, lda,
                       ldb, ldc, tiles_m, tiles_n);
    return lrp_check_launch();
}

int lrp_launch_gemm_pp(const void* A, const void* B, void* C, const void* bias, int M, int N, int K, int64_t lda, int64_t ldb,
                       int64_t ldc, int out_dtype, hipStream_t st) {
    if (out_dtype == LRP_F32) return launch_pp_t<float>(A, B, C, bias, M, N, K, lda, ldb, ldc, st);
    return launch_pp_t<bf16_t>(A, B, C, bias, M, N, K, lda, ldb, ldc, st);
}
