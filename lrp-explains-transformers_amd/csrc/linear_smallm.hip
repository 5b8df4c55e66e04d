// linear_smallm.hip -- K1 in its HBM-bound regime for M <= 16 rows: W-streaming forward and eps-rule dgrad kernels that read W [N,K]
// exactly once from its stored layout (fp32 and bf16, any N, K a multiple of 16 bytes).  ref: lxt/explicit/functional.py:345-364
// (z = xW^T+b ; s = R/(z+eps) ; R_in = x*(sW)).  bf16 problems with K % 64 == 0 are also served -- up to M = 256 rows -- by the split-K
// skinny path of the ping-pong GEMM (gemm.hip: lrp_gemm_skinny); ops.linear_fwd / ops.linear_dgrad choose.
// (The round-1 one-pass kernel -- z recompute + redistribution in one sweep over W, M <= 4 -- was removed in round 3: with forward and
// backward separated in time by the protocol, W is read once per direction either way, and the pair below is faster per direction.)
#include "common.hpp"

namespace {

// =====================================================================================================================
// Small-M forward and dgrad as pure W-streaming kernels (M <= 16).  One WAVE per workgroup owns a [rows x 512 columns]
// panel of W (1 KiB per row and load instruction, fully coalesced), 16 rows in flight per wave, nothing shared: no LDS, no
// barrier.  The 2-D decomposition (k-slabs x row ranges) keeps the cross-workgroup reduction small in BOTH directions:
//   dgrad   c[m,k] = sum_n s[m,n] W[n,k] : lane-local accumulators for the lane's 8 columns; row ranges are summed from
//           fp32 slabs [n_ranges][M][K] by a second tiny kernel (s = g z/(z+eps) or r/(z+eps) is formed on the fly from the
//           stashed z -- the eps-rule's "redistribution" half with W read exactly once and NO W^T copy);
//   forward z[m,n]  = sum_k x[m,k] W[n,k] : per-row dot over the lane's columns + wave reduction; k-slab partials
//           [k_slabs][M][N] fp32 are summed (+ bias, cast) by the same kind of tiny kernel.
// Algorithmic HBM bytes: sizeof(T) * N * K (+ the M-row operands).  ref: lxt/explicit/functional.py:345-364.
// =====================================================================================================================
constexpr int SM_R = 16;            // rows of W in flight per wave

template <typename T, int MM, bool RATIO>
__global__ __launch_bounds__(512) void smallm_dgrad_kernel(
    const T* __restrict__ W, const T* __restrict__ g, const T* __restrict__ z, float* __restrict__ slab, int M, int N, int K,
    int64_t ldg, int64_t ldz, int rows_per, float eps, int rel_in) {
    constexpr int EPC = 16 / (int)sizeof(T), COLS = 64 * EPC;
    __shared__ float red[8][4][64 * 8];                                  // one round: 8 waves x 4 rows m x (up to) 512 columns
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int k = (blockIdx.x * 64 + lane) * EPC;
    const bool kok = k < K;
    // the workgroup's row range is split over its 8 waves in whole batches of SM_R rows
    const int g0 = blockIdx.y * rows_per, g1 = min(N, g0 + rows_per);
    const int per_wave = ((rows_per + 7) / 8 + SM_R - 1) / SM_R * SM_R;
    const int n0 = min(g1, g0 + wave * per_wave), n1 = min(g1, n0 + per_wave);
    float acc[MM][EPC];
#pragma unroll
    for (int m = 0; m < MM; ++m)
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[m][e] = 0.f;
    for (int nb = n0; nb < n1; nb += SM_R) {
        Vec16<T> w[SM_R];
#pragma unroll
        for (int r = 0; r < SM_R; ++r) {
            const int n = nb + r;
            if (kok && n < n1) w[r] = ld16(W + (int64_t)n * K + k);
            else {
#pragma unroll
                for (int e = 0; e < EPC; ++e) w[r].set(e, 0.f);
            }
        }
        // s[m][n] of the batch: value v = r * MM + m is computed by lane (v & 63) of vector (v >> 6) from ONE load of g (and z)
        // each, then handed to every lane through v_readlane (an SGPR operand of the FMAs below)
        constexpr int NV = (SM_R * MM + 63) / 64;
        float svec[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int v = j * 64 + lane, r = v / MM, m = v % MM, n = nb + r;
            float sv = 0.f;
            if (v < SM_R * MM && m < M && n < n1) {
                const float gv = to_f32(g[(int64_t)m * ldg + n]);
                if constexpr (RATIO) {
                    const float zv = to_f32(z[(int64_t)m * ldz + n]);
                    sv = rel_in ? gv / (zv + eps) : gv * eps_ratio(zv, 1.f, eps);
                } else sv = gv;
            }
            svec[j] = sv;
        }
#pragma unroll
        for (int r = 0; r < SM_R; ++r)
#pragma unroll
            for (int m = 0; m < MM; ++m) {
                const int v = r * MM + m;
                const float sv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, svec[v >> 6]), v & 63));
#pragma unroll
                for (int e = 0; e < EPC; ++e) acc[m][e] += sv * w[r].get(e);
            }
    }
    // fold the 8 waves' partial columns through LDS in a FIXED order (deterministic), 4 rows m per round; one fp32 slab row
    // per workgroup and m
    constexpr int ROUNDS = (MM + 3) / 4;
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {
        if (rd > 0) __syncthreads();
#pragma unroll
        for (int mm = 0; mm < 4; ++mm) {
            const int m = rd * 4 + mm;
            if (m < MM) {
#pragma unroll
                for (int e4 = 0; e4 < EPC / 4; ++e4)
                    *reinterpret_cast<f32x4*>(&red[wave][mm][lane * EPC + 4 * e4]) =
                        f32x4{acc[m][4 * e4], acc[m][4 * e4 + 1], acc[m][4 * e4 + 2], acc[m][4 * e4 + 3]};
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 4 * COLS / 4; i += 512) {
            const int mm = (i * 4) / COLS, c = (i * 4) % COLS, m = rd * 4 + mm;
            const int kk = blockIdx.x * COLS + c;
            if (m >= M || kk >= K) continue;
            f32x4 sum = *reinterpret_cast<const f32x4*>(&red[0][mm][c]);
#pragma unroll
            for (int w = 1; w < 8; ++w) sum += *reinterpret_cast<const f32x4*>(&red[w][mm][c]);
            *reinterpret_cast<f32x4*>(slab + ((int64_t)blockIdx.y * M + m) * K + kk) = sum;
        }
    }
}

// out[i] = (sum_b slab[b][i]) (* x[i]) for i < MK, out dtype TO
template <typename T, typename TO>
__global__ void smallm_slab_sum_kernel(const float* __restrict__ slab, const T* __restrict__ x, TO* __restrict__ out, int nslab,
                                       int64_t MK, int rel_out) {
    __shared__ f32x4 part[16][16];
    const int ql = threadIdx.x & 15, sg = threadIdx.x >> 4;
    const int64_t i = ((int64_t)blockIdx.x * 16 + ql) * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (i < MK) {
#pragma unroll 4
        for (int b = sg; b < nslab; b += 16) s += *reinterpret_cast<const f32x4*>(slab + (int64_t)b * MK + i);
    }
    part[sg][ql] = s;
    __syncthreads();
    if (sg == 0 && i < MK) {
#pragma unroll
        for (int kk = 1; kk < 16; ++kk) s += part[kk][ql];
#pragma unroll
        for (int e = 0; e < 4; ++e) out[i + e] = from_f32<TO>(rel_out ? s[e] * to_f32(x[i + e]) : s[e]);
    }
}

// forward on MFMA.  A workgroup of 8 waves owns a range of <= 128 rows of W; the waves split the k-slabs (1 KiB per row each:
// 8 "k-pairs" of 128 B = one cache line per row) among themselves and walk 16-row blocks.  A lane (row r = l & 15, group
// g = l >> 4) loads the 32 contiguous bytes at byte g*32 of each line as two 16-byte pieces: piece 0 feeds MFMA macro step
// 2p, piece 1 step 2p+1 (the contraction-slot order is free as long as the x fragments use the same one), so every load pair
// covers 16 full lines; two blocks (32 KiB) are in flight per wave.  The x fragments of the slab (<= 16 rows, zero padded) stay
// in registers.  Per-wave partial sums z[row][m] live in that wave's own LDS region (no barrier while streaming); one
// barrier at the end, the 8 regions are summed in a FIXED order (deterministic), bias added, z stored: no workspace, no
// second launch.
constexpr int FW_ROWS = 128;                 // max rows per workgroup

template <typename T, typename TO>
__global__ __launch_bounds__(512, 2) void smallm_fwd_kernel(const T* __restrict__ W, const T* __restrict__ x, const T* __restrict__ bias,
                                                            TO* __restrict__ z, int M, int N, int K, int64_t ldx, int64_t ldz,
                                                            int rows_per, int nslab) {
    constexpr int EPC = 16 / (int)sizeof(T), SLAB = 1024 / (int)sizeof(T), NP = 8;       // elements per slab, k-pairs per slab
    typedef typename Mma16<T>::frag frag_t;
    __shared__ float red[8][FW_ROWS][16];                                 // 64 KiB
    const int lane = threadIdx.x & 63, r16 = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n0 = blockIdx.x * rows_per, n1 = min(N, n0 + rows_per);
    auto zero_frag = [] { u32x4 zz = {0, 0, 0, 0}; return __builtin_bit_cast(frag_t, zz); };
    bool first = true;
    for (int sl = wave; sl < nslab; sl += 8) {
        const int k0 = sl * SLAB;
        frag_t xf[NP][2];                                                  // x rows m = r16, same (pair, piece, g) -> k map as W
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k = k0 + p * (8 * EPC) + g * (2 * EPC) + h * EPC;
                xf[p][h] = (r16 < M && k < K) ? *reinterpret_cast<const frag_t*>(x + (int64_t)r16 * ldx + k) : zero_frag();
            }
        auto load_block = [&](frag_t (&w)[NP][2], int nb) {
            int n = nb + r16;
            n = n < N ? n : N - 1;                                      // clamped rows only feed outputs that are never stored
            const T* row = W + (int64_t)n * K + k0 + g * (2 * EPC);
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int k = k0 + p * (8 * EPC) + g * (2 * EPC) + h * EPC;
                    w[p][h] = (k < K) ? *reinterpret_cast<const frag_t*>(row + p * (8 * EPC) + h * EPC) : zero_frag();
                }
        };
        auto compute = [&](const frag_t (&w)[NP][2], int nb) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int h = 0; h < 2; ++h) acc = Mma16<T>::mma(w[p][h], xf[p][h], acc);
            // lane (m = r16, g) holds the partial of rows nb + 4 g + r
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float* dst = &red[wave][nb - n0 + 4 * g + r][r16];
                *dst = first ? acc[r] : *dst + acc[r];
            }
        };
        frag_t w0[NP][2], w1[NP][2];
        int nb = n0;
        if (nb < n1) load_block(w0, nb);
        while (nb < n1) {
            if (nb + 16 < n1) load_block(w1, nb + 16);
            compute(w0, nb);
            if (nb + 16 >= n1) break;
            if (nb + 32 < n1) load_block(w0, nb + 32);
            compute(w1, nb + 16);
            nb += 32;
        }
        first = false;
    }
    __syncthreads();
    const int nw = nslab < 8 ? nslab : 8;                                // waves that wrote a region
    for (int i = threadIdx.x; i < (n1 - n0) * 16; i += 512) {
        const int row = i >> 4, m = i & 15;
        if (m >= M) continue;
        float sum = red[0][row][m];
        for (int w = 1; w < nw; ++w) sum += red[w][row][m];
        if (bias) sum += to_f32(bias[n0 + row]);
        z[(int64_t)m * ldz + n0 + row] = from_f32<TO>(sum);
    }
}

// 2-D form of the MFMA forward for N < 32768: one-wave workgroups, (k-slab x row-range) grid, ~4 independent waves per CU at
// different phases (on these sizes a CU sees only one or two of the 8-wave workgroups above, and their x-fragment prologue /
// LDS epilogue is not hidden by anything: 3.6 TB/s vs 4.6 TB/s in this form on the [14336, 4096] gate/up weight).  k-slab
// partials [k_slabs][M][N] fp32 are summed (+ bias, cast) by smallm_fwd_sum_kernel.
template <typename T>
__global__ __launch_bounds__(64, 2) void smallm_fwd2d_kernel(const T* __restrict__ W, const T* __restrict__ x, float* __restrict__ part,
                                                             int M, int N, int K, int64_t ldx, int rows_per) {
    constexpr int EPC = 16 / (int)sizeof(T), SLAB = 1024 / (int)sizeof(T), NP = 8;
    typedef typename Mma16<T>::frag frag_t;
    const int lane = threadIdx.x, r16 = lane & 15, g = lane >> 4;
    const int k0 = blockIdx.x * SLAB;
    const int n0 = blockIdx.y * rows_per, n1 = min(N, n0 + rows_per);
    auto zero_frag = [] { u32x4 zz = {0, 0, 0, 0}; return __builtin_bit_cast(frag_t, zz); };
    frag_t xf[NP][2];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = k0 + p * (8 * EPC) + g * (2 * EPC) + h * EPC;
            xf[p][h] = (r16 < M && k < K) ? *reinterpret_cast<const frag_t*>(x + (int64_t)r16 * ldx + k) : zero_frag();
        }
    auto load_block = [&](frag_t (&w)[NP][2], int nb) {
        int n = nb + r16;
        n = n < N ? n : N - 1;
        const T* row = W + (int64_t)n * K + k0 + g * (2 * EPC);
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k = k0 + p * (8 * EPC) + g * (2 * EPC) + h * EPC;
                w[p][h] = (k < K) ? *reinterpret_cast<const frag_t*>(row + p * (8 * EPC) + h * EPC) : zero_frag();
            }
    };
    auto compute_store = [&](const frag_t (&w)[NP][2], int nb) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int h = 0; h < 2; ++h) acc = Mma16<T>::mma(w[p][h], xf[p][h], acc);
        if (r16 < M) {                                   // lane (m = r16, g) holds z_partial[n = nb + 4 g + r][m]
            float* dst = part + ((int64_t)blockIdx.x * M + r16) * N + nb + 4 * g;
            if (nb + 4 * g + 3 < n1 && ((N & 3) == 0)) *reinterpret_cast<f32x4*>(dst) = acc;
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (nb + 4 * g + r < n1) dst[r] = acc[r];
            }
        }
    };
    frag_t w0[NP][2], w1[NP][2];
    int nb = n0;
    if (nb < n1) load_block(w0, nb);
    while (nb < n1) {
        if (nb + 16 < n1) load_block(w1, nb + 16);
        compute_store(w0, nb);
        if (nb + 16 >= n1) break;
        if (nb + 32 < n1) load_block(w0, nb + 32);
        compute_store(w1, nb + 16);
        nb += 32;
    }
}

// z[m,n] = sum_ks part[ks][m][n] + bias[n]
template <typename T, typename TO>
__global__ void smallm_fwd_sum_kernel(const float* __restrict__ part, const T* __restrict__ bias, TO* __restrict__ z, int nks, int M,
                                      int N, int64_t ldz) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)M * N) return;
    const int m = (int)(i / N), n = (int)(i % N);
    float s = bias ? to_f32(bias[n]) : 0.f;
    for (int b = 0; b < nks; ++b) s += part[(int64_t)b * M * N + i];
    z[(int64_t)m * ldz + n] = from_f32<TO>(s);
}

inline void smallm_fwd2d_geometry(int N, int K, int sz, int& ks, int& rows_per, int& nr) {
    ks = (K * sz + 1023) / 1024;
    int want = (1024 + ks - 1) / ks;                                   // ~1024 one-wave workgroups on the chip
    const int max_nr = (N + 31) / 32;
    if (want > max_nr) want = max_nr;
    if (want < 1) want = 1;
    rows_per = ((N + want - 1) / want + 15) / 16 * 16;
    nr = (N + rows_per - 1) / rows_per;
}
constexpr int FWD_WG_N = 32768;             // from this many rows on: the 8-wave single-launch form

inline int smallm_row_ranges(int N, int kslabs) {
    // ~one 8-wave workgroup per CU, every wave with at least one batch of 16 rows; the number of row ranges is the number of
    // fp32 slabs the second stage sums (slab traffic = ranges x M x K x 8 bytes: 3 % of W at M = 4 on the gate/up shape)
    int nr = (256 + kslabs - 1) / kslabs;
    const int max_nr = (N + 8 * SM_R - 1) / (8 * SM_R);
    if (nr > max_nr) nr = max_nr;
    return nr < 1 ? 1 : nr;
}

template <typename T, typename TO, int MM>
int launch_dgrad(const void* W, const void* g, const void* z, const void* x, void* out, float* ws, int M, int N, int K, int64_t ldg,
                 int64_t ldz, float eps, int rel_in, int rel_out, hipStream_t st) {
    constexpr int EPC = 16 / (int)sizeof(T);
    const int ks = (K + 64 * EPC - 1) / (64 * EPC), nr = smallm_row_ranges(N, ks);
    const int rows_per = ((N + nr - 1) / nr + SM_R - 1) / SM_R * SM_R;
    const int nr2 = (N + rows_per - 1) / rows_per;
    dim3 grid(ks, nr2);
    if (z) hipLaunchKernelGGL((smallm_dgrad_kernel<T, MM, true>), grid, dim3(512), 0, st, (const T*)W, (const T*)g, (const T*)z, ws, M, N, K, ldg, ldz, rows_per, eps, rel_in);
    else hipLaunchKernelGGL((smallm_dgrad_kernel<T, MM, false>), grid, dim3(512), 0, st, (const T*)W, (const T*)g, (const T*)nullptr, ws, M, N, K, ldg, ldz, rows_per, eps, rel_in);
    const int64_t MK = (int64_t)M * K;
    hipLaunchKernelGGL((smallm_slab_sum_kernel<T, TO>), dim3((unsigned)((MK + 63) / 64)), dim3(256), 0, st, ws, (const T*)x, (TO*)out, nr2, MK, (x && rel_out) ? 1 : 0);
    return lrp_check_launch();
}

template <typename T, typename TO>
int launch_fwd(const void* W, const void* x, const void* bias, void* z, float* ws, int M, int N, int K, int64_t ldx, int64_t ldz,
               hipStream_t st) {
    if (N < FWD_WG_N) {
        int ks, rows_per, nr;
        smallm_fwd2d_geometry(N, K, (int)sizeof(T), ks, rows_per, nr);
        hipLaunchKernelGGL((smallm_fwd2d_kernel<T>), dim3(ks, nr), dim3(64), 0, st, (const T*)W, (const T*)x, ws, M, N, K, ldx, rows_per);
        const int64_t MN = (int64_t)M * N;
        hipLaunchKernelGGL((smallm_fwd_sum_kernel<T, TO>), dim3((unsigned)((MN + 255) / 256)), dim3(256), 0, st, ws, (const T*)bias, (TO*)z, ks,
                           M, N, ldz);
        return lrp_check_launch();
    }
    const int nslab = (K * (int)sizeof(T) + 1023) / 1024;
    // ~1-2 workgroups (8 waves) per CU, at least two 16-row blocks per workgroup, at most FW_ROWS rows
    int rows_per = ((N + 383) / 384 + 15) / 16 * 16;
    if (rows_per < 32) rows_per = 32;
    if (rows_per > FW_ROWS) rows_per = FW_ROWS;
    const int nwg = (N + rows_per - 1) / rows_per;
    hipLaunchKernelGGL((smallm_fwd_kernel<T, TO>), dim3(nwg), dim3(512), 0, st, (const T*)W, (const T*)x, (const T*)bias, (TO*)z, M, N, K,
                       ldx, ldz, rows_per, nslab);
    return lrp_check_launch();
}

#define SMALLM_MM(M_, CALL)                      \
    if (M_ <= 1) { constexpr int MM = 1; CALL }  \
    else if (M_ <= 2) { constexpr int MM = 2; CALL } \
    else if (M_ <= 4) { constexpr int MM = 4; CALL } \
    else if (M_ <= 8) { constexpr int MM = 8; CALL } \
    else { constexpr int MM = 16; CALL }

}  // namespace

extern "C" int64_t lrp_linear_smallm_ws(int M, int N, int K, int dtype) {
    const int epc = dtype == LRP_F32 ? 4 : 8;
    const int ks = (K + 64 * epc - 1) / (64 * epc);
    const int nr = smallm_row_ranges(N, ks);
    int64_t fwd = 0;
    if (N < FWD_WG_N) {
        int fks, frp, fnr;
        smallm_fwd2d_geometry(N, K, dtype == LRP_F32 ? 4 : 2, fks, frp, fnr);
        fwd = (int64_t)fks * M * N;
    }
    const int64_t dgrad = (int64_t)nr * M * K;
    return dgrad > fwd ? dgrad : fwd;
}

extern "C" int lrp_linear_smallm_fwd(const void* x, const void* W, const void* bias, void* z, float* workspace, int M, int N, int K,
                                     int64_t ldx, int64_t ldz, int dtype, int out_dtype, void* stream) {
    if (!x || !W || !z || M < 1 || N < 1 || K < 1) return LRP_EINVAL;
    if ((dtype != LRP_F32 && dtype != LRP_BF16) || (out_dtype != dtype && out_dtype != LRP_F32)) return LRP_EINVAL;
    const int epc = dtype == LRP_F32 ? 4 : 8;
    if ((K % epc) || (ldx % epc) || (reinterpret_cast<uintptr_t>(W) & 15) || (reinterpret_cast<uintptr_t>(x) & 15)) return LRP_EALIGN;
    if (M > 16) return LRP_ESHAPE;
    hipStream_t st = (hipStream_t)stream;
    if (N < FWD_WG_N && !workspace) return LRP_EINVAL;
    if (dtype == LRP_F32) return launch_fwd<float, float>(W, x, bias, z, workspace, M, N, K, ldx, ldz, st);
    if (out_dtype == LRP_F32) return launch_fwd<bf16_t, float>(W, x, bias, z, workspace, M, N, K, ldx, ldz, st);
    return launch_fwd<bf16_t, bf16_t>(W, x, bias, z, workspace, M, N, K, ldx, ldz, st);
}

extern "C" int lrp_linear_smallm_dgrad(const void* g, const void* z, const void* W, const void* x, void* out, float* workspace,
                                       int M, int N, int K, int64_t ldg, int64_t ldz, float eps, int relevance_in, int relevance_out,
                                       int dtype, int out_dtype, void* stream) {
    if (!g || !W || !out || !workspace || M < 1 || N < 1 || K < 1) return LRP_EINVAL;
    if ((dtype != LRP_F32 && dtype != LRP_BF16) || (out_dtype != dtype && out_dtype != LRP_F32)) return LRP_EINVAL;
    const int epc = dtype == LRP_F32 ? 4 : 8;
    if ((K % epc) || (reinterpret_cast<uintptr_t>(W) & 15)) return LRP_EALIGN;
    if (M > 16) return LRP_ESHAPE;
    if (relevance_out && !x) return LRP_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == LRP_F32) { SMALLM_MM(M, return (launch_dgrad<float, float, MM>(W, g, z, x, out, workspace, M, N, K, ldg, ldz, eps, relevance_in, relevance_out, st));) }
    if (out_dtype == LRP_F32) { SMALLM_MM(M, return (launch_dgrad<bf16_t, float, MM>(W, g, z, x, out, workspace, M, N, K, ldg, ldz, eps, relevance_in, relevance_out, st));) }
    SMALLM_MM(M, return (launch_dgrad<bf16_t, bf16_t, MM>(W, g, z, x, out, workspace, M, N, K, ldg, ldz, eps, relevance_in, relevance_out, st));)
}
