// linear_smallm.hip -- K1 in its HBM-bound regime: the Linear eps-rule for M <= 8 rows in ONE
// pass over W.  ref: lxt/explicit/functional.py:345-364 (z = xW^T+b ; s = R/(z+eps) ; R_in = x*(sW)).
//
// z_n needs a full K reduction before s_n exists while c_k = sum_n s_n W[n,k] reduces over N, so a
// single pass over W must be N-blocked: a WAVE owns one row of W at a time, keeps it in registers
// (K*sizeof/1 KiB x 16-byte chunks per lane), computes z_n for every m with a wave reduction,
// forms s_n, and immediately re-uses the registers for the axpy into per-lane partial c.  Rows are
// streamed straight to VGPRs (no LDS round trip: nothing is shared between waves), x sits in LDS,
// and the split-N partial sums are combined with fp32 atomics into the zeroed output.
// Algorithmic HBM bytes = sizeof(T)*(N*K + 2*M*K + M*N): W is read exactly once.
#include "common.hpp"

namespace {

template <typename T, int KCH, int MM>
__global__ __launch_bounds__(256) void linear_eps_smallm_kernel(
    const T* __restrict__ x, const T* __restrict__ W, const T* __restrict__ bias, const T* __restrict__ g,
    float* __restrict__ out, float* __restrict__ z_out, int M, int N, int K, float eps, int rel_in, int rel_out) {
    constexpr int EPC = 16 / (int)sizeof(T);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* sx = reinterpret_cast<T*>(smem);                       // [MM][KCH*64*EPC], zero padded
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int KP = KCH * 64 * EPC;
    for (int i = threadIdx.x; i < MM * KP; i += blockDim.x) {
        const int m = i / KP, k = i % KP;
        sx[i] = (m < M && k < K) ? x[(int64_t)m * K + k] : from_f32<T>(0.f);
    }
    __syncthreads();

    float acc[MM][KCH][EPC];
#pragma unroll
    for (int m = 0; m < MM; ++m)
#pragma unroll
        for (int j = 0; j < KCH; ++j)
#pragma unroll
            for (int e = 0; e < EPC; ++e) acc[m][j][e] = 0.f;

    const int nwave = gridDim.x * 4;
    for (int n = blockIdx.x * 4 + wave; n < N; n += nwave) {
        Vec16<T> w[KCH];
#pragma unroll
        for (int j = 0; j < KCH; ++j) {
            const int k = (j * 64 + lane) * EPC;
            if (k < K) w[j] = ld16(W + (int64_t)n * K + k);
            else {
#pragma unroll
                for (int e = 0; e < EPC; ++e) w[j].set(e, 0.f);
            }
        }
        float s[MM];
#pragma unroll
        for (int m = 0; m < MM; ++m) {
            float d = 0.f;
#pragma unroll
            for (int j = 0; j < KCH; ++j) {
                const Vec16<T> xv = ld16(sx + m * KP + (j * 64 + lane) * EPC);
#pragma unroll
                for (int e = 0; e < EPC; ++e) d += w[j].get(e) * xv.get(e);
            }
            d = wave_sum(d);
            float z = d + (bias ? to_f32(bias[n]) : 0.f);
            z = to_f32(from_f32<T>(z));                              // the forward's stored z (storage dtype)
            if (m < M) {
                if (z_out && lane == 0) z_out[(int64_t)m * N + n] = z;
                const float gv = to_f32(g[(int64_t)m * N + n]);
                s[m] = rel_in ? gv / (z + eps) : gv * eps_ratio(z, 1.f, eps);
            } else s[m] = 0.f;
        }
#pragma unroll
        for (int m = 0; m < MM; ++m)
#pragma unroll
            for (int j = 0; j < KCH; ++j)
#pragma unroll
                for (int e = 0; e < EPC; ++e) acc[m][j][e] += s[m] * w[j].get(e);
    }
#pragma unroll
    for (int m = 0; m < MM; ++m) {
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < KCH; ++j) {
            const int k = (j * 64 + lane) * EPC;
            if (k >= K) continue;
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                float v = acc[m][j][e];
                if (rel_out) v *= to_f32(sx[m * KP + k + e]);
                atomicAdd(out + (int64_t)m * K + k + e, v);
            }
        }
    }
}

template <typename T, int KCH, int MM>
int launch(const void* x, const void* W, const void* bias, const void* g, float* out, float* z_out, int M, int N, int K,
           float eps, int rel_in, int rel_out, hipStream_t st) {
    constexpr int EPC = 16 / (int)sizeof(T);
    const size_t lds = (size_t)MM * KCH * 64 * EPC * sizeof(T);
    auto kern = linear_eps_smallm_kernel<T, KCH, MM>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int nb = (N + 3) / 4;
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(kern, dim3(nb), dim3(256), lds, st, (const T*)x, (const T*)W, (const T*)bias, (const T*)g, out, z_out,
                       M, N, K, eps, rel_in, rel_out);
    return lrp_check_launch();
}

template <typename T, int MM>
int launch_k(int kch, const void* x, const void* W, const void* bias, const void* g, float* out, float* z_out, int M, int N,
             int K, float eps, int rel_in, int rel_out, hipStream_t st) {
    constexpr int EPC = 16 / (int)sizeof(T);
    if (kch <= 1) return launch<T, 1, MM>(x, W, bias, g, out, z_out, M, N, K, eps, rel_in, rel_out, st);
    if (kch <= 2) return launch<T, 2, MM>(x, W, bias, g, out, z_out, M, N, K, eps, rel_in, rel_out, st);
    if (kch <= 4) return launch<T, 4, MM>(x, W, bias, g, out, z_out, M, N, K, eps, rel_in, rel_out, st);
    if constexpr (MM * 8 * EPC <= 128) {
        if (kch <= 8) return launch<T, 8, MM>(x, W, bias, g, out, z_out, M, N, K, eps, rel_in, rel_out, st);
    }
    if constexpr (MM * 16 * EPC <= 128) {
        if (kch <= 16) return launch<T, 16, MM>(x, W, bias, g, out, z_out, M, N, K, eps, rel_in, rel_out, st);
    }
    return LRP_ESHAPE;
}

}  // namespace

extern "C" int lrp_linear_eps_smallm(const void* x, const void* W, const void* bias, const void* g, float* out, float* z_out,
                                     int M, int N, int K, float eps, int relevance_in, int relevance_out, int dtype,
                                     void* stream) {
    if (!x || !W || !g || !out || M < 1 || N < 1 || K < 1) return LRP_EINVAL;
    if (dtype != LRP_F32 && dtype != LRP_BF16) return LRP_EINVAL;
    const int epc = dtype == LRP_F32 ? 4 : 8;
    if ((K % epc) || (reinterpret_cast<uintptr_t>(W) & 15)) return LRP_EALIGN;
    if (M > 4) return LRP_ESHAPE;
    const int kch = (K + 64 * epc - 1) / (64 * epc);
    hipStream_t st = (hipStream_t)stream;
#define GO(T, MM) return launch_k<T, MM>(kch, x, W, bias, g, out, z_out, M, N, K, eps, relevance_in, relevance_out, st)
    if (dtype == LRP_F32) {
        if (M == 1) GO(float, 1);
        if (M == 2) GO(float, 2);
        GO(float, 4);
    } else {
        if (M == 1) GO(bf16_t, 1);
        if (M == 2) GO(bf16_t, 2);
        GO(bf16_t, 4);
    }
#undef GO
}
