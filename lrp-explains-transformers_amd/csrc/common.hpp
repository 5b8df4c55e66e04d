// common.hpp -- shared device helpers for the gfx950 (CDNA4, wave64) AttnLRP kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/lrp_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define LRP_DEVICE __device__ __forceinline__

extern thread_local int g_lrp_last_hip_error;

// opt a kernel in to more than 64 KiB of dynamic LDS, once per DEVICE (the attribute is per device; a process that drives two devices
// must set it on both) and safely from several host threads (main thread + autograd worker): one bit per device in a per-call-site mask
#include <atomic>
static inline void lrp_set_max_lds_once(std::atomic<uint64_t>& done, const void* kern, size_t bytes) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return;
    (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    done.fetch_or(bit, std::memory_order_release);
}
#define LRP_SET_MAX_LDS(kern, bytes)                                                       \
    do {                                                                                   \
        static std::atomic<uint64_t> lds_done_{0};                                         \
        lrp_set_max_lds_once(lds_done_, reinterpret_cast<const void*>(kern), (bytes));     \
    } while (0)

// compute units of the current device (one query per device and process)
static inline int lrp_num_cus() {
    static std::atomic<int> cached[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    int v = cached[dev & 63].load(std::memory_order_relaxed);
    if (v > 0) return v;
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cached[dev & 63].store(n, std::memory_order_relaxed);
    return n;
}

static inline int lrp_check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_lrp_last_hip_error = (int)e; return LRP_ELAUNCH; }
    return LRP_OK;
}

// ---- scalar conversion --------------------------------------------------------------------
LRP_DEVICE float to_f32(float x) { return x; }
LRP_DEVICE float to_f32(bf16_t x) { return (float)x; }
template <typename T> LRP_DEVICE T from_f32(float x);
template <> LRP_DEVICE float from_f32<float>(float x) { return x; }
template <> LRP_DEVICE bf16_t from_f32<bf16_t>(float x) { return (bf16_t)x; }  // RNE

// ---- 16-byte vectors of T: 4 floats or 8 bf16 -----------------------------------------------
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    static constexpr int N = 4;
    f32x4 v;
    LRP_DEVICE float get(int i) const { return v[i]; }
    LRP_DEVICE void set(int i, float x) { v[i] = x; }
};
template <> struct Vec16<bf16_t> {
    static constexpr int N = 8;
    bf16x8 v;
    LRP_DEVICE float get(int i) const { return (float)v[i]; }
    LRP_DEVICE void set(int i, float x) { v[i] = (bf16_t)x; }
};
template <typename T> LRP_DEVICE Vec16<T> ld16(const T* p) {
    Vec16<T> r;
    r.v = *reinterpret_cast<const decltype(r.v)*>(p);
    return r;
}
template <typename T> LRP_DEVICE void st16(T* p, const Vec16<T>& r) {
    *reinterpret_cast<decltype(r.v)*>(p) = r.v;
}

// ---- wave64 reductions ----------------------------------------------------------------------
LRP_DEVICE float wave_sum(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}
LRP_DEVICE float wave_max(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o, 64));
    return x;
}
// block reduction through LDS; red must hold >= blockDim/64 floats; returns value to all threads
LRP_DEVICE float block_sum(float x, float* red) {
    x = wave_sum(x);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = x;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}
LRP_DEVICE float block_max(float x, float* red) {
    x = wave_max(x);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = x;
    __syncthreads();
    float t = red[0];
    for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
    return t;
}

// ---- the eps stabiliser ratio  z/(c z + eps)  (unsigned eps, as the reference) ---------------
LRP_DEVICE float eps_ratio(float z, float c, float eps) {
    // eps == 0: lxt.efficient has no stabiliser at all -> exactly 1/c (never 0/0)
    return (eps == 0.f) ? (1.f / c) : z / (c * z + eps);
}

LRP_DEVICE float act_apply(float x, int act) {
    if (act == LRP_ACT_SILU) return x / (1.f + __expf(-x));
    if (act == LRP_ACT_GELU_TANH) {
        const float k0 = 0.7978845608028654f, k1 = 0.044715f;
        return 0.5f * x * (1.f + tanhf(k0 * (x + k1 * x * x * x)));
    }
    if (act == LRP_ACT_TANH) return tanhf(x);
    return 0.5f * x * (1.f + erff(x * 0.7071067811865476f));
}

// ---- the same rules for bf16 STORAGE: every result is rounded to 8 mantissa bits, so v_rcp_f32 / v_exp_f32 (1 ulp of fp32) replace the
// IEEE division sequences (10-12 VALU each; three of them per element made the fused gated epilogue VALU-bound: 97 instructions per
// element).  FAST = false keeps the exact fp32 forms (fp32 storage: the explicit-mode parity paths).
template <bool FAST> LRP_DEVICE float fdiv_t(float a, float b) {
    if constexpr (FAST) return a * __builtin_amdgcn_rcpf(b);
    else return a / b;
}
// a / b for a denominator that may be a denormal (v_rcp_f32 flushes those to +-inf): rescaled by 2^24 first
template <bool FAST> LRP_DEVICE float fdiv_small_t(float a, float b) {
    if constexpr (FAST) {
        const float sc = (fabsf(b) < 1.17549435e-38f) ? 16777216.f : 1.f;
        return a * (__builtin_amdgcn_rcpf(b * sc) * sc);
    } else return a / b;
}
template <bool FAST> LRP_DEVICE float eps_ratio_t(float z, float c, float eps) {
    return (eps == 0.f) ? fdiv_t<FAST>(1.f, c) : fdiv_t<FAST>(z, c * z + eps);
}
template <bool FAST> LRP_DEVICE float act_apply_t(float x, int act) {
    if constexpr (FAST) {
        if (act == LRP_ACT_SILU) return x * __builtin_amdgcn_rcpf(1.f + __expf(-x));
        if (act == LRP_ACT_GELU_TANH) {                                  // 0.5 x (1 + tanh z) = x sigmoid(2 z)
            const float k0 = 0.7978845608028654f, k1 = 0.044715f;
            return x * __builtin_amdgcn_rcpf(1.f + __expf(-2.f * k0 * (x + k1 * x * x * x)));
        }
    }
    return act_apply(x, act);
}

// ---- the gated-MLP backward rule on ONE (gate, up) pair, bf16 STORAGE: identity rule on act (y / (g + eps_g), 0 where g + eps_g = 0) and uniform
// rule on the product, the up-projection's Linear stabiliser u / (u + eps_lin) folded in (ref lxt/efficient/patches.py:145-157,
// lxt/explicit/models/llama.py:84-86,273-281).  Shared by the fused down-projection dgrad epilogue (gemm_pp.hip, EPI 2) and the stand-alone
// lrp_gated_act_bwd / _il kernels (eltwise.hip): the two must agree bit for bit.  gmh = 0.5 * (bf16-rounded Gm); y is rounded to bf16 (what
// the forward stored in m = y * u was formed from).  LEAN = (eps_g >= 1e-30 and eps_lin == 0), the lxt.efficient placement: g + eps_g is then
// never a denormal (no 2^24 rescue of v_rcp_f32) and the up-projection factor is exactly 1 -- bit-identical to the general form on those
// arguments, at 3 transcendentals + ~19 plain VALU per pair instead of 4 + ~34 (the fused epilogue was VALU-bound: ~640 instructions per 16
// pairs and wave, 20 of the 26 us a tile's epilogue took; profiles/r05_epi2_valu.txt).
template <bool LEAN, int ACT_CT = -1>
LRP_DEVICE void gated_bwd_pair_bf16(float g, float u, float gmh, float eps_g, float eps_lin, int act, float& ag, float& au) {
    const int a_ = ACT_CT >= 0 ? ACT_CT : act;
    const float y = (float)(bf16_t)act_apply_t<true>(g, a_);
    const float den = g + eps_g;
    if constexpr (LEAN) {
        const float q = y * __builtin_amdgcn_rcpf(den);
        ag = gmh * u * ((den == 0.f) ? 0.f : q);
        au = gmh * y;
    } else {
        ag = (den == 0.f) ? 0.f : gmh * u * fdiv_small_t<true>(y, den);
        au = gmh * y * eps_ratio_t<true>(u, 1.f, eps_lin);
    }
}

// ---- the same rule as COEFFICIENTS (round 6; gemm_pp.hip EPI 1 / 2): the gate/up forward, which has g and u in fp32 registers, evaluates the
// activation ONCE and leaves the two factors the backward multiplies Gm by -- cg = 1/2 u act(g) / (g + eps_g) (0 where g + eps_g = 0) and
// cu = 1/2 act(g) u / (u + eps_lin) (1/2 act(g) for eps_lin = 0) -- next to m = act(g) u.  LEAN as above.
template <bool LEAN, int ACT_CT>
LRP_DEVICE void gated_coef(float g, float u, float eps_g, float eps_lin, float& m, float& cg, float& cu) {
    const float y = act_apply_t<true>(g, ACT_CT);
    m = y * u;
    const float den = g + eps_g, hy = 0.5f * y;
    if constexpr (LEAN) {
        const float q = hy * __builtin_amdgcn_rcpf(den);
        cg = (den == 0.f) ? 0.f : u * q;
        cu = hy;
    } else {
        cg = (den == 0.f) ? 0.f : u * fdiv_small_t<true>(hy, den);
        cu = hy * eps_ratio_t<true>(u, 1.f, eps_lin);
    }
}
// the two bf16 halves of a 32-bit word as fp32 (one VALU each)
LRP_DEVICE float bf16_lo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
LRP_DEVICE float bf16_hi(uint32_t w) { return __builtin_bit_cast(float, w & 0xffff0000u); }

// ---- MFMA 16x16 "macro" op, identical byte geometry for both dtypes ---------------------------
// One macro step contracts a 64-BYTE K chunk: lane l supplies the 16 bytes at K-byte offset
// (l>>4)*16 of row (l&15) for each operand (8 bf16 / 4 fp32).  bf16: one v_mfma_f32_16x16x32_bf16;
// fp32: four v_mfma_f32_16x16x4_f32 (element e of the float4 feeds MFMA e, whose k-slot l>>4 is
// then k = (l>>4)*4+e for BOTH operands, so the pairing is consistent).
// D layout (both): lane l holds D[i = (l>>4)*4 + r][j = l&15], r = 0..3, where i indexes the rows of
// the FIRST argument and j the rows of the SECOND.
template <typename T> struct Mma16;
template <> struct Mma16<bf16_t> {
    typedef bf16x8 frag;
    static LRP_DEVICE f32x4 mma(frag a, frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mma16<float> {
    typedef f32x4 frag;
    static LRP_DEVICE f32x4 mma(frag a, frag b, f32x4 c) {
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
        return c;
    }
};

// Grouped tile order: consecutive ids sweep GROUP_M tile-rows of one tile-column, then the next
// column, so the ~32-64 workgroups resident on one XCD cover a compact GROUP_M x n block of output
// tiles and share A/B panels through that XCD's L2 instead of each pulling its own from HBM/MALL.
LRP_DEVICE void grouped_tile(int id, int tiles_m, int tiles_n, int& tm, int& tn) {
    constexpr int GROUP_M = 8;
    const int per_group = GROUP_M * tiles_n;
    const int group = id / per_group;
    const int first_m = group * GROUP_M;
    const int gsz = (tiles_m - first_m) < GROUP_M ? (tiles_m - first_m) : GROUP_M;
    const int in_group = id - group * per_group;
    tm = first_m + in_group % gsz;
    tn = in_group / gsz;
}

// XCD-aware bijective remap of a linear workgroup id (8 XCDs; block b runs on XCD b%8):
// consecutive remapped ids land on the same XCD so neighbouring tiles share that XCD's L2.
LRP_DEVICE int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + loc;
}
