// eltwise.hip -- HBM-bound element-wise LRP rules (K1 eps scale, K3/K8 activation + uniform rule,
// K5 RoPE, add2 rule, casts).  Every kernel is one read + one write per element, 16-byte vector
// accesses per lane (8 bf16 / 4 fp32) in a grid-stride loop capped at 2048 blocks; a scalar
// variant of each covers unaligned / ragged shapes of the explicit rule API.
#include "common.hpp"

namespace {

constexpr int ENT = 256;
inline int grid_for(int64_t work) {
    int64_t b = (work + ENT - 1) / ENT;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}
inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// chunk of W elements: W = 16/sizeof(T) (vector) or 1 (scalar)
template <typename T, int W> struct Chunk {
    float v[W];
    LRP_DEVICE void load(const T* p) {
        if constexpr (W == 1) { v[0] = to_f32(p[0]); }
        else { Vec16<T> t = ld16(p);
#pragma unroll
            for (int i = 0; i < W; ++i) v[i] = t.get(i); }
    }
    LRP_DEVICE void store(T* p) const {
        if constexpr (W == 1) { p[0] = from_f32<T>(v[0]); }
        else { Vec16<T> t;
#pragma unroll
            for (int i = 0; i < W; ++i) t.set(i, v[i]);
            st16(p, t); }
    }
};

// ------------------------------------------------------------------------------------------------
template <typename T, int W>
__global__ void eps_scale_kernel(const T* g, const T* z, T* out, int64_t n, float c, float eps, int mode) {
    const int64_t nchunk = n / W;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nchunk; i += (int64_t)gridDim.x * blockDim.x) {
        Chunk<T, W> a, b, o;
        a.load(g + i * W);
        b.load(z + i * W);
#pragma unroll
        for (int k = 0; k < W; ++k)
            o.v[k] = (mode == 0) ? a.v[k] * eps_ratio(b.v[k], c, eps) : a.v[k] / (c * b.v[k] + eps);
        o.store(out + i * W);
    }
}

template <typename T, int W>
__global__ void eps_scale2d_kernel(const T* g, const T* z, T* out, int rows, int cols, int64_t ldg, int64_t ldz,
                                   int64_t ldo, float c, float eps, int mode) {
    const int cpr = cols / W;
    const int64_t total = (int64_t)rows * cpr;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cpr;
        const int cc = (int)(i - r * cpr) * W;
        Chunk<T, W> a, b, o;
        a.load(g + r * ldg + cc);
        b.load(z + r * ldz + cc);
#pragma unroll
        for (int k = 0; k < W; ++k)
            o.v[k] = (mode == 0) ? a.v[k] * eps_ratio(b.v[k], c, eps) : a.v[k] / (c * b.v[k] + eps);
        o.store(out + r * ldo + cc);
    }
}

template <typename T, int W>
__global__ void mul_kernel(const T* a_, const T* b_, T* out, int64_t n) {
    const int64_t nchunk = n / W;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nchunk; i += (int64_t)gridDim.x * blockDim.x) {
        Chunk<T, W> a, b, o;
        a.load(a_ + i * W);
        b.load(b_ + i * W);
#pragma unroll
        for (int k = 0; k < W; ++k) o.v[k] = a.v[k] * b.v[k];
        o.store(out + i * W);
    }
}

// out[m,:] = x[m,:] + y[m % period,:]
template <typename T, int W>
__global__ void add_bcast_kernel(const T* x, const T* y, T* out, int M, int H, int period) {
    const int cpr = H / W;
    const int64_t total = (int64_t)M * cpr;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cpr;
        const int c = (int)(i - r * cpr) * W;
        Chunk<T, W> a, b, o;
        a.load(x + r * H + c);
        b.load(y + (r % period) * H + c);
#pragma unroll
        for (int k = 0; k < W; ++k) o.v[k] = a.v[k] + b.v[k];
        o.store(out + r * H + c);
    }
}

template <typename T, int W>
__global__ void add2_rule_kernel(const T* a_, const T* b_, const T* R_, T* Ra, T* Rb, int64_t n, float eps) {
    const int64_t nchunk = n / W;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nchunk; i += (int64_t)gridDim.x * blockDim.x) {
        Chunk<T, W> a, b, r, oa, ob;
        a.load(a_ + i * W);
        b.load(b_ + i * W);
        r.load(R_ + i * W);
#pragma unroll
        for (int k = 0; k < W; ++k) {
            const float s = r.v[k] / (a.v[k] + b.v[k] + eps);
            oa.v[k] = s * a.v[k];
            ob.v[k] = s * b.v[k];
        }
        oa.store(Ra + i * W);
        if (Rb) ob.store(Rb + i * W);
    }
}

template <typename TI, typename TO>
__global__ void cast_kernel(const TI* in, TO* out, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = from_f32<TO>(to_f32(in[i]));
}

// ---- activations -------------------------------------------------------------------------------
template <typename T, int W>
__global__ void act_fwd_kernel(const T* x, T* y, int64_t n, int act) {
    const int64_t nchunk = n / W;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nchunk; i += (int64_t)gridDim.x * blockDim.x) {
        Chunk<T, W> a, o;
        a.load(x + i * W);
#pragma unroll
        for (int k = 0; k < W; ++k) o.v[k] = act_apply(a.v[k], act);
        o.store(y + i * W);
    }
}
template <typename T, int W>
__global__ void act_bwd_kernel(const T* Gy, const T* x, T* Gx, int64_t n, float eps_g, int act) {
    const int64_t nchunk = n / W;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nchunk; i += (int64_t)gridDim.x * blockDim.x) {
        Chunk<T, W> g, a, o;
        g.load(Gy + i * W);
        a.load(x + i * W);
#pragma unroll
        for (int k = 0; k < W; ++k) {
            // efficient rule keeps the ratio in the storage dtype's forward value: act(x)/(x+eps)
            const float y = to_f32(from_f32<T>(act_apply(a.v[k], act)));
            const float den = a.v[k] + eps_g;
            o.v[k] = (den == 0.f) ? 0.f : g.v[k] * (y / den);
        }
        o.store(Gx + i * W);
    }
}

// plain derivative of the activation (NO rule): Gx = Gy act'(x) -- the reference's gemma3 map patches nothing in modeling_siglip, so the
// image tower's GELU keeps its ordinary gradient (ref: lxt/efficient/models/gemma3.py:14-19)
LRP_DEVICE float act_deriv(float x, int act) {
    if (act == LRP_ACT_SILU) {
        const float sg = 1.f / (1.f + __expf(-x));
        return sg * (1.f + x * (1.f - sg));
    }
    if (act == LRP_ACT_GELU_TANH) {
        const float k0 = 0.7978845608028654f, k1 = 0.044715f;
        const float t = tanhf(k0 * (x + k1 * x * x * x));
        return 0.5f * (1.f + t) + 0.5f * x * (1.f - t * t) * k0 * (1.f + 3.f * k1 * x * x);
    }
    if (act == LRP_ACT_TANH) {
        const float t = tanhf(x);
        return 1.f - t * t;
    }
    return 0.5f * (1.f + erff(x * 0.7071067811865476f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}
template <typename T, int W>
__global__ void act_grad_kernel(const T* Gy, const T* x, T* Gx, int64_t n, int act) {
    const int64_t nchunk = n / W;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nchunk; i += (int64_t)gridDim.x * blockDim.x) {
        Chunk<T, W> g, a, o;
        g.load(Gy + i * W);
        a.load(x + i * W);
#pragma unroll
        for (int k = 0; k < W; ++k) o.v[k] = g.v[k] * act_deriv(a.v[k], act);
        o.store(Gx + i * W);
    }
}

// 2-D strided gated-MLP kernels: one chunk of W columns per thread-iteration
template <typename T, int W>
__global__ void gated_fwd_kernel(const T* g, const T* u, T* m, int M, int I, int64_t ldg, int64_t ldu, int64_t ldm, int act, int il) {
    const int cpr = I / W;
    const int64_t total = (int64_t)M * cpr;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cpr;
        const int c = (int)(i - r * cpr) * W;
        const int cg = il ? (c / il) * 2 * il + (c % il) : c;       // interleaved [g block | u block] layout of a fused gate/up output
        Chunk<T, W> a, b, o;
        a.load(g + r * ldg + cg);
        b.load(u + r * ldu + cg);
#pragma unroll
        for (int k = 0; k < W; ++k) {
            const float y = to_f32(from_f32<T>(act_apply_t<sizeof(T) == 2>(a.v[k], act)));   // HF rounds act(g) before the product
            o.v[k] = y * b.v[k];
        }
        o.store(m + r * ldm + c);
    }
}
template <typename T, int W>
__global__ void gated_bwd_kernel(const T* Gm, const T* g, const T* u, T* Ag, T* Au, int M, int I,
                                 int64_t ldgm, int64_t ldg, int64_t ldu, int64_t ldag, int64_t ldau,
                                 float eps_g, float eps_lin, int act, int il) {
    const int cpr = I / W;
    const int64_t total = (int64_t)M * cpr;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cpr;
        const int c = (int)(i - r * cpr) * W;
        const int cg = il ? (c / il) * 2 * il + (c % il) : c;
        Chunk<T, W> gm, a, b, og, ou;
        gm.load(Gm + r * ldgm + c);
        a.load(g + r * ldg + cg);
        b.load(u + r * ldu + cg);
        if constexpr (sizeof(T) == 2) {
            // bf16 storage: the rcp / exp forms shared with the fused epilogue of gemm_pp.hip (common.hpp: gated_bwd_pair_bf16), bit for bit
            if (eps_g >= 1e-30f && eps_lin == 0.f) {
#pragma unroll
                for (int k = 0; k < W; ++k) gated_bwd_pair_bf16<true>(a.v[k], b.v[k], 0.5f * gm.v[k], eps_g, eps_lin, act, og.v[k], ou.v[k]);
            } else {
#pragma unroll
                for (int k = 0; k < W; ++k) gated_bwd_pair_bf16<false>(a.v[k], b.v[k], 0.5f * gm.v[k], eps_g, eps_lin, act, og.v[k], ou.v[k]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < W; ++k) {
                const float y = act_apply(a.v[k], act);
                const float half = 0.5f * gm.v[k];
                const float den = a.v[k] + eps_g;
                og.v[k] = (den == 0.f) ? 0.f : half * b.v[k] * (y / den);
                ou.v[k] = half * y * eps_ratio(b.v[k], 1.f, eps_lin);
            }
        }
        og.store(Ag + r * ldag + cg);
        ou.store(Au + r * ldau + cg);
    }
}

// W consecutive fp32 table entries (16-byte loads when W is a multiple of 4: the tables are fp32 [seq, d], d even,
// c a multiple of W, so the address is 16-byte aligned whenever d % 4 == 0 -- checked by the host)
template <int W> LRP_DEVICE void load_tab(float (&t)[W], const float* p) {
    if constexpr (W % 4 == 0) {
#pragma unroll
        for (int q = 0; q < W / 4; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(p + 4 * q);
            t[4 * q] = v[0]; t[4 * q + 1] = v[1]; t[4 * q + 2] = v[2]; t[4 * q + 3] = v[3];
        }
    } else {
#pragma unroll
        for (int q = 0; q < W; ++q) t[q] = p[q];
    }
}

// ---- RoPE: thread handles W consecutive i in [0, d/2) of one (row, head): pairs (i, i+d/2) -------
template <typename T, int W>
__global__ void rope_fwd_kernel(const T* x, T* xr, const float* cs, const float* sn, int rows, int seq,
                                int nh, int d, int64_t ldx, int64_t ldxr) {
    const int hd = d / 2, cph = hd / W;
    const int64_t total = (int64_t)rows * nh * cph;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cph) * W;
        const int64_t rh = i / cph;
        const int h = (int)(rh % nh);
        const int64_t r = rh / nh;
        const int pos = (int)(r % seq);
        const T* px = x + r * ldx + (int64_t)h * d;
        T* po = xr + r * ldxr + (int64_t)h * d;
        Chunk<T, W> a, b, oa, ob;
        a.load(px + c);
        b.load(px + c + hd);
        float c1[W], c2[W], s1[W], s2[W];
        load_tab<W>(c1, cs + (int64_t)pos * d + c);
        load_tab<W>(c2, cs + (int64_t)pos * d + c + hd);
        load_tab<W>(s1, sn + (int64_t)pos * d + c);
        load_tab<W>(s2, sn + (int64_t)pos * d + c + hd);
#pragma unroll
        for (int k = 0; k < W; ++k) {
            oa.v[k] = a.v[k] * c1[k] - b.v[k] * s1[k];
            ob.v[k] = b.v[k] * c2[k] + a.v[k] * s2[k];
        }
        oa.store(po + c);
        ob.store(po + c + hd);
    }
}
template <typename T, int W>
__global__ void rope_bwd_kernel(const T* Gr, const T* xr, const T* x, T* A, const float* cs, const float* sn,
                                int rows, int seq, int nh, int d, int64_t ldg, int64_t ldxr, int64_t ldx,
                                int64_t lda, float eps_rope, float eps_lin) {
    const int hd = d / 2, cph = hd / W;
    const int64_t total = (int64_t)rows * nh * cph;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cph) * W;
        const int64_t rh = i / cph;
        const int h = (int)(rh % nh);
        const int64_t r = rh / nh;
        const int pos = (int)(r % seq);
        const int64_t ho = (int64_t)h * d;
        Chunk<T, W> g1, g2, r1, r2, x1, x2, o1, o2;
        g1.load(Gr + r * ldg + ho + c);
        g2.load(Gr + r * ldg + ho + c + hd);
        if (eps_rope != 0.f) { r1.load(xr + r * ldxr + ho + c); r2.load(xr + r * ldxr + ho + c + hd); }
        if (eps_lin != 0.f) { x1.load(x + r * ldx + ho + c); x2.load(x + r * ldx + ho + c + hd); }
        float c1[W], c2[W], s1[W], s2[W];
        load_tab<W>(c1, cs + (int64_t)pos * d + c);
        load_tab<W>(c2, cs + (int64_t)pos * d + c + hd);
        load_tab<W>(s1, sn + (int64_t)pos * d + c);
        load_tab<W>(s2, sn + (int64_t)pos * d + c + hd);
#pragma unroll
        for (int k = 0; k < W; ++k) {
            float p1 = g1.v[k], p2 = g2.v[k];
            if (eps_rope != 0.f) { p1 *= eps_ratio(r1.v[k], 1.f, eps_rope); p2 *= eps_ratio(r2.v[k], 1.f, eps_rope); }
            float a1 = p1 * c1[k] + p2 * s2[k];
            float a2 = p2 * c2[k] - p1 * s1[k];
            if (eps_lin != 0.f) { a1 *= eps_ratio(x1.v[k], 1.f, eps_lin); a2 *= eps_ratio(x2.v[k], 1.f, eps_lin); }
            o1.v[k] = a1;
            o2.v[k] = a2;
        }
        o1.store(A + r * lda + ho + c);
        o2.store(A + r * lda + ho + c + hd);
    }
}

// ---- transpose through a padded LDS tile -----------------------------------------------------------
template <typename T>
__global__ void transpose_kernel(const T* in, T* out, int rows, int cols, int64_t ld_in, int64_t ld_out,
                                 int64_t s_in, int64_t s_out) {
    __shared__ T tile[64][65];
    in += blockIdx.z * s_in;
    out += blockIdx.z * s_out;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;          // 256 threads: 4 rows per pass
    for (int rr = ty; rr < 64; rr += 4) {
        const int r = r0 + rr, c = c0 + tx;
        if (r < rows && c < cols) tile[rr][tx] = in[(int64_t)r * ld_in + c];
    }
    __syncthreads();
    for (int cc = ty; cc < 64; cc += 4) {
        const int c = c0 + cc, r = r0 + tx;
        if (r < rows && c < cols) out[(int64_t)c * ld_out + r] = tile[tx][cc];
    }
}

}  // namespace

#define DISPATCH_T(dtype, ...)                                              \
    if (dtype == LRP_F32) { typedef float T; __VA_ARGS__ }                  \
    else if (dtype == LRP_BF16) { typedef bf16_t T; __VA_ARGS__ }           \
    else return LRP_EINVAL;

extern "C" int lrp_eps_scale(const void* g, const void* z, void* out, int64_t n, float c, float eps,
                             int mode, int dtype, void* stream) {
    if (!g || !z || !out || n < 0) return LRP_EINVAL;
    if (n == 0) return LRP_OK;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype, {
        constexpr int EPC = 16 / sizeof(T);
        const T *pg = (const T*)g, *pz = (const T*)z; T* po = (T*)out;
        if (al16(g) && al16(z) && al16(out)) {
            const int64_t nb = (n / EPC) * EPC;
            if (nb) hipLaunchKernelGGL((eps_scale_kernel<T, EPC>), dim3(grid_for(nb / EPC)), dim3(ENT), 0, st, pg, pz, po, nb, c, eps, mode);
            if (n > nb) hipLaunchKernelGGL((eps_scale_kernel<T, 1>), dim3(1), dim3(64), 0, st, pg + nb, pz + nb, po + nb, n - nb, c, eps, mode);
        } else {
            hipLaunchKernelGGL((eps_scale_kernel<T, 1>), dim3(grid_for(n)), dim3(ENT), 0, st, pg, pz, po, n, c, eps, mode);
        }
    })
    return lrp_check_launch();
}

extern "C" int lrp_eps_scale2d(const void* g, const void* z, void* out, int rows, int cols, int64_t ldg,
                               int64_t ldz, int64_t ldo, float c, float eps, int mode, int dtype, void* stream) {
    if (!g || !z || !out || rows < 0 || cols < 0) return LRP_EINVAL;
    if (rows == 0 || cols == 0) return LRP_OK;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype, {
        constexpr int EPC = 16 / sizeof(T);
        const bool v = al16(g) && al16(z) && al16(out) && (cols % EPC == 0) && (ldg % EPC == 0) && (ldz % EPC == 0) && (ldo % EPC == 0);
        if (v) hipLaunchKernelGGL((eps_scale2d_kernel<T, EPC>), dim3(grid_for((int64_t)rows * cols / EPC)), dim3(ENT), 0, st, (const T*)g, (const T*)z, (T*)out, rows, cols, ldg, ldz, ldo, c, eps, mode);
        else hipLaunchKernelGGL((eps_scale2d_kernel<T, 1>), dim3(grid_for((int64_t)rows * cols)), dim3(ENT), 0, st, (const T*)g, (const T*)z, (T*)out, rows, cols, ldg, ldz, ldo, c, eps, mode);
    })
    return lrp_check_launch();
}

extern "C" int lrp_mul(const void* a, const void* b, void* out, int64_t n, int dtype, void* stream) {
    if (!a || !b || !out || n < 0) return LRP_EINVAL;
    if (n == 0) return LRP_OK;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype, {
        constexpr int EPC = 16 / sizeof(T);
        const T *pa = (const T*)a, *pb = (const T*)b; T* po = (T*)out;
        if (al16(a) && al16(b) && al16(out)) {
            const int64_t nb = (n / EPC) * EPC;
            if (nb) hipLaunchKernelGGL((mul_kernel<T, EPC>), dim3(grid_for(nb / EPC)), dim3(ENT), 0, st, pa, pb, po, nb);
            if (n > nb) hipLaunchKernelGGL((mul_kernel<T, 1>), dim3(1), dim3(64), 0, st, pa + nb, pb + nb, po + nb, n - nb);
        } else {
            hipLaunchKernelGGL((mul_kernel<T, 1>), dim3(grid_for(n)), dim3(ENT), 0, st, pa, pb, po, n);
        }
    })
    return lrp_check_launch();
}

extern "C" int lrp_add_bcast(const void* x, const void* y, void* out, int M, int H, int period, int dtype, void* stream) {
    if (!x || !y || !out || M < 0 || H < 0 || period <= 0) return LRP_EINVAL;
    if (M == 0 || H == 0) return LRP_OK;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype, {
        constexpr int EPC = 16 / sizeof(T);
        if (al16(x) && al16(y) && al16(out) && (H % EPC == 0))
            hipLaunchKernelGGL((add_bcast_kernel<T, EPC>), dim3(grid_for((int64_t)M * H / EPC)), dim3(ENT), 0, st, (const T*)x, (const T*)y, (T*)out, M, H, period);
        else
            hipLaunchKernelGGL((add_bcast_kernel<T, 1>), dim3(grid_for((int64_t)M * H)), dim3(ENT), 0, st, (const T*)x, (const T*)y, (T*)out, M, H, period);
    })
    return lrp_check_launch();
}

extern "C" int lrp_add2_rule_bwd(const void* a, const void* b, const void* R, void* Ra, void* Rb,
                                 int64_t n, float eps, int dtype, void* stream) {
    if (!a || !b || !R || !Ra || n < 0) return LRP_EINVAL;
    if (n == 0) return LRP_OK;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype, {
        constexpr int EPC = 16 / sizeof(T);
        const T *pa = (const T*)a, *pb = (const T*)b, *pr = (const T*)R; T *oa = (T*)Ra, *ob = (T*)Rb;
        if (al16(a) && al16(b) && al16(R) && al16(Ra) && al16(Rb)) {
            const int64_t nb = (n / EPC) * EPC;
            if (nb) hipLaunchKernelGGL((add2_rule_kernel<T, EPC>), dim3(grid_for(nb / EPC)), dim3(ENT), 0, st, pa, pb, pr, oa, ob, nb, eps);
            if (n > nb) hipLaunchKernelGGL((add2_rule_kernel<T, 1>), dim3(1), dim3(64), 0, st, pa + nb, pb + nb, pr + nb, oa + nb, ob ? ob + nb : ob, n - nb, eps);
        } else {
            hipLaunchKernelGGL((add2_rule_kernel<T, 1>), dim3(grid_for(n)), dim3(ENT), 0, st, pa, pb, pr, oa, ob, n, eps);
        }
    })
    return lrp_check_launch();
}

extern "C" int lrp_cast(const void* in, void* out, int64_t n, int in_dtype, int out_dtype, void* stream) {
    if (!in || !out || n < 0) return LRP_EINVAL;
    if (n == 0) return LRP_OK;
    hipStream_t st = (hipStream_t)stream;
    dim3 g(grid_for(n)), b(ENT);
    if (in_dtype == LRP_F32 && out_dtype == LRP_BF16)
        hipLaunchKernelGGL((cast_kernel<float, bf16_t>), g, b, 0, st, (const float*)in, (bf16_t*)out, n);
    else if (in_dtype == LRP_BF16 && out_dtype == LRP_F32)
        hipLaunchKernelGGL((cast_kernel<bf16_t, float>), g, b, 0, st, (const bf16_t*)in, (float*)out, n);
    else if (in_dtype == LRP_F32 && out_dtype == LRP_F32)
        hipLaunchKernelGGL((cast_kernel<float, float>), g, b, 0, st, (const float*)in, (float*)out, n);
    else if (in_dtype == LRP_BF16 && out_dtype == LRP_BF16)
        hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), g, b, 0, st, (const bf16_t*)in, (bf16_t*)out, n);
    else return LRP_EINVAL;
    return lrp_check_launch();
}

extern "C" int lrp_act_fwd(const void* x, void* y, int64_t n, int act, int dtype, void* stream) {
    if (!x || !y || n < 0 || act < 0 || act > 3) return LRP_EINVAL;
    if (n == 0) return LRP_OK;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype, {
        constexpr int EPC = 16 / sizeof(T);
        const T* px = (const T*)x; T* py = (T*)y;
        if (al16(x) && al16(y)) {
            const int64_t nb = (n / EPC) * EPC;
            if (nb) hipLaunchKernelGGL((act_fwd_kernel<T, EPC>), dim3(grid_for(nb / EPC)), dim3(ENT), 0, st, px, py, nb, act);
            if (n > nb) hipLaunchKernelGGL((act_fwd_kernel<T, 1>), dim3(1), dim3(64), 0, st, px + nb, py + nb, n - nb, act);
        } else {
            hipLaunchKernelGGL((act_fwd_kernel<T, 1>), dim3(grid_for(n)), dim3(ENT), 0, st, px, py, n, act);
        }
    })
    return lrp_check_launch();
}

extern "C" int lrp_act_bwd(const void* Gy, const void* x, void* Gx, int64_t n, float eps_g, int act,
                           int dtype, void* stream) {
    if (!Gy || !x || !Gx || n < 0 || act < 0 || act > 3) return LRP_EINVAL;
    if (n == 0) return LRP_OK;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype, {
        constexpr int EPC = 16 / sizeof(T);
        const T *pg = (const T*)Gy, *px = (const T*)x; T* po = (T*)Gx;
        if (al16(Gy) && al16(x) && al16(Gx)) {
            const int64_t nb = (n / EPC) * EPC;
            if (nb) hipLaunchKernelGGL((act_bwd_kernel<T, EPC>), dim3(grid_for(nb / EPC)), dim3(ENT), 0, st, pg, px, po, nb, eps_g, act);
            if (n > nb) hipLaunchKernelGGL((act_bwd_kernel<T, 1>), dim3(1), dim3(64), 0, st, pg + nb, px + nb, po + nb, n - nb, eps_g, act);
        } else {
            hipLaunchKernelGGL((act_bwd_kernel<T, 1>), dim3(grid_for(n)), dim3(ENT), 0, st, pg, px, po, n, eps_g, act);
        }
    })
    return lrp_check_launch();
}

extern "C" int lrp_act_grad(const void* Gy, const void* x, void* Gx, int64_t n, int act, int dtype, void* stream) {
    if (!Gy || !x || !Gx || n < 0 || act < 0 || act > 3) return LRP_EINVAL;
    if (n == 0) return LRP_OK;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype, {
        constexpr int EPC = 16 / sizeof(T);
        const T *pg = (const T*)Gy, *px = (const T*)x; T* po = (T*)Gx;
        if (al16(Gy) && al16(x) && al16(Gx)) {
            const int64_t nb = (n / EPC) * EPC;
            if (nb) hipLaunchKernelGGL((act_grad_kernel<T, EPC>), dim3(grid_for(nb / EPC)), dim3(ENT), 0, st, pg, px, po, nb, act);
            if (n > nb) hipLaunchKernelGGL((act_grad_kernel<T, 1>), dim3(1), dim3(64), 0, st, pg + nb, px + nb, po + nb, n - nb, act);
        } else {
            hipLaunchKernelGGL((act_grad_kernel<T, 1>), dim3(grid_for(n)), dim3(ENT), 0, st, pg, px, po, n, act);
        }
    })
    return lrp_check_launch();
}

static inline bool ld_ok(int64_t ld, int epc) { return (ld % epc) == 0; }

static int gated_fwd_launch(const void* g, const void* u, void* m, int M, int I, int64_t ldg, int64_t ldu, int64_t ldm, int act, int il,
                            int dtype, hipStream_t st) {
    if (!g || !u || !m || M < 0 || I < 0 || act < 0 || act > 3 || (il && (I % il))) return LRP_EINVAL;
    if (M == 0 || I == 0) return LRP_OK;
    DISPATCH_T(dtype, {
        constexpr int EPC = 16 / sizeof(T);
        const bool v = al16(g) && al16(u) && al16(m) && (I % EPC == 0) && ld_ok(ldg, EPC) && ld_ok(ldu, EPC) && ld_ok(ldm, EPC) && (il % EPC == 0);
        if (v) hipLaunchKernelGGL((gated_fwd_kernel<T, EPC>), dim3(grid_for((int64_t)M * I / EPC)), dim3(ENT), 0, st, (const T*)g, (const T*)u, (T*)m, M, I, ldg, ldu, ldm, act, il);
        else hipLaunchKernelGGL((gated_fwd_kernel<T, 1>), dim3(grid_for((int64_t)M * I)), dim3(ENT), 0, st, (const T*)g, (const T*)u, (T*)m, M, I, ldg, ldu, ldm, act, il);
    })
    return lrp_check_launch();
}

static int gated_bwd_launch(const void* Gm, const void* g, const void* u, void* Ag, void* Au, int M, int I, int64_t ldgm, int64_t ldg,
                            int64_t ldu, int64_t ldag, int64_t ldau, float eps_g, float eps_lin, int act, int il, int dtype, hipStream_t st) {
    if (!Gm || !g || !u || !Ag || !Au || M < 0 || I < 0 || act < 0 || act > 3 || (il && (I % il))) return LRP_EINVAL;
    if (M == 0 || I == 0) return LRP_OK;
    DISPATCH_T(dtype, {
        constexpr int EPC = 16 / sizeof(T);
        const bool v = al16(Gm) && al16(g) && al16(u) && al16(Ag) && al16(Au) && (I % EPC == 0) && ld_ok(ldgm, EPC) &&
                       ld_ok(ldg, EPC) && ld_ok(ldu, EPC) && ld_ok(ldag, EPC) && ld_ok(ldau, EPC) && (il % EPC == 0);
        if (v) hipLaunchKernelGGL((gated_bwd_kernel<T, EPC>), dim3(grid_for((int64_t)M * I / EPC)), dim3(ENT), 0, st, (const T*)Gm, (const T*)g, (const T*)u, (T*)Ag, (T*)Au, M, I, ldgm, ldg, ldu, ldag, ldau, eps_g, eps_lin, act, il);
        else hipLaunchKernelGGL((gated_bwd_kernel<T, 1>), dim3(grid_for((int64_t)M * I)), dim3(ENT), 0, st, (const T*)Gm, (const T*)g, (const T*)u, (T*)Ag, (T*)Au, M, I, ldgm, ldg, ldu, ldag, ldau, eps_g, eps_lin, act, il);
    })
    return lrp_check_launch();
}

extern "C" int lrp_gated_act_fwd(const void* g, const void* u, void* m, int M, int I, int64_t ldg,
                                 int64_t ldu, int64_t ldm, int act, int dtype, void* stream) {
    return gated_fwd_launch(g, u, m, M, I, ldg, ldu, ldm, act, 0, dtype, (hipStream_t)stream);
}

extern "C" int lrp_gated_act_bwd(const void* Gm, const void* g, const void* u, void* Ag, void* Au,
                                 int M, int I, int64_t ldgm, int64_t ldg, int64_t ldu, int64_t ldag,
                                 int64_t ldau, float eps_g, float eps_lin, int act, int dtype, void* stream) {
    return gated_bwd_launch(Gm, g, u, Ag, Au, M, I, ldgm, ldg, ldu, ldag, ldau, eps_g, eps_lin, act, 0, dtype, (hipStream_t)stream);
}

// the same rules on the INTERLEAVED output of a fused gate/up Linear (lrp_gemm_gated_fwd's layout): gu [M, 2 I], column block b of 64 =
// [gate 32 b .. 32 b + 31 | up 32 b .. 32 b + 31]; Agu in the same layout
extern "C" int lrp_gated_act_fwd_il(const void* gu, void* m, int M, int I, int64_t ldgu, int64_t ldm, int act, int dtype, void* stream) {
    const size_t es = dtype == LRP_F32 ? 4 : 2;
    return gated_fwd_launch(gu, gu ? (const char*)gu + 32 * es : nullptr, m, M, I, ldgu, ldgu, ldm, act, LRP_GATED_IL, dtype, (hipStream_t)stream);
}

extern "C" int lrp_gated_act_bwd_il(const void* Gm, const void* gu, void* Agu, int M, int I, int64_t ldgm, int64_t ldgu, int64_t ldagu,
                                    float eps_g, float eps_lin, int act, int dtype, void* stream) {
    const size_t es = dtype == LRP_F32 ? 4 : 2;
    return gated_bwd_launch(Gm, gu, gu ? (const char*)gu + 32 * es : nullptr, Agu, Agu ? (char*)Agu + 32 * es : nullptr, M, I, ldgm, ldgu, ldgu,
                            ldagu, ldagu, eps_g, eps_lin, act, LRP_GATED_IL, dtype, (hipStream_t)stream);
}

extern "C" int lrp_rope_fwd(const void* x, void* xr, const float* cos_t, const float* sin_t, int rows,
                            int seq, int n_heads, int d, int64_t ldx, int64_t ldxr, int dtype, void* stream) {
    if (!x || !xr || !cos_t || !sin_t || rows < 0 || seq < 1 || n_heads < 1 || d < 2 || (d & 1)) return LRP_EINVAL;
    if (rows == 0) return LRP_OK;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype, {
        constexpr int EPC = 16 / sizeof(T);
        const bool v = al16(x) && al16(xr) && al16(cos_t) && al16(sin_t) && ((d / 2) % EPC == 0) && ld_ok(ldx, EPC) && ld_ok(ldxr, EPC);
        const int64_t work = (int64_t)rows * n_heads * (d / 2);
        if (v) hipLaunchKernelGGL((rope_fwd_kernel<T, EPC>), dim3(grid_for(work / EPC)), dim3(ENT), 0, st, (const T*)x, (T*)xr, cos_t, sin_t, rows, seq, n_heads, d, ldx, ldxr);
        else hipLaunchKernelGGL((rope_fwd_kernel<T, 1>), dim3(grid_for(work)), dim3(ENT), 0, st, (const T*)x, (T*)xr, cos_t, sin_t, rows, seq, n_heads, d, ldx, ldxr);
    })
    return lrp_check_launch();
}

extern "C" int lrp_rope_bwd(const void* Gr, const void* xr, const void* x, void* A, const float* cos_t,
                            const float* sin_t, int rows, int seq, int n_heads, int d, int64_t ldg,
                            int64_t ldxr, int64_t ldx, int64_t lda, float eps_rope, float eps_lin,
                            int dtype, void* stream) {
    if (!Gr || !A || !cos_t || !sin_t || rows < 0 || seq < 1 || n_heads < 1 || d < 2 || (d & 1)) return LRP_EINVAL;
    if ((eps_rope != 0.f && !xr) || (eps_lin != 0.f && !x)) return LRP_EINVAL;
    if (rows == 0) return LRP_OK;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype, {
        constexpr int EPC = 16 / sizeof(T);
        const bool v = al16(Gr) && al16(A) && al16(cos_t) && al16(sin_t) && (!xr || al16(xr)) && (!x || al16(x)) && ((d / 2) % EPC == 0) &&
                       ld_ok(ldg, EPC) && ld_ok(lda, EPC) && ld_ok(ldxr, EPC) && ld_ok(ldx, EPC);
        const int64_t work = (int64_t)rows * n_heads * (d / 2);
        if (v) hipLaunchKernelGGL((rope_bwd_kernel<T, EPC>), dim3(grid_for(work / EPC)), dim3(ENT), 0, st, (const T*)Gr, (const T*)xr, (const T*)x, (T*)A, cos_t, sin_t, rows, seq, n_heads, d, ldg, ldxr, ldx, lda, eps_rope, eps_lin);
        else hipLaunchKernelGGL((rope_bwd_kernel<T, 1>), dim3(grid_for(work)), dim3(ENT), 0, st, (const T*)Gr, (const T*)xr, (const T*)x, (T*)A, cos_t, sin_t, rows, seq, n_heads, d, ldg, ldxr, ldx, lda, eps_rope, eps_lin);
    })
    return lrp_check_launch();
}

extern "C" int lrp_transpose(const void* in, void* out, int rows, int cols, int64_t ld_in, int64_t ld_out,
                             int batch, int64_t s_in, int64_t s_out, int dtype, void* stream) {
    if (!in || !out || rows < 0 || cols < 0 || batch < 1) return LRP_EINVAL;
    if (rows == 0 || cols == 0) return LRP_OK;
    if (batch > 65535 || (rows + 63) / 64 > 65535) return LRP_ESHAPE;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((cols + 63) / 64, (rows + 63) / 64, batch), block(256);
    DISPATCH_T(dtype, {
        hipLaunchKernelGGL((transpose_kernel<T>), grid, block, 0, st, (const T*)in, (T*)out, rows, cols, ld_in, ld_out, s_in, s_out);
    })
    return lrp_check_launch();
}
