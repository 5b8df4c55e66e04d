// rowops.hip -- HBM-bound row kernels with wavefront reductions:
//   K2/K6 RMSNorm identity rule fused with the residual add2 rule, K7 LayerNorm, the explicit
//   softmax rule on materialised rows, K9 read-out, last-token head seed, attention backward prep.
// One workgroup per row (64 threads for short rows, 256 otherwise); 16-byte vector accesses when
// the row is aligned, scalar otherwise.  Reductions: DPP/shuffle inside the wave, LDS across waves.
#include "common.hpp"

namespace {

inline bool al16(const void* p) { return !p || (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline int row_threads(int H, int epc) { return (H / epc <= 64) ? 64 : 256; }

template <typename T, int W> struct RChunk {
    float v[W];
    LRP_DEVICE void load(const T* p) {
        if constexpr (W == 1) v[0] = to_f32(p[0]);
        else { Vec16<T> t = ld16(p);
#pragma unroll
            for (int i = 0; i < W; ++i) v[i] = t.get(i); }
    }
    LRP_DEVICE void store(T* p) const {
        if constexpr (W == 1) p[0] = from_f32<T>(v[0]);
        else { Vec16<T> t;
#pragma unroll
            for (int i = 0; i < W; ++i) t.set(i, v[i]);
            st16(p, t); }
    }
};

// ---------------------------------------------------------------------------------------------------
// forward: hsum = h (+ branch) ; rstd ; y = w' * (hsum * rstd)
template <typename T, int W>
__global__ void add_rmsnorm_fwd_kernel(const T* h, const T* branch, const T* w, T* hsum_out, T* y,
                                       float* rstd, int H, float eps, float w_off) {
    __shared__ float red[4];
    const int64_t row = blockIdx.x;
    const T* ph = h + row * H;
    const T* pb = branch ? branch + row * H : nullptr;
    T* ps = hsum_out ? hsum_out + row * H : nullptr;
    float ss = 0.f;
    for (int c = threadIdx.x * W; c < H; c += blockDim.x * W) {
        RChunk<T, W> a, b;
        a.load(ph + c);
        if (pb) {
            b.load(pb + c);
#pragma unroll
            for (int k = 0; k < W; ++k) a.v[k] = to_f32(from_f32<T>(a.v[k] + b.v[k]));   // residual add in storage dtype
            if (ps) a.store(ps + c);
        } else if (ps) a.store(ps + c);
#pragma unroll
        for (int k = 0; k < W; ++k) ss += a.v[k] * a.v[k];
    }
    ss = block_sum(ss, red);
    const float rs = rsqrtf(ss / (float)H + eps);
    if (threadIdx.x == 0) rstd[row] = rs;
    __syncthreads();   // hsum_out written by this block is re-read below (same threads, same addresses)
    const T* src = ps ? ps : ph;
    for (int c = threadIdx.x * W; c < H; c += blockDim.x * W) {
        RChunk<T, W> a, ww, o;
        if (ps || !pb) a.load(src + c);
        else { RChunk<T, W> b; a.load(ph + c); b.load(pb + c);
#pragma unroll
            for (int k = 0; k < W; ++k) a.v[k] = to_f32(from_f32<T>(a.v[k] + b.v[k])); }
        ww.load(w + c);
#pragma unroll
        for (int k = 0; k < W; ++k) {
            if (w_off == 0.f) o.v[k] = ww.v[k] * to_f32(from_f32<T>(a.v[k] * rs));   // HF Llama: w * (x*rstd).to(dtype)
            else o.v[k] = (a.v[k] * rs) * (w_off + ww.v[k]);                          // Gemma3: fp32 product, then cast
        }
        o.store(y + row * H + c);
    }
}

// backward: Gh = Gres + Gx*w'*rstd ; Gs = Gh*hsum/(hsum+eps_add) ; A = Gs*branch/(branch+eps_lin)
template <typename T, int W>
__global__ void rmsnorm_bwd_add2_kernel(const T* Gres, const T* Gx, const T* w, const float* rstd, const T* hsum,
                                        const T* branch, T* Gs_out, T* A_out, float* rel_out, int H,
                                        float w_off, float eps_add, float eps_lin) {
    __shared__ float red[4];
    const int64_t row = blockIdx.x;
    const float rs = Gx ? rstd[row] : 0.f;
    float rel = 0.f;
    for (int c = threadIdx.x * W; c < H; c += blockDim.x * W) {
        RChunk<T, W> gr, gx, ww, hs, br, os, oa;
        if (Gx) { gx.load(Gx + row * H + c); ww.load(w + c); }
        if (Gres) gr.load(Gres + row * H + c);
        const bool need_h = (rel_out != nullptr) || (branch && eps_add != 0.f);
        if (need_h) hs.load(hsum + row * H + c);
        if (branch && eps_lin != 0.f) br.load(branch + row * H + c);
#pragma unroll
        for (int k = 0; k < W; ++k) {
            float gh = Gx ? gx.v[k] * (ww.v[k] + w_off) * rs : 0.f;
            if (Gres) gh += gr.v[k];
            if (rel_out) rel += hs.v[k] * gh;
            float gs = gh;
            if (branch) {
                gs = gh * ((eps_add == 0.f) ? 1.f : eps_ratio(hs.v[k], 1.f, eps_add));
                oa.v[k] = gs * ((eps_lin == 0.f) ? 1.f : eps_ratio(br.v[k], 1.f, eps_lin));
            }
            os.v[k] = gs;
        }
        os.store(Gs_out + row * H + c);
        if (branch && A_out) oa.store(A_out + row * H + c);
    }
    if (rel_out) {
        rel = block_sum(rel, red);
        if (threadIdx.x == 0) rel_out[row] = rel;
    }
}

// ---- LayerNorm ---------------------------------------------------------------------------------------
template <typename T, int W>
__global__ void layernorm_fwd_kernel(const T* x, const T* w, const T* b, T* y, float* mean, float* rstd, int H, float eps) {
    __shared__ float red[4];
    const int64_t row = blockIdx.x;
    const T* px = x + row * H;
    float s = 0.f;
    for (int c = threadIdx.x * W; c < H; c += blockDim.x * W) {
        RChunk<T, W> a; a.load(px + c);
#pragma unroll
        for (int k = 0; k < W; ++k) s += a.v[k];
    }
    const float mu = block_sum(s, red) / (float)H;
    float v = 0.f;
    for (int c = threadIdx.x * W; c < H; c += blockDim.x * W) {
        RChunk<T, W> a; a.load(px + c);
#pragma unroll
        for (int k = 0; k < W; ++k) { const float dlt = a.v[k] - mu; v += dlt * dlt; }
    }
    const float var = block_sum(v, red) / (float)H;
    const float rs = 1.f / sqrtf(var + eps);
    if (threadIdx.x == 0) { if (mean) mean[row] = mu; rstd[row] = rs; }
    for (int c = threadIdx.x * W; c < H; c += blockDim.x * W) {
        RChunk<T, W> a, ww, bb, o;
        a.load(px + c);
        if (w) ww.load(w + c);
        if (b) bb.load(b + c);
#pragma unroll
        for (int k = 0; k < W; ++k) {
            float t = (a.v[k] - mu) * rs;
            if (w) t *= ww.v[k];
            if (b) t += bb.v[k];
            o.v[k] = t;
        }
        o.store(y + row * H + c);
    }
}
template <typename T, int W>
__global__ void layernorm_bwd_kernel(const T* Gy, const T* y, const T* w, const float* rstd, T* Gx, int H, float eps_y) {
    __shared__ float red[4];
    const int64_t row = blockIdx.x;
    const float rs = rstd[row];
    float s = 0.f;
    for (int c = threadIdx.x * W; c < H; c += blockDim.x * W) {
        RChunk<T, W> g, yy, ww;
        g.load(Gy + row * H + c);
        if (eps_y != 0.f) yy.load(y + row * H + c);
        if (w) ww.load(w + c);
#pragma unroll
        for (int k = 0; k < W; ++k) {
            float u = g.v[k] * rs;
            if (eps_y != 0.f) u *= eps_ratio(yy.v[k], 1.f, eps_y);
            if (w) u *= ww.v[k];
            s += u;
        }
    }
    const float mu = block_sum(s, red) / (float)H;
    for (int c = threadIdx.x * W; c < H; c += blockDim.x * W) {
        RChunk<T, W> g, yy, ww, o;
        g.load(Gy + row * H + c);
        if (eps_y != 0.f) yy.load(y + row * H + c);
        if (w) ww.load(w + c);
#pragma unroll
        for (int k = 0; k < W; ++k) {
            float u = g.v[k] * rs;
            if (eps_y != 0.f) u *= eps_ratio(yy.v[k], 1.f, eps_y);
            if (w) u *= ww.v[k];
            o.v[k] = u - mu;
        }
        o.store(Gx + row * H + c);
    }
}

// full LayerNorm VJP (NO rule: mean and 1/std both differentiated): u = Gy (*) w, xh = (x - mean) rstd,
// Gx = rstd (u - mean_row(u) - xh mean_row(u (*) xh)) -- the image tower of Gemma-3 under the reference's gemma3 map (nothing patched in
// modeling_siglip: ref lxt/efficient/models/gemma3.py:14-19)
template <typename T, int W>
__global__ void layernorm_bwd_plain_kernel(const T* Gy, const T* x, const T* w, const float* mean, const float* rstd, T* Gx, int H) {
    __shared__ float red[4];
    const int64_t row = blockIdx.x;
    const float rs = rstd[row], mu = mean[row];
    float s1 = 0.f, s2 = 0.f;
    for (int c = threadIdx.x * W; c < H; c += blockDim.x * W) {
        RChunk<T, W> g, xx, ww;
        g.load(Gy + row * H + c);
        xx.load(x + row * H + c);
        if (w) ww.load(w + c);
#pragma unroll
        for (int k = 0; k < W; ++k) {
            const float u = w ? g.v[k] * ww.v[k] : g.v[k];
            s1 += u;
            s2 += u * ((xx.v[k] - mu) * rs);
        }
    }
    const float m1 = block_sum(s1, red) / (float)H;
    __syncthreads();
    const float m2 = block_sum(s2, red) / (float)H;
    for (int c = threadIdx.x * W; c < H; c += blockDim.x * W) {
        RChunk<T, W> g, xx, ww, o;
        g.load(Gy + row * H + c);
        xx.load(x + row * H + c);
        if (w) ww.load(w + c);
#pragma unroll
        for (int k = 0; k < W; ++k) {
            const float u = w ? g.v[k] * ww.v[k] : g.v[k];
            o.v[k] = rs * (u - m1 - (xx.v[k] - mu) * rs * m2);
        }
        o.store(Gx + row * H + c);
    }
}

// ---- explicit softmax rule on materialised rows --------------------------------------------------------
template <typename T>
__global__ void softmax_fwd_kernel(const T* x, T* p, int n, float inv_t) {
    __shared__ float red[4];
    const int64_t row = blockIdx.x;
    const T* px = x + row * n;
    float m = -INFINITY;
    for (int c = threadIdx.x; c < n; c += blockDim.x) m = fmaxf(m, to_f32(px[c]) * inv_t);
    m = block_max(m, red);
    float s = 0.f;
    for (int c = threadIdx.x; c < n; c += blockDim.x) s += __expf(to_f32(px[c]) * inv_t - m);
    s = block_sum(s, red);
    const float inv = 1.f / s;
    for (int c = threadIdx.x; c < n; c += blockDim.x) p[row * n + c] = from_f32<T>(__expf(to_f32(px[c]) * inv_t - m) * inv);
}
template <typename T>
__global__ void softmax_rule_bwd_kernel(const T* x, const T* p, const T* Rp, T* Rx, int n, float inv_t) {
    __shared__ float red[4];
    const int64_t row = blockIdx.x;
    float s = 0.f;
    for (int c = threadIdx.x; c < n; c += blockDim.x) s += to_f32(Rp[row * n + c]);
    s = block_sum(s, red);
    for (int c = threadIdx.x; c < n; c += blockDim.x) {
        float xv = to_f32(x[row * n + c]) * inv_t;
        if (isinf(xv) && xv < 0.f) xv = 0.f;                       // -inf (mask) -> 0, as the reference
        Rx[row * n + c] = from_f32<T>(xv * (to_f32(Rp[row * n + c]) - to_f32(p[row * n + c]) * s));
    }
}

// ---- read-out: R_tok[row] = sum_h emb*G -----------------------------------------------------------------
template <typename T, int W>
__global__ void readout_kernel(const T* e, const T* g, float* out, int H) {
    __shared__ float red[4];
    const int64_t row = blockIdx.x;
    float s = 0.f;
    for (int c = threadIdx.x * W; c < H; c += blockDim.x * W) {
        RChunk<T, W> a, b;
        a.load(e + row * H + c);
        b.load(g + row * H + c);
#pragma unroll
        for (int k = 0; k < W; ++k) s += a.v[k] * b.v[k];
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[row] = s;
}

// ---- last-token head: Gh_last[b,:] = z/(z+eps) * W_lm[idx[b],:] * w' * rstd[b] ------------------------------
template <typename T>
__global__ void head_seed_kernel(const T* Wlm, const float* logits, const int* idx, const T* wn, const float* rstd,
                                 T* out, int H, int64_t ldl, float w_off, float eps) {
    const int b = blockIdx.x;
    const int i = idx[b];
    const float z = logits[(int64_t)b * ldl + i];
    const float f = eps_ratio(z, 1.f, eps) * rstd[b];
    for (int c = threadIdx.x; c < H; c += blockDim.x)
        out[(int64_t)b * H + c] = from_f32<T>(f * to_f32(Wlm[(int64_t)i * H + c]) * (to_f32(wn[c]) + w_off));
}

// one workgroup of 1024 threads per row; 16-byte loads, four independent ones in flight per thread (the one-dword-at-a-time loop of
// round 2 took 150-440 us on a 128 k ... 262 k vocabulary: latency-bound).  Ties resolve to the lowest index.
__global__ __launch_bounds__(1024) void argmax_rows_kernel(const float* logits, int* idx, float* val, int V, int64_t ld) {
    __shared__ float sv[16];
    __shared__ int si[16];
    const float* p = logits + (int64_t)blockIdx.x * ld;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    auto take = [&](float v, int c) {
        if (v > bv || (v == bv && c < bi)) { bv = v; bi = c; }
    };
    const bool vec = ((reinterpret_cast<uintptr_t>(p) & 15) == 0);
    const int nv = vec ? V / 4 : 0;
    int q = threadIdx.x;
    for (; q + 3 * 1024 < nv; q += 4 * 1024) {
        f32x4 x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = *reinterpret_cast<const f32x4*>(p + 4 * (q + u * 1024));
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) take(x[u][e], 4 * (q + u * 1024) + e);
    }
    for (; q < nv; q += 1024) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(p + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) take(x[e], 4 * q + e);
    }
    for (int c = 4 * nv + threadIdx.x; c < V; c += 1024) take(p[c], c);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sv[w] = bv; si[w] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 16; ++k)
            if (sv[k] > bv || (sv[k] == bv && si[k] < bi)) { bv = sv[k]; bi = si[k]; }
        idx[blockIdx.x] = bi;
        if (val) val[blockIdx.x] = bv;
    }
}

// ---- attention backward prep: Gho = f*Go*o/(o+eps) ; D[b,h,s] = sum_d Gho*o ------------------------------------
// one lane group of d/W lanes per (row, head); W = 16 B chunk
template <typename T, int W>
__global__ void attn_bwd_prep_kernel(const T* Go, const T* o, T* Gho, float* D, int B, int S, int Hq, int d,
                                     int64_t ldgo, int64_t ldo, int64_t ldgho, float eps_pv, float factor) {
    const int lpw = d / W;                                   // working lanes per (row, head) group
    int lpg = 1;                                             // lanes per group: the next power of two <= 64 (d = 96: 12 of 16 lanes work)
    while (lpg < lpw) lpg <<= 1;
    const int gpb = blockDim.x / lpg;
    const int64_t ngroups = (int64_t)B * S * Hq;
    const int lg = threadIdx.x % lpg;
    const bool on = lg < lpw;
    // all lanes of a group share gidx, so a whole group leaves the loop together and the
    // xor-shuffles (offsets < lpg) never cross into a retired group
    for (int64_t gidx = (int64_t)blockIdx.x * gpb + threadIdx.x / lpg; gidx < ngroups; gidx += (int64_t)gridDim.x * gpb) {
        const int h = (int)(gidx % Hq);
        const int64_t row = gidx / Hq;
        RChunk<T, W> g, oo, r;
        float s = 0.f;
        if (on) {
            g.load(Go + row * ldgo + (int64_t)h * d + lg * W);
            oo.load(o + row * ldo + (int64_t)h * d + lg * W);
#pragma unroll
            for (int k = 0; k < W; ++k) {
                // D is taken from the ROUNDED Gho so that sum_j P_ij dP_ij == D_i holds for what the
                // backward kernels actually multiply with
                const float t = to_f32(from_f32<T>(factor * g.v[k] * eps_ratio(oo.v[k], 1.f, eps_pv)));
                r.v[k] = t;
                s += t * oo.v[k];
            }
        }
        for (int off = lpg >> 1; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (on) r.store(Gho + row * ldgho + (int64_t)h * d + lg * W);
        if (lg == 0) {
            const int64_t b = row / S, sidx = row % S;
            D[(b * Hq + h) * S + sidx] = s;
        }
    }
}

// ---- per-head RMSNorm (Gemma-3 q_norm / k_norm: head_dim-wide rows inside a fused [rows, heads*d (+ ...)] projection output) --------------
// one lane group of d/W lanes per (row, head), 16-byte chunk per lane (as attn_bwd_prep): strided in, strided out, no contiguous copies.
// forward: y = (w + w_off) (*) x * rstd, rstd = rsqrt(mean_d(x^2) + eps) (detached in the backward: ref lxt/efficient/models/gemma3.py:11-12)
// backward: out = G (*) (w + w_off) * rstd
template <typename T, int W>
__global__ void head_rmsnorm_fwd_kernel(const T* x, const T* w, T* y, float* rstd, int64_t rows, int heads, int d, int64_t ldx, int64_t ldy,
                                        float eps, float w_off) {
    const int lpg = d / W, gpb = blockDim.x / lpg, lg = threadIdx.x % lpg;
    const int64_t ngroups = rows * heads;
    for (int64_t gidx = (int64_t)blockIdx.x * gpb + threadIdx.x / lpg; gidx < ngroups; gidx += (int64_t)gridDim.x * gpb) {
        const int h = (int)(gidx % heads);
        const int64_t row = gidx / heads;
        RChunk<T, W> a, ww, o;
        a.load(x + row * ldx + (int64_t)h * d + lg * W);
        ww.load(w + lg * W);
        float ss = 0.f;
#pragma unroll
        for (int k = 0; k < W; ++k) ss += a.v[k] * a.v[k];
        for (int off = lpg >> 1; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
        const float rs = rsqrtf(ss / (float)d + eps);
#pragma unroll
        for (int k = 0; k < W; ++k) {
            if (w_off == 0.f) o.v[k] = ww.v[k] * to_f32(from_f32<T>(a.v[k] * rs));
            else o.v[k] = (a.v[k] * rs) * (w_off + ww.v[k]);
        }
        o.store(y + row * ldy + (int64_t)h * d + lg * W);
        if (lg == 0) rstd[gidx] = rs;
    }
}
template <typename T, int W>
__global__ void head_rmsnorm_bwd_kernel(const T* G, const T* w, const float* rstd, T* out, int64_t rows, int heads, int d, int64_t ldg,
                                        int64_t ldo, float w_off) {
    const int lpg = d / W, gpb = blockDim.x / lpg, lg = threadIdx.x % lpg;
    const int64_t ngroups = rows * heads;
    for (int64_t gidx = (int64_t)blockIdx.x * gpb + threadIdx.x / lpg; gidx < ngroups; gidx += (int64_t)gridDim.x * gpb) {
        const int h = (int)(gidx % heads);
        const int64_t row = gidx / heads;
        const float rs = rstd[gidx];
        RChunk<T, W> g, ww, o;
        g.load(G + row * ldg + (int64_t)h * d + lg * W);
        ww.load(w + lg * W);
#pragma unroll
        for (int k = 0; k < W; ++k) o.v[k] = g.v[k] * (ww.v[k] + w_off) * rs;
        o.store(out + row * ldo + (int64_t)h * d + lg * W);
    }
}

}  // namespace

#define DISPATCH_T(dtype, ...)                                              \
    if (dtype == LRP_F32) { typedef float T; __VA_ARGS__ }                  \
    else if (dtype == LRP_BF16) { typedef bf16_t T; __VA_ARGS__ }           \
    else return LRP_EINVAL;

extern "C" int lrp_add_rmsnorm_fwd(const void* h, const void* branch, const void* w, void* hsum_out, void* y,
                                   float* rstd, int M, int H, float eps, float w_offset, int dtype, void* stream) {
    if (!h || !w || !y || !rstd || M < 0 || H < 1) return LRP_EINVAL;
    if (M == 0) return LRP_OK;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype, {
        constexpr int EPC = 16 / sizeof(T);
        const bool v = (H % EPC == 0) && al16(h) && al16(branch) && al16(w) && al16(hsum_out) && al16(y);
        if (v) hipLaunchKernelGGL((add_rmsnorm_fwd_kernel<T, EPC>), dim3(M), dim3(row_threads(H, EPC)), 0, st, (const T*)h, (const T*)branch, (const T*)w, (T*)hsum_out, (T*)y, rstd, H, eps, w_offset);
        else hipLaunchKernelGGL((add_rmsnorm_fwd_kernel<T, 1>), dim3(M), dim3(row_threads(H, 1)), 0, st, (const T*)h, (const T*)branch, (const T*)w, (T*)hsum_out, (T*)y, rstd, H, eps, w_offset);
    })
    return lrp_check_launch();
}

// K1n: rstd[m] = rsqrt(sum_p ssq[p][m] / H + eps) from the per-64-column partial sums of squares lrp_gemm_res_ssq's epilogue wrote.
// 256 threads per 64 rows: wave q sums the partials p = q, q + 4, ... of row (lane) -- consecutive lanes read consecutive rows of a partial
// (coalesced), every wave keeps parts / 4 independent loads in flight (one thread per row walking 64 strided partials: 24 us for 2 MB) -- and the
// four sums meet in LDS in wave order: the same summation tree for every row, whatever M (deterministic; batch-size independent).
__global__ __launch_bounds__(256) void rms_rstd_kernel(const float* __restrict__ ssq, int parts, int64_t ldssq, int M, float inv_h, float eps,
                                                       float* __restrict__ rstd) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int m = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (m < M)
        for (int p = q; p < parts; p += 4) s += ssq[(int64_t)p * ldssq + m];
    red[q][lane] = s;
    __syncthreads();
    if (q == 0 && m < M) rstd[m] = rsqrtf(((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane])) * inv_h + eps);
}

extern "C" int lrp_rms_rstd(const float* ssq, int parts, int64_t ldssq, int M, int H, float eps, float* rstd, void* stream) {
    if (!ssq || !rstd || parts < 1 || M < 0 || H < 1 || ldssq < M) return LRP_EINVAL;
    if (M == 0) return LRP_OK;
    hipLaunchKernelGGL(rms_rstd_kernel, dim3((M + 63) / 64), dim3(256), 0, (hipStream_t)stream, ssq, parts, ldssq, M, 1.f / (float)H, eps, rstd);
    return lrp_check_launch();
}

extern "C" int lrp_rmsnorm_bwd_add2(const void* Gres, const void* Gx, const void* w, const float* rstd,
                                    const void* hsum, const void* branch, void* Gs_out, void* A_out,
                                    float* rel_out, int M, int H, float w_offset, float eps_add, float eps_lin,
                                    int dtype, void* stream) {
    if ((!Gx && !Gres) || (Gx && (!w || !rstd)) || !Gs_out || M < 0 || H < 1) return LRP_EINVAL;
    if ((rel_out || (branch && eps_add != 0.f)) && !hsum) return LRP_EINVAL;
    if (branch && !A_out) return LRP_EINVAL;
    if (M == 0) return LRP_OK;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype, {
        constexpr int EPC = 16 / sizeof(T);
        const bool v = (H % EPC == 0) && al16(Gres) && al16(Gx) && al16(w) && al16(hsum) && al16(branch) && al16(Gs_out) && al16(A_out);
        if (v) hipLaunchKernelGGL((rmsnorm_bwd_add2_kernel<T, EPC>), dim3(M), dim3(row_threads(H, EPC)), 0, st, (const T*)Gres, (const T*)Gx, (const T*)w, rstd, (const T*)hsum, (const T*)branch, (T*)Gs_out, (T*)A_out, rel_out, H, w_offset, eps_add, eps_lin);
        else hipLaunchKernelGGL((rmsnorm_bwd_add2_kernel<T, 1>), dim3(M), dim3(row_threads(H, 1)), 0, st, (const T*)Gres, (const T*)Gx, (const T*)w, rstd, (const T*)hsum, (const T*)branch, (T*)Gs_out, (T*)A_out, rel_out, H, w_offset, eps_add, eps_lin);
    })
    return lrp_check_launch();
}

extern "C" int lrp_layernorm_fwd(const void* x, const void* w, const void* b, void* y, float* mean, float* rstd,
                                 int M, int H, float eps, int dtype, void* stream) {
    if (!x || !y || !rstd || M < 0 || H < 1) return LRP_EINVAL;
    if (M == 0) return LRP_OK;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype, {
        constexpr int EPC = 16 / sizeof(T);
        const bool v = (H % EPC == 0) && al16(x) && al16(w) && al16(b) && al16(y);
        if (v) hipLaunchKernelGGL((layernorm_fwd_kernel<T, EPC>), dim3(M), dim3(row_threads(H, EPC)), 0, st, (const T*)x, (const T*)w, (const T*)b, (T*)y, mean, rstd, H, eps);
        else hipLaunchKernelGGL((layernorm_fwd_kernel<T, 1>), dim3(M), dim3(row_threads(H, 1)), 0, st, (const T*)x, (const T*)w, (const T*)b, (T*)y, mean, rstd, H, eps);
    })
    return lrp_check_launch();
}

extern "C" int lrp_layernorm_bwd(const void* Gy, const void* y, const void* w, const float* rstd, void* Gx,
                                 int M, int H, float eps_y, int dtype, void* stream) {
    if (!Gy || !rstd || !Gx || M < 0 || H < 1 || (eps_y != 0.f && !y)) return LRP_EINVAL;
    if (M == 0) return LRP_OK;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype, {
        constexpr int EPC = 16 / sizeof(T);
        const bool v = (H % EPC == 0) && al16(Gy) && al16(y) && al16(w) && al16(Gx);
        if (v) hipLaunchKernelGGL((layernorm_bwd_kernel<T, EPC>), dim3(M), dim3(row_threads(H, EPC)), 0, st, (const T*)Gy, (const T*)y, (const T*)w, rstd, (T*)Gx, H, eps_y);
        else hipLaunchKernelGGL((layernorm_bwd_kernel<T, 1>), dim3(M), dim3(row_threads(H, 1)), 0, st, (const T*)Gy, (const T*)y, (const T*)w, rstd, (T*)Gx, H, eps_y);
    })
    return lrp_check_launch();
}

extern "C" int lrp_layernorm_bwd_plain(const void* Gy, const void* x, const void* w, const float* mean, const float* rstd, void* Gx,
                                       int M, int H, int dtype, void* stream) {
    if (!Gy || !x || !mean || !rstd || !Gx || M < 0 || H < 1) return LRP_EINVAL;
    if (M == 0) return LRP_OK;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype, {
        constexpr int EPC = 16 / sizeof(T);
        const bool v = (H % EPC == 0) && al16(Gy) && al16(x) && al16(w) && al16(Gx);
        if (v) hipLaunchKernelGGL((layernorm_bwd_plain_kernel<T, EPC>), dim3(M), dim3(row_threads(H, EPC)), 0, st, (const T*)Gy, (const T*)x, (const T*)w, mean, rstd, (T*)Gx, H);
        else hipLaunchKernelGGL((layernorm_bwd_plain_kernel<T, 1>), dim3(M), dim3(row_threads(H, 1)), 0, st, (const T*)Gy, (const T*)x, (const T*)w, mean, rstd, (T*)Gx, H);
    })
    return lrp_check_launch();
}

extern "C" int lrp_softmax_fwd(const void* x, void* p, int64_t rows, int n, float inv_temp, int dtype, void* stream) {
    if (!x || !p || rows < 0 || n < 1) return LRP_EINVAL;
    if (rows == 0) return LRP_OK;
    if (rows > 0x7fffffff) return LRP_ESHAPE;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype, {
        hipLaunchKernelGGL((softmax_fwd_kernel<T>), dim3((unsigned)rows), dim3(n <= 64 ? 64 : 256), 0, st, (const T*)x, (T*)p, n, inv_temp);
    })
    return lrp_check_launch();
}

extern "C" int lrp_softmax_rule_bwd(const void* x, const void* p, const void* Rp, void* Rx, int64_t rows, int n,
                                    float inv_temp, int dtype, void* stream) {
    if (!x || !p || !Rp || !Rx || rows < 0 || n < 1) return LRP_EINVAL;
    if (rows == 0) return LRP_OK;
    if (rows > 0x7fffffff) return LRP_ESHAPE;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype, {
        hipLaunchKernelGGL((softmax_rule_bwd_kernel<T>), dim3((unsigned)rows), dim3(n <= 64 ? 64 : 256), 0, st, (const T*)x, (const T*)p, (const T*)Rp, (T*)Rx, n, inv_temp);
    })
    return lrp_check_launch();
}

extern "C" int lrp_readout(const void* emb, const void* G, float* R_tok, int M, int H, int dtype, void* stream) {
    if (!emb || !G || !R_tok || M < 0 || H < 1) return LRP_EINVAL;
    if (M == 0) return LRP_OK;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype, {
        constexpr int EPC = 16 / sizeof(T);
        const bool v = (H % EPC == 0) && al16(emb) && al16(G);
        if (v) hipLaunchKernelGGL((readout_kernel<T, EPC>), dim3(M), dim3(row_threads(H, EPC)), 0, st, (const T*)emb, (const T*)G, R_tok, H);
        else hipLaunchKernelGGL((readout_kernel<T, 1>), dim3(M), dim3(row_threads(H, 1)), 0, st, (const T*)emb, (const T*)G, R_tok, H);
    })
    return lrp_check_launch();
}

extern "C" int lrp_head_seed(const void* W_lm, const float* logits, const int* idx, const void* w_norm,
                             const float* rstd_last, void* Gh_last, int B, int V, int H, int64_t ld_logits,
                             float w_offset, float eps_lin, int dtype, void* stream) {
    if (!W_lm || !logits || !idx || !w_norm || !rstd_last || !Gh_last || B < 0 || V < 1 || H < 1) return LRP_EINVAL;
    if (B == 0) return LRP_OK;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype, {
        hipLaunchKernelGGL((head_seed_kernel<T>), dim3(B), dim3(256), 0, st, (const T*)W_lm, logits, idx, (const T*)w_norm, rstd_last, (T*)Gh_last, H, ld_logits, w_offset, eps_lin);
    })
    return lrp_check_launch();
}

extern "C" int lrp_argmax_rows(const float* logits, int* idx, float* val, int B, int V, int64_t ld, void* stream) {
    if (!logits || !idx || B < 0 || V < 1) return LRP_EINVAL;
    if (B == 0) return LRP_OK;
    hipLaunchKernelGGL(argmax_rows_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, logits, idx, val, V, ld);
    return lrp_check_launch();
}

extern "C" int lrp_attn_bwd_prep(const void* Go, const void* o, void* Gho, float* D, int B, int S, int Hq, int d,
                                 int64_t ldgo, int64_t ldo, int64_t ldgho, float eps_pv, float factor,
                                 int dtype, void* stream) {
    if (!Go || !o || !Gho || !D || B < 0 || S < 0 || Hq < 1 || d < 1) return LRP_EINVAL;
    if (B == 0 || S == 0) return LRP_OK;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype, {
        constexpr int EPC = 16 / sizeof(T);
        const int lpw = d / EPC;
        if (d % EPC || lpw < 1 || lpw > 64) return LRP_ESHAPE;
        int lpg = 1;                                         // lane group = the next power of two (d = 96 in bf16: 12 working lanes of 16)
        while (lpg < lpw) lpg <<= 1;
        if (!al16(Go) || !al16(o) || !al16(Gho) || (ldgo % EPC) || (ldo % EPC) || (ldgho % EPC)) return LRP_EALIGN;
        const int gpb = 256 / lpg;
        const int64_t ngroups = (int64_t)B * S * Hq;
        int64_t nb = (ngroups + gpb - 1) / gpb;
        if (nb > 4096) nb = 4096;
        hipLaunchKernelGGL((attn_bwd_prep_kernel<T, EPC>), dim3((unsigned)nb), dim3(256), 0, st, (const T*)Go, (const T*)o, (T*)Gho, D, B, S, Hq, d, ldgo, ldo, ldgho, eps_pv, factor);
    })
    return lrp_check_launch();
}

extern "C" int lrp_head_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, int64_t rows, int heads, int d, int64_t ldx,
                                    int64_t ldy, float eps, float w_offset, int dtype, void* stream) {
    if (!x || !w || !y || !rstd || rows < 0 || heads < 1 || d < 1) return LRP_EINVAL;
    if (rows == 0) return LRP_OK;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype, {
        constexpr int EPC = 16 / sizeof(T);
        const int lpg = d / EPC;
        if (d % EPC || lpg < 1 || lpg > 64 || (lpg & (lpg - 1))) return LRP_ESHAPE;
        if (!al16(x) || !al16(w) || !al16(y) || (ldx % EPC) || (ldy % EPC)) return LRP_EALIGN;
        const int gpb = 256 / lpg;
        int64_t nb = (rows * heads + gpb - 1) / gpb;
        if (nb > 8192) nb = 8192;
        hipLaunchKernelGGL((head_rmsnorm_fwd_kernel<T, EPC>), dim3((unsigned)nb), dim3(256), 0, st, (const T*)x, (const T*)w, (T*)y, rstd, rows, heads, d, ldx, ldy, eps, w_offset);
    })
    return lrp_check_launch();
}

extern "C" int lrp_head_rmsnorm_bwd(const void* G, const void* w, const float* rstd, void* out, int64_t rows, int heads, int d, int64_t ldg,
                                    int64_t ldo, float w_offset, int dtype, void* stream) {
    if (!G || !w || !rstd || !out || rows < 0 || heads < 1 || d < 1) return LRP_EINVAL;
    if (rows == 0) return LRP_OK;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype, {
        constexpr int EPC = 16 / sizeof(T);
        const int lpg = d / EPC;
        if (d % EPC || lpg < 1 || lpg > 64 || (lpg & (lpg - 1))) return LRP_ESHAPE;
        if (!al16(G) || !al16(w) || !al16(out) || (ldg % EPC) || (ldo % EPC)) return LRP_EALIGN;
        const int gpb = 256 / lpg;
        int64_t nb = (rows * heads + gpb - 1) / gpb;
        if (nb > 8192) nb = 8192;
        hipLaunchKernelGGL((head_rmsnorm_bwd_kernel<T, EPC>), dim3((unsigned)nb), dim3(256), 0, st, (const T*)G, (const T*)w, rstd, (T*)out, rows, heads, d, ldg, ldo, w_offset);
    })
    return lrp_check_launch();
}
