// sandwich.hip -- Gemma-3's row kernels, fused per SITE instead of per module (round 6; HBM-bound, one pass each):
//   * the "sandwich" norms around a sub-layer output (ref HF Gemma3DecoderLayer: h1 = h + post_attention_layernorm(a); x2 =
//     pre_feedforward_layernorm(h1) -- and the same pair at the layer boundary: h' = h1 + post_feedforward_layernorm(dn); x' =
//     input_layernorm(h')), patched by lxt/efficient/models/gemma3.py:11-19 as identity rules (rstd detached): lrp_sandwich_norm_fwd / _bwd
//     replace two lrp_add_rmsnorm_fwd / two lrp_rmsnorm_bwd_add2 launches (the normed branch never goes to memory);
//   * per-head q / k RMSNorm + RoPE on the fused projection output (HF Gemma3Attention: q_norm, k_norm, apply_rotary_pos_emb):
//     lrp_qk_norm_rope_fwd replaces 2 x lrp_head_rmsnorm_fwd + 2 x lrp_rope_fwd; lrp_qkv_bwd_pack replaces 2 x lrp_gqa_reduce + 2 x lrp_rope_bwd +
//     2 x lrp_head_rmsnorm_bwd: it writes the qkv dgrad's operand [dq-part | dk-part | dv-part] in one pass.
// Every kernel reproduces the arithmetic of the launch sequence it replaces, INCLUDING the roundings to the storage type at the points where
// that sequence went through memory: results are bit-identical to the un-fused sequence (tests/test_kernels_gpu.py::test_sandwich_*).
#include "common.hpp"

namespace {

inline bool al16(const void* p) { return !p || (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline int row_threads(int H, int epc) { return (H / epc <= 64) ? 64 : 256; }          // (as rowops.hip: the reduction tree must be the same)

template <typename T, int W> struct SChunk {
    float v[W];
    LRP_DEVICE void load(const T* p) {
        Vec16<T> t = ld16(p);
#pragma unroll
        for (int i = 0; i < W; ++i) v[i] = t.get(i);
    }
    LRP_DEVICE void store(T* p) const {
        Vec16<T> t;
#pragma unroll
        for (int i = 0; i < W; ++i) t.set(i, v[i]);
        st16(p, t);
    }
};

// y = w' (*) x rstd in the two conventions of add_rmsnorm_fwd_kernel: HF Llama (w_off = 0: w * (x rstd).to(dtype)), Gemma-3 (fp32 product)
template <typename T> LRP_DEVICE float norm_scale(float x, float rs, float w, float w_off) {
    return (w_off == 0.f) ? w * to_f32(from_f32<T>(x * rs)) : (x * rs) * (w_off + w);
}
template <typename T> LRP_DEVICE float rnd(float x) { return to_f32(from_f32<T>(x)); }
// W consecutive fp32 table entries as 16-byte loads (tables [seq, d] fp32, column a multiple of W = 4 or 8, base 16-byte aligned: checked by the host)
template <int W> LRP_DEVICE void load_tab(float (&t)[W], const float* p) {
#pragma unroll
    for (int q = 0; q < W / 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p + 4 * q);
        t[4 * q] = v[0]; t[4 * q + 1] = v[1]; t[4 * q + 2] = v[2]; t[4 * q + 3] = v[3];
    }
}

// hsum = res + T(w_post' (*) x rstd(x));  y = w_pre' (*) hsum rstd(hsum).  One workgroup per row, the row stays in registers (CH chunks per thread)
template <typename T, int W, int CH>
__global__ __launch_bounds__(256) void sandwich_norm_fwd_kernel(const T* __restrict__ x, const T* __restrict__ res, const T* __restrict__ w_post,
                                                                const T* __restrict__ w_pre, T* __restrict__ hsum_out, T* __restrict__ y,
                                                                float* __restrict__ rstd_post, float* __restrict__ rstd_pre, int H, float eps,
                                                                float w_off) {
    __shared__ float red[4];
    const int64_t row = blockIdx.x;
    const int nch = H / W;
    SChunk<T, W> a[CH];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int c = threadIdx.x + i * blockDim.x;
        if (c < nch) {
            a[i].load(x + row * H + (int64_t)c * W);
#pragma unroll
            for (int k = 0; k < W; ++k) ss += a[i].v[k] * a[i].v[k];
        }
    }
    ss = block_sum(ss, red);
    const float rs_a = rsqrtf(ss / (float)H + eps);
    if (threadIdx.x == 0) rstd_post[row] = rs_a;
    float ss2 = 0.f;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int c = threadIdx.x + i * blockDim.x;
        if (c < nch) {
            SChunk<T, W> r, ww;
            r.load(res + row * H + (int64_t)c * W);
            ww.load(w_post + (int64_t)c * W);
#pragma unroll
            for (int k = 0; k < W; ++k) {
                const float pa = rnd<T>(norm_scale<T>(a[i].v[k], rs_a, ww.v[k], w_off));       // (the branch as the first launch stored it)
                a[i].v[k] = rnd<T>(r.v[k] + pa);                                               // residual add in the storage type
                ss2 += a[i].v[k] * a[i].v[k];
            }
            a[i].store(hsum_out + row * H + (int64_t)c * W);
        }
    }
    ss2 = block_sum(ss2, red);
    const float rs_h = rsqrtf(ss2 / (float)H + eps);
    if (threadIdx.x == 0) rstd_pre[row] = rs_h;
    if (y == nullptr) return;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int c = threadIdx.x + i * blockDim.x;
        if (c < nch) {
            SChunk<T, W> ww, o;
            ww.load(w_pre + (int64_t)c * W);
#pragma unroll
            for (int k = 0; k < W; ++k) o.v[k] = norm_scale<T>(a[i].v[k], rs_h, ww.v[k], w_off);
            o.store(y + row * H + (int64_t)c * W);
        }
    }
}

// Gs = T(Gres + Gx w_pre' rstd_pre)  (gradient w.r.t. the residual sum: the pre-norm's identity rule + the add);  Ga = Gs w_post' rstd_post
template <typename T, int W>
__global__ void sandwich_norm_bwd_kernel(const T* __restrict__ Gres, const T* __restrict__ Gx, const T* __restrict__ w_pre,
                                         const float* __restrict__ rstd_pre, const T* __restrict__ w_post, const float* __restrict__ rstd_post,
                                         T* __restrict__ Gs_out, T* __restrict__ Ga_out, int64_t M, int H, float w_off) {
    const int nch = H / W;
    const int64_t total = M * nch;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / nch;
        const int c = (int)(i - row * nch) * W;
        SChunk<T, W> gx, gr, w1, w2, os, oa;
        gx.load(Gx + row * H + c);
        w1.load(w_pre + c);
        w2.load(w_post + c);
        if (Gres) gr.load(Gres + row * H + c);
        const float rs1 = rstd_pre[row], rs2 = rstd_post[row];
#pragma unroll
        for (int k = 0; k < W; ++k) {
            float gh = gx.v[k] * (w1.v[k] + w_off) * rs1;
            if (Gres) gh += gr.v[k];
            const float gs = rnd<T>(gh);
            os.v[k] = gs;
            oa.v[k] = gs * (w2.v[k] + w_off) * rs2;
        }
        os.store(Gs_out + row * H + c);
        oa.store(Ga_out + row * H + c);
    }
}

// one lane group of d / W lanes per (row, head) of the fused projection output [q heads | k heads | v heads]: RMSNorm over the head (rounded to
// the storage type, as the stand-alone kernel stored it), then RoPE in rotate-half form: the partner column c +- d/2 sits lpg / 2 lanes away
template <typename T, int W>
__global__ void qk_norm_rope_fwd_kernel(const T* __restrict__ qkv, const T* __restrict__ wq, const T* __restrict__ wk, T* __restrict__ qr,
                                        T* __restrict__ kr, float* __restrict__ rstd_q, float* __restrict__ rstd_k, const float* __restrict__ cs,
                                        const float* __restrict__ sn, int64_t rows, int seq, int nq, int nk, int d, int64_t ldx, int64_t ldq,
                                        int64_t ldk, float eps, float w_off) {
    const int lpg = d / W, gpb = blockDim.x / lpg, lg = threadIdx.x % lpg, heads = nq + nk;
    const int64_t ngroups = rows * heads;
    const bool first = lg < (lpg >> 1);
    for (int64_t gidx = (int64_t)blockIdx.x * gpb + threadIdx.x / lpg; gidx < ngroups; gidx += (int64_t)gridDim.x * gpb) {
        const int h = (int)(gidx % heads);
        const int64_t row = gidx / heads;
        const bool isq = h < nq;
        SChunk<T, W> a, ww, o;
        a.load(qkv + row * ldx + (int64_t)h * d + lg * W);
        ww.load((isq ? wq : wk) + lg * W);
        float ss = 0.f;
#pragma unroll
        for (int k = 0; k < W; ++k) ss += a.v[k] * a.v[k];
        for (int off = lpg >> 1; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
        const float rs = rsqrtf(ss / (float)d + eps);
        if (lg == 0) {
            if (isq) rstd_q[row * nq + h] = rs;
            else rstd_k[row * nk + (h - nq)] = rs;
        }
        const int pos = (int)(row % seq);
        float pc[W], ps[W];
        load_tab<W>(pc, cs + (int64_t)pos * d + lg * W);
        load_tab<W>(ps, sn + (int64_t)pos * d + lg * W);
#pragma unroll
        for (int k = 0; k < W; ++k) {
            const float n = rnd<T>(norm_scale<T>(a.v[k], rs, ww.v[k], w_off));
            const float p = __shfl_xor(n, lpg >> 1, 64);
            o.v[k] = first ? n * pc[k] - p * ps[k] : n * pc[k] + p * ps[k];
        }
        if (isq) o.store(qr + row * ldq + (int64_t)h * d + lg * W);
        else o.store(kr + row * ldk + (int64_t)(h - nq) * d + lg * W);
    }
}

// the qkv dgrad's operand in one pass: q part = rope^T(dq) (*) wq' rstd_q; k part = rope^T(sum over the group's query heads of dk_h) (*) wk' rstd_k;
// v part = the group sum of dv_h.  Roundings to the storage type where the six-launch sequence went through memory.
template <typename T, int W>
__global__ void qkv_bwd_pack_kernel(const T* __restrict__ dq, const T* __restrict__ dk_h, const T* __restrict__ dv_h, const T* __restrict__ wq,
                                    const T* __restrict__ wk, const float* __restrict__ rstd_q, const float* __restrict__ rstd_k,
                                    const float* __restrict__ cs, const float* __restrict__ sn, T* __restrict__ A, int64_t rows, int seq, int nq,
                                    int nk, int rep, int d, int64_t lddq, int64_t lddk, int64_t lddv, int64_t lda, float w_off) {
    const int lpg = d / W, gpb = blockDim.x / lpg, lg = threadIdx.x % lpg, heads = nq + 2 * nk, hd = d >> 1;
    const int64_t ngroups = rows * heads;
    const bool first = lg < (lpg >> 1);
    for (int64_t gidx = (int64_t)blockIdx.x * gpb + threadIdx.x / lpg; gidx < ngroups; gidx += (int64_t)gridDim.x * gpb) {
        const int h = (int)(gidx % heads);
        const int64_t row = gidx / heads;
        SChunk<T, W> p, ww, o;
        float rs = 0.f;
        if (h < nq) {
            p.load(dq + row * lddq + (int64_t)h * d + lg * W);
            ww.load(wq + lg * W);
            rs = rstd_q[row * nq + h];
        } else {
            const bool isk = h < nq + nk;
            const int hk = isk ? h - nq : h - nq - nk;
            const T* src = (isk ? dk_h + row * lddk : dv_h + row * lddv) + (int64_t)(hk * rep) * d + lg * W;
            float acc[W];
#pragma unroll
            for (int k = 0; k < W; ++k) acc[k] = 0.f;
            for (int gq = 0; gq < rep; ++gq) {
                SChunk<T, W> t;
                t.load(src + (int64_t)gq * d);
#pragma unroll
                for (int k = 0; k < W; ++k) acc[k] += t.v[k];
            }
#pragma unroll
            for (int k = 0; k < W; ++k) p.v[k] = rnd<T>(acc[k]);
            if (!isk) {
                p.store(A + row * lda + (int64_t)h * d + lg * W);
                continue;
            }
            ww.load(wk + lg * W);
            rs = rstd_k[row * nk + hk];
        }
        const int pos = (int)(row % seq);
        float pc[W], ps[W];
        load_tab<W>(pc, cs + (int64_t)pos * d + lg * W);
        load_tab<W>(ps, sn + (int64_t)pos * d + (first ? lg * W + hd : lg * W - hd));       // the PARTNER's sine
#pragma unroll
        for (int k = 0; k < W; ++k) {
            const float q = __shfl_xor(p.v[k], lpg >> 1, 64);
            const float a = rnd<T>(first ? p.v[k] * pc[k] + q * ps[k] : p.v[k] * pc[k] - q * ps[k]);
            o.v[k] = a * (ww.v[k] + w_off) * rs;
        }
        o.store(A + row * lda + (int64_t)h * d + lg * W);
    }
}

template <typename T>
void launch_sandwich_fwd(const void* x, const void* res, const void* w_post, const void* w_pre, void* hsum_out, void* y, float* rstd_post,
                         float* rstd_pre, int M, int H, float eps, float w_off, hipStream_t st) {
    constexpr int EPC = 16 / sizeof(T);
    const int nt = row_threads(H, EPC), need = (H / EPC + nt - 1) / nt;
#define LRP_SW_FWD(CH) hipLaunchKernelGGL((sandwich_norm_fwd_kernel<T, EPC, CH>), dim3(M), dim3(nt), 0, st, (const T*)x, (const T*)res,       \
                                          (const T*)w_post, (const T*)w_pre, (T*)hsum_out, (T*)y, rstd_post, rstd_pre, H, eps, w_off)
    if (need <= 1) LRP_SW_FWD(1);
    else if (need <= 2) LRP_SW_FWD(2);
    else LRP_SW_FWD(4);
#undef LRP_SW_FWD
}

}  // namespace

#define DISPATCH_T(dtype, ...)                                              \
    if (dtype == LRP_F32) { typedef float T; __VA_ARGS__ }                  \
    else if (dtype == LRP_BF16) { typedef bf16_t T; __VA_ARGS__ }           \
    else return LRP_EINVAL;

extern "C" int lrp_sandwich_norm_ok(int H, int dtype) {
    const int epc = dtype == LRP_F32 ? 4 : 8;
    if (dtype != LRP_F32 && dtype != LRP_BF16) return 0;
    return H >= epc && H % epc == 0 && (H / epc + row_threads(H, epc) - 1) / row_threads(H, epc) <= 4;
}

extern "C" int lrp_sandwich_norm_fwd(const void* x, const void* res, const void* w_post, const void* w_pre, void* hsum_out, void* y,
                                     float* rstd_post, float* rstd_pre, int M, int H, float eps, float w_offset, int dtype, void* stream) {
    if (!x || !res || !w_post || !hsum_out || !rstd_post || !rstd_pre || (y && !w_pre) || M < 0 || H < 1) return LRP_EINVAL;
    if (M == 0) return LRP_OK;
    if (!lrp_sandwich_norm_ok(H, dtype)) return LRP_ESHAPE;
    if (!al16(x) || !al16(res) || !al16(w_post) || !al16(w_pre) || !al16(hsum_out) || !al16(y)) return LRP_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype, { launch_sandwich_fwd<T>(x, res, w_post, w_pre, hsum_out, y, rstd_post, rstd_pre, M, H, eps, w_offset, st); })
    return lrp_check_launch();
}

extern "C" int lrp_sandwich_norm_bwd(const void* Gres, const void* Gx, const void* w_pre, const float* rstd_pre, const void* w_post,
                                     const float* rstd_post, void* Gs_out, void* Ga_out, int M, int H, float w_offset, int dtype, void* stream) {
    if (!Gx || !w_pre || !rstd_pre || !w_post || !rstd_post || !Gs_out || !Ga_out || M < 0 || H < 1) return LRP_EINVAL;
    if (M == 0) return LRP_OK;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype, {
        constexpr int EPC = 16 / sizeof(T);
        if (H % EPC) return LRP_ESHAPE;
        if (!al16(Gres) || !al16(Gx) || !al16(w_pre) || !al16(w_post) || !al16(Gs_out) || !al16(Ga_out)) return LRP_EALIGN;
        int64_t nb = ((int64_t)M * (H / EPC) + 255) / 256;
        if (nb > 16384) nb = 16384;
        hipLaunchKernelGGL((sandwich_norm_bwd_kernel<T, EPC>), dim3((unsigned)nb), dim3(256), 0, st, (const T*)Gres, (const T*)Gx, (const T*)w_pre,
                           rstd_pre, (const T*)w_post, rstd_post, (T*)Gs_out, (T*)Ga_out, (int64_t)M, H, w_offset);
    })
    return lrp_check_launch();
}

static int qk_shape_ok(int d, int dtype) {
    const int epc = dtype == LRP_F32 ? 4 : 8;
    const int lpg = d / epc;
    return (d % epc == 0) && lpg >= 2 && lpg <= 64 && !(lpg & (lpg - 1));
}

extern "C" int lrp_qk_norm_rope_fwd(const void* qkv, const void* wq, const void* wk, void* qr, void* kr, float* rstd_q, float* rstd_k,
                                    const float* cos_t, const float* sin_t, int64_t rows, int seq, int nq, int nk, int d, int64_t ldqkv,
                                    int64_t ldq, int64_t ldk, float eps, float w_offset, int dtype, void* stream) {
    if (!qkv || !wq || !wk || !qr || !kr || !rstd_q || !rstd_k || !cos_t || !sin_t || rows < 0 || seq < 1 || nq < 1 || nk < 1 || d < 2) return LRP_EINVAL;
    if (rows == 0) return LRP_OK;
    if (dtype != LRP_F32 && dtype != LRP_BF16) return LRP_EINVAL;
    if (!qk_shape_ok(d, dtype)) return LRP_ESHAPE;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype, {
        constexpr int EPC = 16 / sizeof(T);
        if (!al16(qkv) || !al16(wq) || !al16(wk) || !al16(qr) || !al16(kr) || !al16(cos_t) || !al16(sin_t) || (ldqkv % EPC) || (ldq % EPC) || (ldk % EPC))
            return LRP_EALIGN;
        const int gpb = 256 / (d / EPC);
        int64_t nb = (rows * (nq + nk) + gpb - 1) / gpb;
        if (nb > 16384) nb = 16384;
        hipLaunchKernelGGL((qk_norm_rope_fwd_kernel<T, EPC>), dim3((unsigned)nb), dim3(256), 0, st, (const T*)qkv, (const T*)wq, (const T*)wk, (T*)qr,
                           (T*)kr, rstd_q, rstd_k, cos_t, sin_t, rows, seq, nq, nk, d, ldqkv, ldq, ldk, eps, w_offset);
    })
    return lrp_check_launch();
}

extern "C" int lrp_qkv_bwd_pack(const void* dq, const void* dk_h, const void* dv_h, const void* wq, const void* wk, const float* rstd_q,
                                const float* rstd_k, const float* cos_t, const float* sin_t, void* A, int64_t rows, int seq, int nq, int nk, int d,
                                int64_t lddq, int64_t lddk, int64_t lddv, int64_t lda, float w_offset, int dtype, void* stream) {
    if (!dq || !dk_h || !dv_h || !wq || !wk || !rstd_q || !rstd_k || !cos_t || !sin_t || !A || rows < 0 || seq < 1 || nq < 1 || nk < 1 || d < 2)
        return LRP_EINVAL;
    if (nq % nk) return LRP_ESHAPE;
    if (rows == 0) return LRP_OK;
    if (dtype != LRP_F32 && dtype != LRP_BF16) return LRP_EINVAL;
    if (!qk_shape_ok(d, dtype)) return LRP_ESHAPE;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype, {
        constexpr int EPC = 16 / sizeof(T);
        if (!al16(dq) || !al16(dk_h) || !al16(dv_h) || !al16(wq) || !al16(wk) || !al16(A) || !al16(cos_t) || !al16(sin_t) || (lddq % EPC) ||
            (lddk % EPC) || (lddv % EPC) || (lda % EPC))
            return LRP_EALIGN;
        const int gpb = 256 / (d / EPC);
        int64_t nb = (rows * (nq + 2 * nk) + gpb - 1) / gpb;
        if (nb > 16384) nb = 16384;
        hipLaunchKernelGGL((qkv_bwd_pack_kernel<T, EPC>), dim3((unsigned)nb), dim3(256), 0, st, (const T*)dq, (const T*)dk_h, (const T*)dv_h,
                           (const T*)wq, (const T*)wk, rstd_q, rstd_k, cos_t, sin_t, (T*)A, rows, seq, nq, nk, nq / nk, d, lddq, lddk, lddv, lda,
                           w_offset);
    })
    return lrp_check_launch();
}
