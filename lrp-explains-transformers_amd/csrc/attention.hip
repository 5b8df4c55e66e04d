// attention.hip -- K4: flash-style AttnLRP attention on MFMA (forward, dQ backward, dK/dV backward).
//
// Orientation trick (no LDS round trip for P, no in-kernel transposes):
//   every kernel has a "row side" that lives in registers (forward/dQ: 16 queries per wave; dK/dV:
//   16 keys per wave) and a "column side" that streams through LDS in tiles of 128 BYTES per
//   transposed row (64 bf16 / 32 fp32 columns).  Scores are computed TRANSPOSED w.r.t. the row
//   side -- mma(col-side frag, row-side frag) -- so lane l owns row-side element (l & 15) and the
//   four D registers are four CONSECUTIVE column-side elements (l>>4)*4 + r.  Hence
//     * the softmax statistics of a row (m, l, lse, D) are per-lane scalars (2 xor-shuffles reduce),
//     * P / Ghs in registers ARE the second MFMA operand of the next contraction (k-slot (l>>4, e)
//       <-> column (l>>4)*4+e of a 16-column sub-tile; bf16 packs two sub-tiles into K=32),
//     * the other operand of that contraction must be column-contiguous, i.e. a TRANSPOSED copy
//       (V^T for O, K^T for dQ, Q^T / Gho^T for dK / dV).  Those copies are made once per layer in
//       HBM by lrp_transpose_heads (288 GB: keep both layouts), so every LDS fill is a straight copy.
//   The out^T accumulators put 4 consecutive head-dim columns of one row in a lane: 8/16-byte stores.
// LDS tiles are XOR-swizzled per 16-byte chunk with (row & min(chunks_per_row,16)-1): conflict-free
// ds_read_b128 / ds_read_b64 for every pitch used here.  One LDS buffer + register prefetch of the
// next tile (global loads are issued before the MFMAs of the current tile).
#include "common.hpp"
#include <type_traits>

namespace {

constexpr int ANT = 256;      // 4 waves

template <typename T> struct AT {
    static constexpr int SZ = sizeof(T);
    static constexpr int EPC = 16 / SZ;
    static constexpr int CT = 128 / SZ;       // column-side tile: 64 bf16 / 32 fp32 elements = 128 B
    static constexpr int NC16 = CT / 16;      // 16-wide sub-tiles per column tile (4 / 2)
};

LRP_DEVICE int swz_mask(int pitch_bytes) {
    const int cpr = pitch_bytes >> 4;
    return (cpr >= 16 ? 16 : cpr) - 1;
}

// cooperative global->register fetch of a [ROWS x PITCH bytes] tile; NCH 16-byte chunks per thread
template <int ROWS, int PITCH>
struct TileStage {
    static constexpr int CPR = PITCH / 16;
    static constexpr int NCHUNK = ROWS * CPR;
    static constexpr int NCH = (NCHUNK + ANT - 1) / ANT;
    u32x4 r[NCH];
    // gbase: address of (row 0, byte 0); row_stride in bytes; rows_valid / bytes_valid bound the read
    LRP_DEVICE void gload(const char* gbase, int64_t row_stride, int rows_valid, int bytes_valid) {
#pragma unroll
        for (int p = 0; p < NCH; ++p) {
            const int id = threadIdx.x + p * ANT;
            const int row = id / CPR, c = id % CPR;
            const bool ok = (id < NCHUNK) && (row < rows_valid) && (c * 16 < bytes_valid);
            r[p] = ok ? *reinterpret_cast<const u32x4*>(gbase + (int64_t)row * row_stride + c * 16) : u32x4{0, 0, 0, 0};
        }
    }
    LRP_DEVICE void swrite(char* lds) const {
        constexpr int SW = (CPR >= 16 ? 16 : CPR) - 1;
#pragma unroll
        for (int p = 0; p < NCH; ++p) {
            const int id = threadIdx.x + p * ANT;
            const int row = id / CPR, c = id % CPR;
            if (id < NCHUNK) *reinterpret_cast<u32x4*>(lds + row * PITCH + ((c ^ (row & SW)) << 4)) = r[p];
        }
    }
};

template <int PITCH> LRP_DEVICE const char* lds_chunk(const char* lds, int row, int chunk) {
    constexpr int CPR = PITCH / 16;
    constexpr int SW = (CPR >= 16 ? 16 : CPR) - 1;
    return lds + row * PITCH + ((chunk ^ (row & SW)) << 4);
}

// row-major tile [rows][D]: operand fragment of row (r16*16 + l&15), 64-byte K chunk c
template <typename T, int D> LRP_DEVICE typename Mma16<T>::frag rm_frag(const char* lds, int r16, int c, int lane) {
    constexpr int PITCH = D * (int)sizeof(T);
    const int row = r16 * 16 + (lane & 15);
    return *reinterpret_cast<const typename Mma16<T>::frag*>(lds_chunk<PITCH>(lds, row, c * 4 + (lane >> 4)));
}
// transposed tile [D rows][128 B]: fragment of head-dim row (dt*16 + l&15) for macro k-step kc (64 B
// of columns).  Slot order must match the register-resident operand built by pack_cols():
//   bf16: e<4 -> column 32kc + 4g + e ; e>=4 -> column 32kc + 16 + 4g + (e-4)   (two 8-byte reads)
//   fp32: e   -> column 16kc + 4g + e                                           (one 16-byte read)
template <typename T> LRP_DEVICE typename Mma16<T>::frag tr_frag(const char* lds, int dt, int kc, int lane);
template <> LRP_DEVICE bf16x8 tr_frag<bf16_t>(const char* lds, int dt, int kc, int lane) {
    const int row = dt * 16 + (lane & 15), g = lane >> 4;
    const char* p0 = lds_chunk<128>(lds, row, 4 * kc + (g >> 1)) + (g & 1) * 8;
    const char* p1 = lds_chunk<128>(lds, row, 4 * kc + 2 + (g >> 1)) + (g & 1) * 8;
    const u32x2 a = *reinterpret_cast<const u32x2*>(p0);
    const u32x2 b = *reinterpret_cast<const u32x2*>(p1);
    u32x4 v = {a[0], a[1], b[0], b[1]};
    return __builtin_bit_cast(bf16x8, v);
}
template <> LRP_DEVICE f32x4 tr_frag<float>(const char* lds, int dt, int kc, int lane) {
    const int row = dt * 16 + (lane & 15), g = lane >> 4;
    return *reinterpret_cast<const f32x4*>(lds_chunk<128>(lds, row, 4 * kc + g));
}
// register-resident second operand from the D registers of the column sub-tiles of macro step kc
template <typename T> struct PackCols;
template <> struct PackCols<bf16_t> {   // sub-tiles 2kc, 2kc+1
    static LRP_DEVICE bf16x8 pack(const f32x4* x, int kc) {
        bf16x8 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) { r[e] = (bf16_t)x[2 * kc][e]; r[4 + e] = (bf16_t)x[2 * kc + 1][e]; }
        return r;
    }
};
template <> struct PackCols<float> {    // sub-tile kc
    static LRP_DEVICE f32x4 pack(const f32x4* x, int kc) { return x[kc]; }
};

// row-side operand fragments straight from global memory (token-major [rows, H, D], one head)
template <typename T, int D>
LRP_DEVICE void load_row_frags(typename Mma16<T>::frag* f, const T* base, int64_t ld, int row, int rows_valid, int lane) {
    constexpr int NDC = D * (int)sizeof(T) / 64;
    constexpr int EPC = 16 / (int)sizeof(T);
    const bool ok = row < rows_valid;
#pragma unroll
    for (int c = 0; c < NDC; ++c) {
        if (ok) f[c] = *reinterpret_cast<const typename Mma16<T>::frag*>(base + (int64_t)row * ld + (c * 4 + (lane >> 4)) * EPC);
        else {
            u32x4 z = {0, 0, 0, 0};
            f[c] = __builtin_bit_cast(typename Mma16<T>::frag, z);
        }
    }
}

// store out^T accumulators: lane holds out[row = l&15][col = dt*16 + (l>>4)*4 + r]
template <typename T, int D>
LRP_DEVICE void store_rows(T* base, int64_t ld, int row, int rows_valid, const f32x4* acc, float mul, int lane) {
    if (row >= rows_valid) return;
#pragma unroll
    for (int dt = 0; dt < D / 16; ++dt) {
        T* dst = base + (int64_t)row * ld + dt * 16 + (lane >> 4) * 4;
        if constexpr (sizeof(T) == 4) {
            f32x4 v = acc[dt] * mul;
            *reinterpret_cast<f32x4*>(dst) = v;
        } else {
            bf16x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (bf16_t)(acc[dt][r] * mul);
            *reinterpret_cast<bf16x4*>(dst) = v;
        }
    }
}

// (causal, window) describe the STRUCTURE of the mask (they also bound which tiles are visited); [lo, hi) is the optional
// per-query-row key interval (padding, packed sequences, bidirectional blocks) that refines it element-wise
LRP_DEVICE bool visible(int q, int key, int S, int causal, int window, int lo, int hi) {
    return key < S && (!causal || key <= q) && (window <= 0 || key > q - window) && key >= lo && key < hi;
}

// =================================================================================================
// forward: o = softmax(scale q k^T + mask) v ; lse
// =================================================================================================
template <typename T, int D, int QSUB>
__global__ __launch_bounds__(ANT, (D >= 256 ? 1 : 2)) void attn_fwd_kernel(
    const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ vt, T* __restrict__ o, float* __restrict__ lse,
    int S, int Hq, int Hkv, int64_t ldq, int64_t ldk, int64_t ldt, int64_t ldo, float scale, int causal, int window, int q_begin,
    const int* __restrict__ row_lo, const int* __restrict__ row_hi) {
    typedef AT<T> A;
    typedef typename Mma16<T>::frag frag_t;
    constexpr int SZ = A::SZ, CT = A::CT, NC16 = A::NC16, NDC = D * SZ / 64, ND16 = D / 16;
    constexpr int KP = D * SZ;                       // K tile pitch (bytes)
    constexpr int BQ = 4 * 16 * QSUB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sK = smem;                                 // [CT][KP]
    char* sV = smem + CT * KP;                       // [D][128]

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y, hk = h / (Hq / Hkv);
    const int qblk = gridDim.x - 1 - blockIdx.x;     // heavy (late) causal blocks first
    const int q0 = qblk * BQ, qw = q0 + wave * 16 * QSUB;
    if (q0 + BQ <= q_begin) return;                  // rows below q_begin are not needed (top-layer sparsity)
    const T* qb = q + (int64_t)b * S * ldq + (int64_t)h * D;
    const T* kb = k + (int64_t)b * S * ldk + (int64_t)hk * D;
    const T* vtb = vt + ((int64_t)b * Hkv + hk) * D * ldt;

    frag_t qf[QSUB][NDC];
#pragma unroll
    for (int s = 0; s < QSUB; ++s) load_row_frags<T, D>(qf[s], qb, ldq, qw + s * 16 + (lane & 15), S, lane);

    float m_run[QSUB], l_run[QSUB];
    f32x4 oacc[QSUB][ND16];
#pragma unroll
    for (int s = 0; s < QSUB; ++s) {
        m_run[s] = -INFINITY; l_run[s] = 0.f;
#pragma unroll
        for (int dt = 0; dt < ND16; ++dt) oacc[s][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    int ivlo[QSUB], ivhi[QSUB];                      // per-row key interval (whole row when no interval arrays are given)
#pragma unroll
    for (int s = 0; s < QSUB; ++s) {
        const int qi = qw + s * 16 + (lane & 15);
        ivlo[s] = 0; ivhi[s] = S;
        if (row_lo != nullptr && qi < S) { ivlo[s] = row_lo[(int64_t)b * S + qi]; ivhi[s] = row_hi[(int64_t)b * S + qi]; }
    }

    int kend = S;
    if (causal) kend = min(S, q0 + BQ);
    int kbeg = 0;
    if (window > 0) { kbeg = q0 - window + 1; kbeg = kbeg < 0 ? 0 : (kbeg / CT) * CT; }

    TileStage<CT, KP> stK;
    TileStage<D, 128> stV;
    auto fetch = [&](int kt0) {
        stK.gload(reinterpret_cast<const char*>(kb + (int64_t)kt0 * ldk), ldk * SZ, S - kt0, KP);
        stV.gload(reinterpret_cast<const char*>(vtb + kt0), ldt * SZ, D, (int)min((int64_t)128, (ldt - kt0) * SZ));
    };
    if (kbeg < kend) fetch(kbeg);

    for (int kt0 = kbeg; kt0 < kend; kt0 += CT) {
        __syncthreads();
        stK.swrite(sK);
        stV.swrite(sV);
        __syncthreads();
        if (kt0 + CT < kend) fetch(kt0 + CT);

        // ---- S^T tiles: st[s][t][r] = score(query qw+16s+(l&15), key kt0+16t+4g+r)
        f32x4 st[QSUB][NC16];
#pragma unroll
        for (int s = 0; s < QSUB; ++s)
#pragma unroll
            for (int t = 0; t < NC16; ++t) st[s][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NC16; ++t)
#pragma unroll
            for (int c = 0; c < NDC; ++c) {
                const frag_t kf = rm_frag<T, D>(sK, t, c, lane);
#pragma unroll
                for (int s = 0; s < QSUB; ++s) st[s][t] = Mma16<T>::mma(kf, qf[s][c], st[s][t]);
            }

        const bool need_mask = (kt0 + CT > S) || (causal && kt0 + CT - 1 > qw) || (window > 0) || (row_lo != nullptr);
#pragma unroll
        for (int s = 0; s < QSUB; ++s) {
            const int qi = qw + s * 16 + (lane & 15);
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < NC16; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = st[s][t][r] * scale;
                    if (need_mask && !visible(qi, kt0 + t * 16 + g * 4 + r, S, causal, window, ivlo[s], ivhi[s])) v = -INFINITY;
                    st[s][t][r] = v;
                    mx = fmaxf(mx, v);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run[s], mx);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = __expf(m_run[s] - m_use);
            float rs = 0.f;
#pragma unroll
            for (int t = 0; t < NC16; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __expf(st[s][t][r] - m_use);
                    st[s][t][r] = p;
                    rs += p;
                }
            rs += __shfl_xor(rs, 16, 64);
            rs += __shfl_xor(rs, 32, 64);
            l_run[s] = l_run[s] * alpha + rs;
            m_run[s] = m_new;
#pragma unroll
            for (int dt = 0; dt < ND16; ++dt) oacc[s][dt] *= alpha;
        }

        // ---- O^T += V^T P^T
        frag_t pf[QSUB][2];
#pragma unroll
        for (int s = 0; s < QSUB; ++s)
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) pf[s][kc] = PackCols<T>::pack(st[s], kc);
#pragma unroll
        for (int dt = 0; dt < ND16; ++dt)
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
                const frag_t vf = tr_frag<T>(sV, dt, kc, lane);
#pragma unroll
                for (int s = 0; s < QSUB; ++s) oacc[s][dt] = Mma16<T>::mma(vf, pf[s][kc], oacc[s][dt]);
            }
    }

    T* ob = o + (int64_t)b * S * ldo + (int64_t)h * D;
#pragma unroll
    for (int s = 0; s < QSUB; ++s) {
        const int qi = qw + s * 16 + (lane & 15);
        const float inv = (l_run[s] > 0.f) ? 1.f / l_run[s] : 0.f;
        store_rows<T, D>(ob, ldo, qi, S, oacc[s], inv, lane);
        if (g == 0 && qi < S) lse[((int64_t)b * Hq + h) * S + qi] = m_run[s] + __logf(l_run[s]);
    }
}

// the LRP score-gradient modifier shared by the backward kernels
LRP_DEVICE float lrp_ds(float s_raw, float p, float dp, float Dq, float scale, float eps_mask, float eps_qk) {
    float ds = p * (dp - Dq) * scale;
    if (eps_mask != 0.f) { const float s2 = s_raw * scale; ds *= s2 / (s2 + eps_mask); }
    ds *= (eps_qk == 0.f) ? 0.5f : s_raw / (2.f * s_raw + eps_qk);
    return ds;
}
// branch-free form for the v2 kernels.  EXPL=false: lxt.efficient (factor 1/2).  EXPL=true: both
// explicit stabilisers folded into ONE reciprocal:  s2/(s2+e1) * s/(2s+e2) = s2*s / ((s2+e1)(2s+e2))
// (v_rcp_f32, 1 ulp; the pole at s = -e is the reference's own).
template <bool EXPL>
LRP_DEVICE float lrp_ds2(float s_raw, float p, float dp, float Dq, float scale, float eps_mask, float eps_qk) {
    const float ds = p * (dp - Dq) * scale;
    if constexpr (!EXPL) return ds * 0.5f;
    else {
        const float s2 = s_raw * scale;
        return ds * (s2 * s_raw) * __builtin_amdgcn_rcpf((s2 + eps_mask) * (2.f * s_raw + eps_qk));
    }
}

// softmax in the log2 domain: exp(x*scale - m) = exp2(fma(x, scale*log2e, -m*log2e)) -- one v_fma + one v_exp per element
// instead of mul, sub, mul, exp (the score kernels are VALU-bound next to their MFMAs, so element ops are what counts)
#define LRP_LOG2E 1.4426950408889634f
#define LRP_LN2 0.6931471805599453f
LRP_DEVICE float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// XCD-aware 1-D grid decode.  Workgroups that share column-side tiles (same kv head for forward /
// dQ, same query head for dK/dV) must sit on ONE XCD so those tiles are served by its 4 MiB L2
// instead of being pulled through the fabric by all eight: linear id L runs on XCD L % 8, so the
// sharing group index goes into the low bits.  ngroups sharing groups, per_group work items each;
// returns false for the padding ids of a partially filled last round of 8 groups.
LRP_DEVICE bool xcd_group_decode(int L, int ngroups, int per_group, int& group, int& item) {
    const int rounds = (ngroups + 7) >> 3;
    const int xcd = L & 7, i = L >> 3;
    item = i % per_group;
    group = xcd + 8 * (i / per_group);
    (void)rounds;
    return group < ngroups;
}
inline int xcd_group_grid(int ngroups, int per_group) { return ((ngroups + 7) / 8) * 8 * per_group; }

// =================================================================================================
// backward dQ: row side = queries (registers); column side = keys (K, V row-major + K^T in LDS)
// =================================================================================================
template <typename T, int D>
__global__ __launch_bounds__(ANT, (D >= 256 ? 1 : 2)) void attn_bwd_dq_kernel(
    const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v, const T* __restrict__ kt,
    const T* __restrict__ gho, const float* __restrict__ lse, const float* __restrict__ Dd, T* __restrict__ dq,
    int S, int Hq, int Hkv, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldt, int64_t ldg, int64_t lddq,
    float scale, float eps_mask, float eps_qk, int causal, int window, int q_begin,
    const int* __restrict__ row_lo, const int* __restrict__ row_hi) {
    typedef AT<T> A;
    typedef typename Mma16<T>::frag frag_t;
    constexpr int SZ = A::SZ, CT = A::CT, NC16 = A::NC16, NDC = D * SZ / 64, ND16 = D / 16;
    constexpr int KP = D * SZ;
    constexpr int BQ = 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sK = smem;                       // [CT][KP]
    char* sV = sK + CT * KP;               // [CT][KP]
    char* sKt = sV + CT * KP;              // [D][128]

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y, hk = h / (Hq / Hkv);
    const int qblk = gridDim.x - 1 - blockIdx.x;
    const int q0 = qblk * BQ, qi = q0 + wave * 16 + (lane & 15);
    if (q0 + BQ <= q_begin) return;
    const T* kb = k + (int64_t)b * S * ldk + (int64_t)hk * D;
    const T* vb = v + (int64_t)b * S * ldv + (int64_t)hk * D;
    const T* ktb = kt + ((int64_t)b * Hkv + hk) * D * ldt;

    frag_t qf[NDC], gf[NDC];
    load_row_frags<T, D>(qf, q + (int64_t)b * S * ldq + (int64_t)h * D, ldq, qi, S, lane);
    load_row_frags<T, D>(gf, gho + (int64_t)b * S * ldg + (int64_t)h * D, ldg, qi, S, lane);
    const float lse_q = (qi < S) ? lse[((int64_t)b * Hq + h) * S + qi] : 0.f;
    const float D_q = (qi < S) ? Dd[((int64_t)b * Hq + h) * S + qi] : 0.f;
    int ivlo = 0, ivhi = S;
    if (row_lo != nullptr && qi < S) { ivlo = row_lo[(int64_t)b * S + qi]; ivhi = row_hi[(int64_t)b * S + qi]; }

    f32x4 acc[ND16];
#pragma unroll
    for (int dt = 0; dt < ND16; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

    int kend = S;
    if (causal) kend = min(S, q0 + BQ);
    int kbeg = 0;
    if (window > 0) { kbeg = q0 - window + 1; kbeg = kbeg < 0 ? 0 : (kbeg / CT) * CT; }

    TileStage<CT, KP> stK, stV;
    TileStage<D, 128> stKt;
    auto fetch = [&](int kt0) {
        stK.gload(reinterpret_cast<const char*>(kb + (int64_t)kt0 * ldk), ldk * SZ, S - kt0, KP);
        stV.gload(reinterpret_cast<const char*>(vb + (int64_t)kt0 * ldv), ldv * SZ, S - kt0, KP);
        stKt.gload(reinterpret_cast<const char*>(ktb + kt0), ldt * SZ, D, (int)min((int64_t)128, (ldt - kt0) * SZ));
    };
    if (kbeg < kend) fetch(kbeg);

    for (int kt0 = kbeg; kt0 < kend; kt0 += CT) {
        __syncthreads();
        stK.swrite(sK);
        stV.swrite(sV);
        stKt.swrite(sKt);
        __syncthreads();
        if (kt0 + CT < kend) fetch(kt0 + CT);

        f32x4 st[NC16], dp[NC16];
#pragma unroll
        for (int t = 0; t < NC16; ++t) { st[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int t = 0; t < NC16; ++t)
#pragma unroll
            for (int c = 0; c < NDC; ++c) {
                st[t] = Mma16<T>::mma(rm_frag<T, D>(sK, t, c, lane), qf[c], st[t]);
                dp[t] = Mma16<T>::mma(rm_frag<T, D>(sV, t, c, lane), gf[c], dp[t]);
            }
#pragma unroll
        for (int t = 0; t < NC16; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt0 + t * 16 + g * 4 + r;
                const float s_raw = st[t][r];
                const float p = visible(qi, key, S, causal, window, ivlo, ivhi) ? __expf(s_raw * scale - lse_q) : 0.f;
                st[t][r] = lrp_ds(s_raw, p, dp[t][r], D_q, scale, eps_mask, eps_qk);
            }
        frag_t df[2];
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) df[kc] = PackCols<T>::pack(st, kc);
#pragma unroll
        for (int dt = 0; dt < ND16; ++dt)
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) acc[dt] = Mma16<T>::mma(tr_frag<T>(sKt, dt, kc, lane), df[kc], acc[dt]);
    }
    store_rows<T, D>(dq + (int64_t)b * S * lddq + (int64_t)h * D, lddq, qi, S, acc, 1.f, lane);
}

// =================================================================================================
// backward dK/dV (per query head): row side = keys (registers); column side = queries
// (Q, Gho row-major + Q^T, Gho^T in LDS)
// =================================================================================================
template <typename T, int D>
__global__ __launch_bounds__(ANT, (D >= 256 ? 1 : 2)) void attn_bwd_dkv_kernel(
    const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v, const T* __restrict__ qt,
    const T* __restrict__ gho, const T* __restrict__ ghot, const float* __restrict__ lse, const float* __restrict__ Dd,
    T* __restrict__ dk, T* __restrict__ dv, int S, int Hq, int Hkv, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldt,
    int64_t ldg, int64_t lddk, int64_t lddv, float scale, float eps_mask, float eps_qk, int causal, int window, int q_begin,
    const int* __restrict__ row_lo, const int* __restrict__ row_hi) {
    typedef AT<T> A;
    typedef typename Mma16<T>::frag frag_t;
    constexpr int SZ = A::SZ, CT = A::CT, NC16 = A::NC16, NDC = D * SZ / 64, ND16 = D / 16;
    constexpr int KP = D * SZ;
    constexpr int BK = 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sQ = smem;                       // [CT][KP]
    char* sG = sQ + CT * KP;               // [CT][KP]
    char* sQt = sG + CT * KP;              // [D][128]
    char* sGt = sQt + D * 128;             // [D][128]

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y, hk = h / (Hq / Hkv);
    const int k0 = blockIdx.x * BK, ki = k0 + wave * 16 + (lane & 15);    // early key blocks are the heavy ones
    const T* qb = q + (int64_t)b * S * ldq + (int64_t)h * D;
    const T* gb = gho + (int64_t)b * S * ldg + (int64_t)h * D;
    const T* qtb = qt + ((int64_t)b * Hq + h) * D * ldt;
    const T* gtb = ghot + ((int64_t)b * Hq + h) * D * ldt;
    const float* lse_b = lse + ((int64_t)b * Hq + h) * S;
    const float* D_b = Dd + ((int64_t)b * Hq + h) * S;

    frag_t kf[NDC], vf[NDC];
    load_row_frags<T, D>(kf, k + (int64_t)b * S * ldk + (int64_t)hk * D, ldk, ki, S, lane);
    load_row_frags<T, D>(vf, v + (int64_t)b * S * ldv + (int64_t)hk * D, ldv, ki, S, lane);

    f32x4 dkacc[ND16], dvacc[ND16];
#pragma unroll
    for (int dt = 0; dt < ND16; ++dt) { dkacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dvacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    int qbeg = 0, qend = S;
    if (causal) qbeg = (k0 / CT) * CT;
    if (window > 0) qend = min(S, k0 + BK - 1 + window);
    if (q_begin > qbeg) qbeg = (q_begin / CT) * CT;      // queries below q_begin carry no relevance

    for (int qt0 = qbeg; qt0 < qend; qt0 += CT) {
        __syncthreads();
        {
            TileStage<CT, KP> s1;
            s1.gload(reinterpret_cast<const char*>(qb + (int64_t)qt0 * ldq), ldq * SZ, S - qt0, KP);
            s1.swrite(sQ);
            s1.gload(reinterpret_cast<const char*>(gb + (int64_t)qt0 * ldg), ldg * SZ, S - qt0, KP);
            s1.swrite(sG);
            TileStage<D, 128> s2;
            const int bv = (int)min((int64_t)128, (ldt - qt0) * SZ);
            s2.gload(reinterpret_cast<const char*>(qtb + qt0), ldt * SZ, D, bv);
            s2.swrite(sQt);
            s2.gload(reinterpret_cast<const char*>(gtb + qt0), ldt * SZ, D, bv);
            s2.swrite(sGt);
        }
        __syncthreads();

        // st[t][r] = score(query qt0+16t+4g+r, key ki) ; dp likewise
        f32x4 st[NC16], dp[NC16];
#pragma unroll
        for (int t = 0; t < NC16; ++t) { st[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int t = 0; t < NC16; ++t)
#pragma unroll
            for (int c = 0; c < NDC; ++c) {
                st[t] = Mma16<T>::mma(rm_frag<T, D>(sQ, t, c, lane), kf[c], st[t]);
                dp[t] = Mma16<T>::mma(rm_frag<T, D>(sG, t, c, lane), vf[c], dp[t]);
            }
        f32x4 pp[NC16];
#pragma unroll
        for (int t = 0; t < NC16; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qi = qt0 + t * 16 + g * 4 + r;
                int ivlo = 0, ivhi = S;
                if (row_lo != nullptr && qi < S) { ivlo = row_lo[(int64_t)b * S + qi]; ivhi = row_hi[(int64_t)b * S + qi]; }
                const bool ok = (qi < S) && visible(qi, ki, S, causal, window, ivlo, ivhi);
                const float lq = (qi < S) ? lse_b[qi] : 0.f;
                const float Dq = (qi < S) ? D_b[qi] : 0.f;
                const float s_raw = st[t][r];
                const float p = ok ? __expf(s_raw * scale - lq) : 0.f;
                pp[t][r] = p;
                st[t][r] = lrp_ds(s_raw, p, dp[t][r], Dq, scale, eps_mask, eps_qk);
            }
        frag_t pf[2], df[2];
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) { pf[kc] = PackCols<T>::pack(pp, kc); df[kc] = PackCols<T>::pack(st, kc); }
#pragma unroll
        for (int dt = 0; dt < ND16; ++dt)
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
                dvacc[dt] = Mma16<T>::mma(tr_frag<T>(sGt, dt, kc, lane), pf[kc], dvacc[dt]);
                dkacc[dt] = Mma16<T>::mma(tr_frag<T>(sQt, dt, kc, lane), df[kc], dkacc[dt]);
            }
    }
    store_rows<T, D>(dk + (int64_t)b * S * lddk + (int64_t)h * D, lddk, ki, S, dkacc, 1.f, lane);
    store_rows<T, D>(dv + (int64_t)b * S * lddv + (int64_t)h * D, lddv, ki, S, dvacc, 1.f, lane);
}


// =================================================================================================
// v2 kernels: 8 waves, direct-to-LDS staging (global_load_lds_dwordx4: 1 KiB per wave instruction,
// lane l -> LDS byte 16 l), TWO LDS stages, one barrier per tile.  The tile of step t+1 streams
// into the other stage while the MFMAs of step t run; no staging VGPRs, no ds_write pass, and a
// tile is shared by 8 waves instead of 4 (half the LDS-fill and L2 traffic per MFMA).
// The XOR swizzle is applied to the SOURCE address (the LDS image of a wave instruction is
// lane-linear), the fragment reads use the same involution -- identical LDS image to v1.
// No per-lane predication: row indices are clamped to S-1 (duplicates are masked by `visible`),
// transposed operands are read up to the zero padded ldt (ldt % CT == 0 required).
// =================================================================================================
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

// row-major tile [CT rows][D*SZ bytes] of a token-major operand (row stride ld elements)
template <typename T, int D, int NW>
LRP_DEVICE void glds_rowmajor(const T* base, int64_t ld, int row0, int S, char* lds, int wave, int lane) {
    constexpr int SZ = sizeof(T), EPC = 16 / SZ, KP = D * SZ, CPR = KP / 16, CT = 128 / SZ;
    constexpr int SW = (CPR >= 16 ? 16 : CPR) - 1;
    constexpr int RPG = 64 / CPR;                 // rows per 1-KiB group (CPR <= 64)
    constexpr int NG = CT * KP / 1024;
#pragma unroll
    for (int g = 0; g < (NG + NW - 1) / NW; ++g) {
        const int grp = g * NW + wave;
        if (grp < NG) {
            const int row = grp * RPG + lane / CPR, slot = lane % CPR;
            const int chunk = slot ^ (row & SW);
            int gr = row0 + row;
            gr = gr < S ? gr : S - 1;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(base + (int64_t)gr * ld + chunk * EPC), (lds_ptr_t)(lds + grp * 1024), 16, 0, 0);
        }
    }
}
// transposed tile [D rows][128 B] of a head-transposed operand (row stride ldt), columns c0..c0+CT
template <typename T, int D, int NW>
LRP_DEVICE void glds_transposed(const T* base, int64_t ldt, int c0, char* lds, int wave, int lane) {
    constexpr int EPC = 16 / (int)sizeof(T);
    constexpr int NG = D / 8;
#pragma unroll
    for (int g = 0; g < (NG + NW - 1) / NW; ++g) {
        const int grp = g * NW + wave;
        if (grp < NG) {
            const int row = grp * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ (lane >> 3);
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(base + (int64_t)row * ldt + c0 + chunk * EPC), (lds_ptr_t)(lds + grp * 1024), 16, 0, 0);
        }
    }
}
// 64 fp32 row statistics (lse or D) of rows r0..r0+63 (clamped) -> lds[0..63]
LRP_DEVICE void glds_stats(const float* base, int r0, int S, char* lds, int lane) {
    int r = r0 + lane;
    r = r < S ? r : S - 1;
    __builtin_amdgcn_global_load_lds((glb_ptr_t)(base + r), (lds_ptr_t)lds, 4, 0, 0);
}

template <typename T, int D, bool EXPL>
__global__ __launch_bounds__(512, 2) void attn_bwd_dkv_v2_kernel(
    const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v, const T* __restrict__ qt,
    const T* __restrict__ gho, const T* __restrict__ ghot, const float* __restrict__ lse, const float* __restrict__ Dd,
    T* __restrict__ dk, T* __restrict__ dv, int S, int Hq, int Hkv, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldt,
    int64_t ldg, int64_t lddk, int64_t lddv, float scale, float eps_mask, float eps_qk, int causal, int window, int B, int q_begin,
    const int* __restrict__ row_lo, const int* __restrict__ row_hi) {
    typedef AT<T> A;
    typedef typename Mma16<T>::frag frag_t;
    constexpr int SZ = A::SZ, CT = A::CT, NC16 = A::NC16, NDC = D * SZ / 64, ND16 = D / 16;
    constexpr int NW = 8, BK = 16 * NW;
    constexpr int TILE = 128 * D;                   // bytes of one staged tile (both layouts)
    constexpr int STAGE = 4 * TILE + 512;           // Q, G, Qt, Gt, lse[64], D[64]
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // sharing group = (batch, query head): all its key blocks read the same Q/Gho tiles -> one XCD
    int bh, kblk;
    if (!xcd_group_decode(blockIdx.x, B * Hq, (S + BK - 1) / BK, bh, kblk)) return;
    const int b = bh / Hq, h = bh % Hq, hk = h / (Hq / Hkv);
    const int k0 = kblk * BK, ki = k0 + wave * 16 + (lane & 15);
    const T* qb = q + (int64_t)b * S * ldq + (int64_t)h * D;
    const T* gb = gho + (int64_t)b * S * ldg + (int64_t)h * D;
    const T* qtb = qt + ((int64_t)b * Hq + h) * D * ldt;
    const T* gtb = ghot + ((int64_t)b * Hq + h) * D * ldt;
    const float* lse_b = lse + ((int64_t)b * Hq + h) * S;
    const float* D_b = Dd + ((int64_t)b * Hq + h) * S;

    frag_t kf[NDC], vf[NDC];
    load_row_frags<T, D>(kf, k + (int64_t)b * S * ldk + (int64_t)hk * D, ldk, ki, S, lane);
    load_row_frags<T, D>(vf, v + (int64_t)b * S * ldv + (int64_t)hk * D, ldv, ki, S, lane);

    f32x4 dkacc[ND16], dvacc[ND16];
#pragma unroll
    for (int dt = 0; dt < ND16; ++dt) { dkacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dvacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    const float c1 = scale * LRP_LOG2E;
    const int kw_max = k0 + wave * 16 + 15;            // largest key of this wave (tile-level mask test)
    int qbeg = 0, qend = S;
    if (causal) qbeg = (k0 / CT) * CT;
    if (window > 0) qend = min(S, k0 + BK - 1 + window);
    if (q_begin > qbeg) qbeg = (q_begin / CT) * CT;      // queries below q_begin carry no relevance

    auto stage = [&](int qt0, int buf) {
        char* sb = smem + buf * STAGE;
        glds_rowmajor<T, D, NW>(qb, ldq, qt0, S, sb, wave, lane);
        glds_rowmajor<T, D, NW>(gb, ldg, qt0, S, sb + TILE, wave, lane);
        glds_transposed<T, D, NW>(qtb, ldt, qt0, sb + 2 * TILE, wave, lane);
        glds_transposed<T, D, NW>(gtb, ldt, qt0, sb + 3 * TILE, wave, lane);
        if (wave == 0) glds_stats(lse_b, qt0, S, sb + 4 * TILE, lane);
        if (wave == 1) glds_stats(D_b, qt0, S, sb + 4 * TILE + 256, lane);
    };
    if (qbeg < qend) stage(qbeg, 0);
    __syncthreads();

    int cur = 0;
    for (int qt0 = qbeg; qt0 < qend; qt0 += CT) {
        if (qt0 + CT < qend) stage(qt0 + CT, cur ^ 1);
        const char* sQ = smem + cur * STAGE;
        const char* sG = sQ + TILE;
        const char* sQt = sQ + 2 * TILE;
        const char* sGt = sQ + 3 * TILE;
        const float* sL = reinterpret_cast<const float*>(sQ + 4 * TILE);
        const float* sD = sL + 64;

        f32x4 st[NC16], dp[NC16];
#pragma unroll
        for (int t = 0; t < NC16; ++t) { st[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int t = 0; t < NC16; ++t)
#pragma unroll
            for (int c = 0; c < NDC; ++c) {
                st[t] = Mma16<T>::mma(rm_frag<T, D>(sQ, t, c, lane), kf[c], st[t]);
                dp[t] = Mma16<T>::mma(rm_frag<T, D>(sG, t, c, lane), vf[c], dp[t]);
            }
        f32x4 pp[NC16];
        // interior tiles (every query of the tile sees every key of the wave) skip the per-element mask predicate
        const bool tile_masked = (qt0 + CT > S) || (causal && qt0 < kw_max) || (window > 0) || (row_lo != nullptr);
        {
#pragma unroll
        for (int t = 0; t < NC16; ++t) {
            const f32x4 l4 = *reinterpret_cast<const f32x4*>(sL + t * 16 + g * 4);
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(sD + t * 16 + g * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float s_raw = st[t][r];
                float p = fast_exp2(__builtin_fmaf(s_raw, c1, -(l4[r] * LRP_LOG2E)));
                if (tile_masked) {
                    const int qi = qt0 + t * 16 + g * 4 + r;
                    int ivlo = 0, ivhi = S;
                    if (row_lo != nullptr && qi < S) { ivlo = row_lo[(int64_t)b * S + qi]; ivhi = row_hi[(int64_t)b * S + qi]; }
                    if (!((qi < S) && visible(qi, ki, S, causal, window, ivlo, ivhi))) p = 0.f;
                }
                pp[t][r] = p;
                if constexpr (EXPL) st[t][r] = lrp_ds2<true>(s_raw, p, dp[t][r], d4[r], scale, eps_mask, eps_qk);
                else st[t][r] = p * (dp[t][r] - d4[r]);      // * scale/2 folded into the dK store
            }
        }
        }
        frag_t pf[2], df[2];
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) { pf[kc] = PackCols<T>::pack(pp, kc); df[kc] = PackCols<T>::pack(st, kc); }
#pragma unroll
        for (int dt = 0; dt < ND16; ++dt)
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
                dvacc[dt] = Mma16<T>::mma(tr_frag<T>(sGt, dt, kc, lane), pf[kc], dvacc[dt]);
                dkacc[dt] = Mma16<T>::mma(tr_frag<T>(sQt, dt, kc, lane), df[kc], dkacc[dt]);
            }
        __syncthreads();
        cur ^= 1;
    }
    store_rows<T, D>(dk + (int64_t)b * S * lddk + (int64_t)h * D, lddk, ki, S, dkacc, EXPL ? 1.f : 0.5f * scale, lane);
    store_rows<T, D>(dv + (int64_t)b * S * lddv + (int64_t)h * D, lddv, ki, S, dvacc, 1.f, lane);
}


// ---- v2 forward: 8 waves x (16*QSUB) queries, K + V^T tiles, two LDS stages -----------------------
template <typename T, int D, int QSUB>
__global__ __launch_bounds__(512, 2) void attn_fwd_v2_kernel(
    const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ vt, T* __restrict__ o, float* __restrict__ lse,
    int S, int Hq, int Hkv, int64_t ldq, int64_t ldk, int64_t ldt, int64_t ldo, float scale, int causal, int window, int B, int q_begin,
    const int* __restrict__ row_lo, const int* __restrict__ row_hi) {
    typedef AT<T> A;
    typedef typename Mma16<T>::frag frag_t;
    constexpr int SZ = A::SZ, CT = A::CT, NC16 = A::NC16, NDC = D * SZ / 64, ND16 = D / 16;
    constexpr int NW = 8, BQ = NW * 16 * QSUB;
    constexpr int TILE = 128 * D, STAGE = 2 * TILE;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rep = Hq / Hkv, nqb = (S + BQ - 1) / BQ;
    int bhk, item;      // sharing group = (batch, kv head): its rep*nqb workgroups read the same K/V tiles
    if (!xcd_group_decode(blockIdx.x, B * Hkv, rep * nqb, bhk, item)) return;
    const int b = bhk / Hkv, hk = bhk % Hkv, h = hk * rep + item % rep;
    const int qblk = nqb - 1 - item / rep;           // heavy (late) causal blocks first
    const int q0 = qblk * BQ, qw = q0 + wave * 16 * QSUB;
    if (q0 + BQ <= q_begin) return;
    const T* qb = q + (int64_t)b * S * ldq + (int64_t)h * D;
    const T* kb = k + (int64_t)b * S * ldk + (int64_t)hk * D;
    const T* vtb = vt + ((int64_t)b * Hkv + hk) * D * ldt;

    frag_t qf[QSUB][NDC];
#pragma unroll
    for (int s = 0; s < QSUB; ++s) load_row_frags<T, D>(qf[s], qb, ldq, qw + s * 16 + (lane & 15), S, lane);
    float m_run[QSUB], l_run[QSUB];
    f32x4 oacc[QSUB][ND16];
#pragma unroll
    for (int s = 0; s < QSUB; ++s) {
        m_run[s] = -INFINITY; l_run[s] = 0.f;
#pragma unroll
        for (int dt = 0; dt < ND16; ++dt) oacc[s][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    int ivlo[QSUB], ivhi[QSUB];                      // per-row key interval (whole row when no interval arrays are given)
#pragma unroll
    for (int s = 0; s < QSUB; ++s) {
        const int qi = qw + s * 16 + (lane & 15);
        ivlo[s] = 0; ivhi[s] = S;
        if (row_lo != nullptr && qi < S) { ivlo[s] = row_lo[(int64_t)b * S + qi]; ivhi[s] = row_hi[(int64_t)b * S + qi]; }
    }
    const float c1 = scale * LRP_LOG2E;
    int kend = S;
    if (causal) kend = min(S, q0 + BQ);
    int kbeg = 0;
    if (window > 0) { kbeg = q0 - window + 1; kbeg = kbeg < 0 ? 0 : (kbeg / CT) * CT; }

    auto stage = [&](int kt0, int buf) {
        char* sb = smem + buf * STAGE;
        glds_rowmajor<T, D, NW>(kb, ldk, kt0, S, sb, wave, lane);
        glds_transposed<T, D, NW>(vtb, ldt, kt0, sb + TILE, wave, lane);
    };
    if (kbeg < kend) stage(kbeg, 0);
    __syncthreads();
    int cur = 0;
    for (int kt0 = kbeg; kt0 < kend; kt0 += CT) {
        if (kt0 + CT < kend) stage(kt0 + CT, cur ^ 1);
        const char* sK = smem + cur * STAGE;
        const char* sV = sK + TILE;
        f32x4 st[QSUB][NC16];
#pragma unroll
        for (int s = 0; s < QSUB; ++s)
#pragma unroll
            for (int t = 0; t < NC16; ++t) st[s][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NC16; ++t)
#pragma unroll
            for (int c = 0; c < NDC; ++c) {
                const frag_t kf = rm_frag<T, D>(sK, t, c, lane);
#pragma unroll
                for (int s = 0; s < QSUB; ++s) st[s][t] = Mma16<T>::mma(kf, qf[s][c], st[s][t]);
            }
        const bool need_mask = (kt0 + CT > S) || (causal && kt0 + CT - 1 > qw) || (window > 0) || (row_lo != nullptr);
        // running max m_run is kept in RAW score units (scale > 0, checked by the host); probabilities in the log2 domain:
        // p = exp2(fma(s, c1, -m*c1)) -- per element: max, fma, exp2, add
#pragma unroll
        for (int s = 0; s < QSUB; ++s) {
            const int qi = qw + s * 16 + (lane & 15);
            float mx = -INFINITY;
            if (need_mask) {
#pragma unroll
                for (int t = 0; t < NC16; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (!visible(qi, kt0 + t * 16 + g * 4 + r, S, causal, window, ivlo[s], ivhi[s])) st[s][t][r] = -INFINITY;
            }
#pragma unroll
            for (int t = 0; t < NC16; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[s][t][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run[s], mx);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = fast_exp2((m_run[s] - m_use) * c1);
            const float nm2 = -m_use * c1;
            float rs = 0.f;
#pragma unroll
            for (int t = 0; t < NC16; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = fast_exp2(__builtin_fmaf(st[s][t][r], c1, nm2));
                    st[s][t][r] = p;
                    rs += p;
                }
            rs += __shfl_xor(rs, 16, 64);
            rs += __shfl_xor(rs, 32, 64);
            l_run[s] = l_run[s] * alpha + rs;
            m_run[s] = m_new;
#pragma unroll
            for (int dt = 0; dt < ND16; ++dt) oacc[s][dt] *= alpha;
        }
        frag_t pf[QSUB][2];
#pragma unroll
        for (int s = 0; s < QSUB; ++s)
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) pf[s][kc] = PackCols<T>::pack(st[s], kc);
        if constexpr (sizeof(T) == 4) {
            // fp32 (parity path): the tile's P.V goes into a fresh accumulator that is then ADDED to the running one -- an fp32
            // MFMA chain is a sequential fmaf chain, and o = sum over ALL keys in one chain carries ~3x the rounding error of a
            // blocked sum; the explicit P.V rule o/(o + 1e-6) amplifies exactly that error next to its pole
#pragma unroll
            for (int dt = 0; dt < ND16; ++dt) {
                f32x4 part[QSUB];
#pragma unroll
                for (int s = 0; s < QSUB; ++s) part[s] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kc = 0; kc < 2; ++kc) {
                    const frag_t vf = tr_frag<T>(sV, dt, kc, lane);
#pragma unroll
                    for (int s = 0; s < QSUB; ++s) part[s] = Mma16<T>::mma(vf, pf[s][kc], part[s]);
                }
#pragma unroll
                for (int s = 0; s < QSUB; ++s) oacc[s][dt] += part[s];
            }
        } else {
#pragma unroll
            for (int dt = 0; dt < ND16; ++dt)
#pragma unroll
                for (int kc = 0; kc < 2; ++kc) {
                    const frag_t vf = tr_frag<T>(sV, dt, kc, lane);
#pragma unroll
                    for (int s = 0; s < QSUB; ++s) oacc[s][dt] = Mma16<T>::mma(vf, pf[s][kc], oacc[s][dt]);
                }
        }
        __syncthreads();
        cur ^= 1;
    }
    T* ob = o + (int64_t)b * S * ldo + (int64_t)h * D;
#pragma unroll
    for (int s = 0; s < QSUB; ++s) {
        const int qi = qw + s * 16 + (lane & 15);
        const float inv = (l_run[s] > 0.f) ? 1.f / l_run[s] : 0.f;
        store_rows<T, D>(ob, ldo, qi, S, oacc[s], inv, lane);
        if (g == 0 && qi < S) lse[((int64_t)b * Hq + h) * S + qi] = m_run[s] * scale + __logf(l_run[s]);
    }
}

// ---- v2 dQ: 8 waves x (16*QS) queries, K + V + K^T tiles, two LDS stages ---------------------------------
// (16 rows per wave: every column-side fragment read feeds ONE MFMA, the kernel is LDS-bandwidth-bound -- the bf16 / d = 128
// shapes that matter run on the 32x32x16 kernels of attention32.hip instead; this form serves fp32 and the other head dims)
template <typename T, int D, bool EXPL>
__global__ __launch_bounds__(512, 2) void attn_bwd_dq_v2_kernel(
    const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v, const T* __restrict__ kt,
    const T* __restrict__ gho, const float* __restrict__ lse, const float* __restrict__ Dd, T* __restrict__ dq,
    int S, int Hq, int Hkv, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldt, int64_t ldg, int64_t lddq,
    float scale, float eps_mask, float eps_qk, int causal, int window, int B, int q_begin,
    const int* __restrict__ row_lo, const int* __restrict__ row_hi) {
    typedef AT<T> A;
    typedef typename Mma16<T>::frag frag_t;
    constexpr int SZ = A::SZ, CT = A::CT, NC16 = A::NC16, NDC = D * SZ / 64, ND16 = D / 16;
    constexpr int QS = 1;                                  // 16-row sub-tiles per wave
    constexpr int NW = 8, BQ = NW * 16 * QS;
    constexpr int TILE = 128 * D, STAGE = 3 * TILE;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rep = Hq / Hkv, nqb = (S + BQ - 1) / BQ;
    int bhk, item;
    if (!xcd_group_decode(blockIdx.x, B * Hkv, rep * nqb, bhk, item)) return;
    const int b = bhk / Hkv, hk = bhk % Hkv, h = hk * rep + item % rep;
    const int qblk = nqb - 1 - item / rep;
    const int q0 = qblk * BQ, qw = q0 + wave * 16 * QS;
    if (q0 + BQ <= q_begin) return;
    const T* kb = k + (int64_t)b * S * ldk + (int64_t)hk * D;
    const T* vb = v + (int64_t)b * S * ldv + (int64_t)hk * D;
    const T* ktb = kt + ((int64_t)b * Hkv + hk) * D * ldt;

    frag_t qf[QS][NDC], gf[QS][NDC];
    float lse2_q[QS], D_q[QS];
    int ivlo[QS], ivhi[QS], qi[QS];
    f32x4 acc[QS][ND16];
#pragma unroll
    for (int s = 0; s < QS; ++s) {
        qi[s] = qw + s * 16 + (lane & 15);
        load_row_frags<T, D>(qf[s], q + (int64_t)b * S * ldq + (int64_t)h * D, ldq, qi[s], S, lane);
        load_row_frags<T, D>(gf[s], gho + (int64_t)b * S * ldg + (int64_t)h * D, ldg, qi[s], S, lane);
        lse2_q[s] = ((qi[s] < S) ? lse[((int64_t)b * Hq + h) * S + qi[s]] : 0.f) * LRP_LOG2E;
        D_q[s] = (qi[s] < S) ? Dd[((int64_t)b * Hq + h) * S + qi[s]] : 0.f;
        ivlo[s] = 0; ivhi[s] = S;
        if (row_lo != nullptr && qi[s] < S) { ivlo[s] = row_lo[(int64_t)b * S + qi[s]]; ivhi[s] = row_hi[(int64_t)b * S + qi[s]]; }
#pragma unroll
        for (int dt = 0; dt < ND16; ++dt) acc[s][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float c1 = scale * LRP_LOG2E;
    int kend = S;
    if (causal) kend = min(S, q0 + BQ);
    int kbeg = 0;
    if (window > 0) { kbeg = q0 - window + 1; kbeg = kbeg < 0 ? 0 : (kbeg / CT) * CT; }

    auto stage = [&](int kt0, int buf) {
        char* sb = smem + buf * STAGE;
        glds_rowmajor<T, D, NW>(kb, ldk, kt0, S, sb, wave, lane);
        glds_rowmajor<T, D, NW>(vb, ldv, kt0, S, sb + TILE, wave, lane);
        glds_transposed<T, D, NW>(ktb, ldt, kt0, sb + 2 * TILE, wave, lane);
    };
    if (kbeg < kend) stage(kbeg, 0);
    __syncthreads();
    int cur = 0;
    for (int kt0 = kbeg; kt0 < kend; kt0 += CT) {
        if (kt0 + CT < kend) stage(kt0 + CT, cur ^ 1);
        const char* sK = smem + cur * STAGE;
        const char* sV = sK + TILE;
        const char* sKt = sK + 2 * TILE;
        f32x4 st[QS][NC16], dp[QS][NC16];
#pragma unroll
        for (int s = 0; s < QS; ++s)
#pragma unroll
            for (int t = 0; t < NC16; ++t) { st[s][t] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[s][t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int t = 0; t < NC16; ++t)
#pragma unroll
            for (int c = 0; c < NDC; ++c) {
                const frag_t kf = rm_frag<T, D>(sK, t, c, lane), vf = rm_frag<T, D>(sV, t, c, lane);
#pragma unroll
                for (int s = 0; s < QS; ++s) {
                    st[s][t] = Mma16<T>::mma(kf, qf[s][c], st[s][t]);
                    dp[s][t] = Mma16<T>::mma(vf, gf[s][c], dp[s][t]);
                }
            }
        frag_t df[QS][2];
#pragma unroll
        for (int s = 0; s < QS; ++s) {
            // interior tiles (every key visible to every row of the sub-tile) skip the per-element mask predicate altogether
            const bool tile_masked = (kt0 + CT > S) || (causal && kt0 + CT - 1 > qw + s * 16) || (window > 0) || (row_lo != nullptr);
            {
#pragma unroll
                for (int t = 0; t < NC16; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float s_raw = st[s][t][r];
                        float p = fast_exp2(__builtin_fmaf(s_raw, c1, -lse2_q[s]));
                        if (tile_masked && !visible(qi[s], kt0 + t * 16 + g * 4 + r, S, causal, window, ivlo[s], ivhi[s])) p = 0.f;
                        if constexpr (EXPL) st[s][t][r] = lrp_ds2<true>(s_raw, p, dp[s][t][r], D_q[s], scale, eps_mask, eps_qk);
                        else st[s][t][r] = p * (dp[s][t][r] - D_q[s]);        // * scale/2 folded into the final store
                    }
            }
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) df[s][kc] = PackCols<T>::pack(st[s], kc);
        }
#pragma unroll
        for (int dt = 0; dt < ND16; ++dt)
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
                const frag_t ktf = tr_frag<T>(sKt, dt, kc, lane);
#pragma unroll
                for (int s = 0; s < QS; ++s) acc[s][dt] = Mma16<T>::mma(ktf, df[s][kc], acc[s][dt]);
            }
        __syncthreads();
        cur ^= 1;
    }
#pragma unroll
    for (int s = 0; s < QS; ++s)
        store_rows<T, D>(dq + (int64_t)b * S * lddq + (int64_t)h * D, lddq, qi[s], S, acc[s], EXPL ? 1.f : 0.5f * scale, lane);
}

// ---- helpers: head transpose and GQA group reduction --------------------------------------------------
template <typename T>
__global__ void transpose_heads_kernel(const T* x, T* xt, int S, int H, int d, int64_t ldx, int64_t ldt) {
    __shared__ T tile[64][65];
    const int bh = blockIdx.z, b = bh / H, h = bh % H;
    const T* in = x + (int64_t)b * S * ldx + (int64_t)h * d;       // [S rows][d cols], row stride ldx
    T* out = xt + (int64_t)bh * d * ldt;                           // [d rows][S cols], row stride ldt
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int rr = ty; rr < 64; rr += 4) {
        const int r = r0 + rr, c = c0 + tx;
        if (r < S && c < d) tile[rr][tx] = in[(int64_t)r * ldx + c];
    }
    __syncthreads();
    for (int cc = ty; cc < 64; cc += 4) {
        const int c = c0 + cc, r = r0 + tx;
        if (r < S && c < d) out[(int64_t)c * ldt + r] = tile[tx][cc];
    }
}

// 16-byte form (bf16, d % 8 == 0, S % 8 == 0, aligned): 64 x 64 tile, 16-byte loads along d, 16-byte stores along S
__global__ __launch_bounds__(256) void transpose_heads_vec_kernel(const bf16_t* x, bf16_t* xt, int S, int H, int d, int64_t ldx, int64_t ldt) {
    __shared__ bf16_t tile[64][66];
    const int bh = blockIdx.z, b = bh / H, h = bh % H;
    const bf16_t* in = x + (int64_t)b * S * ldx + (int64_t)h * d;
    bf16_t* out = xt + (int64_t)bh * d * ldt;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tid = threadIdx.x;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int idx = tid + it * 256, rr = idx >> 3, ch = idx & 7;          // row of the tile, 8-element chunk along d
        const int r = r0 + rr, c = c0 + ch * 8;
        if (r < S && c < d) {
            const Vec16<bf16_t> v = ld16(in + (int64_t)r * ldx + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) tile[rr][ch * 8 + e] = v.v[e];
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int idx = tid + it * 256, cc = idx >> 3, s8 = idx & 7;          // output row (d index), 8-element chunk along S
        const int c = c0 + cc, r = r0 + s8 * 8;
        if (c < d && r < S) {
            Vec16<bf16_t> o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o.v[e] = tile[s8 * 8 + e][cc];
            st16(out + (int64_t)c * ldt + r, o);
        }
    }
}

template <typename T>
__global__ void gqa_reduce_kernel(const T* in, T* out, int64_t rows, int Hkv, int rep, int d, int64_t ld_in, int64_t ld_out) {
    constexpr int EPC = 16 / (int)sizeof(T);
    const int cpr = d / EPC;
    const int64_t total = rows * Hkv * cpr;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cpr) * EPC;
        const int64_t rh = i / cpr;
        const int hk = (int)(rh % Hkv);
        const int64_t r = rh / Hkv;
        float acc[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
        for (int gq = 0; gq < rep; ++gq) {
            const Vec16<T> t = ld16(in + r * ld_in + (int64_t)(hk * rep + gq) * d + c);
#pragma unroll
            for (int e = 0; e < EPC; ++e) acc[e] += t.get(e);
        }
        Vec16<T> o;
#pragma unroll
        for (int e = 0; e < EPC; ++e) o.set(e, acc[e]);
        st16(out + r * ld_out + (int64_t)hk * d + c, o);
    }
}

// group sum of the per-query-head dK + the inverse rotation of RoPE (rotate-half pairs (c, c + d/2), efficient placement: plain transposed
// rotation a1 = g1 cos(c) + g2 sin(c + d/2), a2 = g2 cos(c + d/2) - g1 sin(c), the formula of rope_bwd_kernel with eps = 0) in ONE pass
template <typename T>
__global__ void gqa_reduce_rope_kernel(const T* in, T* out, int64_t rows, int seq, int Hkv, int rep, int d, int64_t ld_in, int64_t ld_out,
                                       const float* __restrict__ cs, const float* __restrict__ sn) {
    constexpr int EPC = 16 / (int)sizeof(T);
    const int hd = d / 2, cpr = hd / EPC;
    const int64_t total = rows * Hkv * cpr;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cpr) * EPC;
        const int64_t rh = i / cpr;
        const int hk = (int)(rh % Hkv);
        const int64_t r = rh / Hkv;
        const int pos = (int)(r % seq);
        float g1[EPC], g2[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) g1[e] = g2[e] = 0.f;
        for (int gq = 0; gq < rep; ++gq) {
            const T* p = in + r * ld_in + (int64_t)(hk * rep + gq) * d + c;
            const Vec16<T> t1 = ld16(p), t2 = ld16(p + hd);
#pragma unroll
            for (int e = 0; e < EPC; ++e) { g1[e] += t1.get(e); g2[e] += t2.get(e); }
        }
        const float* pc = cs + (int64_t)pos * d + c;
        const float* ps = sn + (int64_t)pos * d + c;
        Vec16<T> o1, o2;
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            o1.set(e, g1[e] * pc[e] + g2[e] * ps[e + hd]);
            o2.set(e, g2[e] * pc[e + hd] - g1[e] * ps[e]);
        }
        T* po = out + r * ld_out + (int64_t)hk * d + c;
        st16(po, o1);
        st16(po + hd, o2);
    }
}

template <typename K> void set_lds(K kern, size_t bytes) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// bf16, d in {64, 96, 128}: 32x32x16 MFMA kernels of attention32.hip (no head-transposed operands)
int lrp_attn32_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int B, int S, int Hq, int Hkv, int d, int64_t ldq,
                   int64_t ldk, int64_t ldv, int64_t ldo, float scale, int causal, int window, int q_begin, const int* row_lo,
                   const int* row_hi, hipStream_t st);
int lrp_attn32_dq(const void* q, const void* k, const void* v, const void* gho, const float* lse, const float* D_, void* dq, int B,
                  int S, int Hq, int Hkv, int d, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldg, int64_t lddq, float scale,
                  float eps_mask, float eps_qk, int causal, int window, int q_begin, const int* row_lo, const int* row_hi,
                  hipStream_t st, const void* o = nullptr, int64_t ldo = 0, float* Dout = nullptr, const float* cos_t = nullptr,
                  const float* sin_t = nullptr);
int lrp_attn32_dkv(const void* q, const void* k, const void* v, const void* gho, const float* lse, const float* D_, void* dk,
                   void* dv, int B, int S, int Hq, int Hkv, int d, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldg, int64_t lddk,
                   int64_t lddv, float scale, float eps_mask, float eps_qk, int causal, int window, int q_begin,
                   const int* row_lo, const int* row_hi, hipStream_t st);
int lrp_attn32_fwd_d256(const void* q, const void* k, const void* v, void* o, float* lse, int B, int S, int Hq, int Hkv, int64_t ldq,
                        int64_t ldk, int64_t ldv, int64_t ldo, float scale, int causal, int window, int q_begin, const int* row_lo,
                        const int* row_hi, hipStream_t st);
int lrp_attn32_dq_d256(const void* q, const void* k, const void* v, const void* gho, const float* lse, const float* D_, void* dq, int B,
                       int S, int Hq, int Hkv, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldg, int64_t lddq, float scale,
                       float eps_mask, float eps_qk, int causal, int window, int q_begin, const int* row_lo, const int* row_hi,
                       hipStream_t st);
int lrp_attn32_dkv_d256(const void* q, const void* k, const void* v, const void* gho, const float* lse, const float* D_, void* dk,
                        void* dv, int B, int S, int Hq, int Hkv, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldg, int64_t lddk,
                        int64_t lddv, float scale, float eps_mask, float eps_qk, int causal, int window, int q_begin,
                        const int* row_lo, const int* row_hi, hipStream_t st);
// bf16, d in {64, 96, 128, 256}: the 32x32x16 / transpose-read kernels of attention32.hip (no head-transposed operands)
static inline bool use_attn32(int dtype, int d) { return dtype == LRP_BF16 && (d == 64 || d == 96 || d == 128 || d == 256); }
extern "C" int lrp_attn_needs_transposed(int dtype, int d) { return use_attn32(dtype, d) ? 0 : 1; }

#define ATT_DISPATCH_D(T, d, ...)                                   \
    if ((size_t)d * sizeof(T) < 64) return LRP_ESHAPE;              \
    switch (d) {                                                    \
        case 16: { constexpr int DD = (sizeof(T) == 4) ? 16 : 32; __VA_ARGS__ } break;      \
        case 32: { constexpr int DD = 32; __VA_ARGS__ } break;      \
        case 64: { constexpr int DD = 64; __VA_ARGS__ } break;      \
        case 128: { constexpr int DD = 128; __VA_ARGS__ } break;    \
        case 256: { constexpr int DD = 256; __VA_ARGS__ } break;    \
        default: return LRP_ESHAPE;                                 \
    }

template <typename T>
static int attn_fwd_t(const void* q, const void* k, const void* vt, void* o, float* lse, int B, int S, int Hq, int Hkv,
                      int d, int64_t ldq, int64_t ldk, int64_t ldt, int64_t ldo, float scale, int causal, int window,
                      int q_begin, const int* row_lo, const int* row_hi, hipStream_t st) {
    constexpr int SZ = sizeof(T), CT = 128 / SZ;
    if ((ldt % CT) == 0 && ((size_t)d * SZ >= 64) && d <= 128) {
        ATT_DISPATCH_D(T, d, {
            if constexpr (DD <= 128) {
                const size_t lds = 2 * (2 * (size_t)128 * DD);
                const int rep = Hq / Hkv;
                // 256-query workgroups (two 16-row sub-tiles per wave: every K/V fragment read feeds two
                // MFMAs) once that still leaves >= 2 workgroups per CU, else 128-query workgroups
                if ((int64_t)B * Hq * ((S + 255) / 256) >= 512) {
                    auto kern = attn_fwd_v2_kernel<T, DD, 2>;
                    set_lds(kern, lds);
                    dim3 grid(xcd_group_grid(B * Hkv, rep * ((S + 255) / 256)));
                    hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, (const T*)q, (const T*)k, (const T*)vt, (T*)o, lse, S, Hq,
                                       Hkv, ldq, ldk, ldt, ldo, scale, causal, window, B, q_begin, row_lo, row_hi);
                } else {
                    auto kern = attn_fwd_v2_kernel<T, DD, 1>;
                    set_lds(kern, lds);
                    dim3 grid(xcd_group_grid(B * Hkv, rep * ((S + 127) / 128)));
                    hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, (const T*)q, (const T*)k, (const T*)vt, (T*)o, lse, S, Hq,
                                       Hkv, ldq, ldk, ldt, ldo, scale, causal, window, B, q_begin, row_lo, row_hi);
                }
            }
        })
        return lrp_check_launch();
    }
    ATT_DISPATCH_D(T, d, {
        constexpr int QSUB = (DD >= 256) ? 1 : 2;
        const size_t lds = (size_t)CT * DD * SZ + (size_t)DD * 128;
        auto kern = attn_fwd_kernel<T, DD, QSUB>;
        set_lds(kern, lds);
        const int BQ = 64 * QSUB;
        dim3 grid((S + BQ - 1) / BQ, Hq, B);
        hipLaunchKernelGGL(kern, grid, dim3(ANT), lds, st, (const T*)q, (const T*)k, (const T*)vt, (T*)o, lse, S, Hq, Hkv,
                           ldq, ldk, ldt, ldo, scale, causal, window, q_begin, row_lo, row_hi);
    })
    return lrp_check_launch();
}

static int attn_common_check(int B, int S, int Hq, int Hkv, int d, int dtype, float scale = 1.f) {
    if (!(scale > 0.f)) return LRP_EINVAL;          // the running max is tracked on raw scores
    if (B < 0 || S < 0 || Hq < 1 || Hkv < 1 || (Hq % Hkv) || d < 16) return LRP_EINVAL;
    if (dtype != LRP_F32 && dtype != LRP_BF16) return LRP_EINVAL;
    if (B > 65535 || Hq > 65535) return LRP_ESHAPE;
    return LRP_OK;
}

extern "C" int lrp_attn_fwd(const void* q, const void* k, const void* v, const void* v_t, void* o, float* lse, int B, int S,
                            int Hq, int Hkv, int d, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldt, int64_t ldo, float scale,
                            int causal, int window, int q_begin, const int* row_lo, const int* row_hi, int dtype, void* stream) {
    if ((row_lo == nullptr) != (row_hi == nullptr)) return LRP_EINVAL;
    if (!q || !k || !o || !lse) return LRP_EINVAL;
    int rc = attn_common_check(B, S, Hq, Hkv, d, dtype, scale);
    if (rc) return rc;
    if (B == 0 || S == 0) return LRP_OK;
    const int epc = dtype == LRP_F32 ? 4 : 8;
    hipStream_t st = (hipStream_t)stream;
    if (use_attn32(dtype, d)) {
        if (!v) return LRP_EINVAL;
        if (!al16(q) || !al16(k) || !al16(v) || !al16(o) || (ldq % epc) || (ldk % epc) || (ldv % epc) || (ldo % 4)) return LRP_EALIGN;
        if (d == 256) return lrp_attn32_fwd_d256(q, k, v, o, lse, B, S, Hq, Hkv, ldq, ldk, ldv, ldo, scale, causal, window, q_begin, row_lo, row_hi, st);
        return lrp_attn32_fwd(q, k, v, o, lse, B, S, Hq, Hkv, d, ldq, ldk, ldv, ldo, scale, causal, window, q_begin, row_lo, row_hi, st);
    }
    if (!v_t) return LRP_EINVAL;
    if (!al16(q) || !al16(k) || !al16(v_t) || !al16(o) || (ldq % epc) || (ldk % epc) || (ldt % epc) || (ldo % 4) || ldt < S) return LRP_EALIGN;
    if (dtype == LRP_F32) return attn_fwd_t<float>(q, k, v_t, o, lse, B, S, Hq, Hkv, d, ldq, ldk, ldt, ldo, scale, causal, window, q_begin, row_lo, row_hi, st);
    return attn_fwd_t<bf16_t>(q, k, v_t, o, lse, B, S, Hq, Hkv, d, ldq, ldk, ldt, ldo, scale, causal, window, q_begin, row_lo, row_hi, st);
}

template <typename T>
static int attn_dq_t(const void* q, const void* k, const void* v, const void* kt, const void* gho, const float* lse,
                     const float* D, void* dq, int B, int S, int Hq, int Hkv, int d, int64_t ldq, int64_t ldk, int64_t ldv,
                     int64_t ldt, int64_t ldg, int64_t lddq, float scale, float eps_mask, float eps_qk, int causal, int window,
                     int q_begin, const int* row_lo, const int* row_hi, hipStream_t st) {
    constexpr int SZ = sizeof(T), CT = 128 / SZ;
    if ((ldt % CT) == 0 && ((size_t)d * SZ >= 64) && d <= 128) {
        ATT_DISPATCH_D(T, d, {
            if constexpr (DD <= 128) {
                const size_t lds = 2 * (3 * (size_t)128 * DD);
                dim3 grid(xcd_group_grid(B * Hkv, (Hq / Hkv) * ((S + 127) / 128)));
                if (eps_mask != 0.f || eps_qk != 0.f) {
                    auto kern = attn_bwd_dq_v2_kernel<T, DD, true>;
                    set_lds(kern, lds);
                    hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, (const T*)q, (const T*)k, (const T*)v, (const T*)kt,
                                       (const T*)gho, lse, D, (T*)dq, S, Hq, Hkv, ldq, ldk, ldv, ldt, ldg, lddq, scale, eps_mask,
                                       eps_qk, causal, window, B, q_begin, row_lo, row_hi);
                } else {
                    auto kern = attn_bwd_dq_v2_kernel<T, DD, false>;
                    set_lds(kern, lds);
                    hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, (const T*)q, (const T*)k, (const T*)v, (const T*)kt,
                                       (const T*)gho, lse, D, (T*)dq, S, Hq, Hkv, ldq, ldk, ldv, ldt, ldg, lddq, scale, eps_mask,
                                       eps_qk, causal, window, B, q_begin, row_lo, row_hi);
                }
            }
        })
        return lrp_check_launch();
    }
    ATT_DISPATCH_D(T, d, {
        const size_t lds = 2 * (size_t)CT * DD * SZ + (size_t)DD * 128;
        auto kern = attn_bwd_dq_kernel<T, DD>;
        set_lds(kern, lds);
        dim3 grid((S + 63) / 64, Hq, B);
        hipLaunchKernelGGL(kern, grid, dim3(ANT), lds, st, (const T*)q, (const T*)k, (const T*)v, (const T*)kt, (const T*)gho,
                           lse, D, (T*)dq, S, Hq, Hkv, ldq, ldk, ldv, ldt, ldg, lddq, scale, eps_mask, eps_qk, causal, window, q_begin, row_lo, row_hi);
    })
    return lrp_check_launch();
}

extern "C" int lrp_attn_bwd_dq(const void* q, const void* k, const void* v, const void* k_t, const void* Gho,
                               const float* lse, const float* D, void* dq, int B, int S, int Hq, int Hkv, int d,
                               int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldt, int64_t ldgho, int64_t lddq, float scale,
                               float eps_mask, float eps_qk, int causal, int window, int q_begin, const int* row_lo, const int* row_hi, int dtype, void* stream) {
    if ((row_lo == nullptr) != (row_hi == nullptr)) return LRP_EINVAL;
    if (!q || !k || !v || !Gho || !lse || !D || !dq) return LRP_EINVAL;
    int rc = attn_common_check(B, S, Hq, Hkv, d, dtype, scale);
    if (rc) return rc;
    if (B == 0 || S == 0) return LRP_OK;
    const int epc = dtype == LRP_F32 ? 4 : 8;
    if (!al16(q) || !al16(k) || !al16(v) || !al16(Gho) || !al16(dq) || (ldq % epc) || (ldk % epc) ||
        (ldv % epc) || (ldgho % epc) || (lddq % 4)) return LRP_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    if (use_attn32(dtype, d) && d == 256)
        return lrp_attn32_dq_d256(q, k, v, Gho, lse, D, dq, B, S, Hq, Hkv, ldq, ldk, ldv, ldgho, lddq, scale, eps_mask, eps_qk, causal, window, q_begin, row_lo, row_hi, st);
    if (use_attn32(dtype, d))
        return lrp_attn32_dq(q, k, v, Gho, lse, D, dq, B, S, Hq, Hkv, d, ldq, ldk, ldv, ldgho, lddq, scale, eps_mask, eps_qk, causal, window, q_begin, row_lo, row_hi, st);
    if (!k_t) return LRP_EINVAL;
    if (!al16(k_t) || (ldt % epc) || ldt < S) return LRP_EALIGN;
    if (dtype == LRP_F32) return attn_dq_t<float>(q, k, v, k_t, Gho, lse, D, dq, B, S, Hq, Hkv, d, ldq, ldk, ldv, ldt, ldgho, lddq, scale, eps_mask, eps_qk, causal, window, q_begin, row_lo, row_hi, st);
    return attn_dq_t<bf16_t>(q, k, v, k_t, Gho, lse, D, dq, B, S, Hq, Hkv, d, ldq, ldk, ldv, ldt, ldgho, lddq, scale, eps_mask, eps_qk, causal, window, q_begin, row_lo, row_hi, st);
}

// dQ with the statistic D_i = sum_d Gho_i o_i formed in the kernel's prologue (lxt.efficient placement: no stabiliser on P.V or the scores, so
// Gho is the o-projection's dgrad times the uniform rule's 1/2 and needs no pass of its own): D is an OUTPUT here, for the dK / dV kernel
extern "C" int lrp_attn_bwd_dq_d_ok(int dtype, int d) { return (use_attn32(dtype, d) && d != 256) ? 1 : 0; }

extern "C" int lrp_attn_bwd_dq_d(const void* q, const void* k, const void* v, const void* Gho, const void* o, const float* lse, float* D,
                                 void* dq, int B, int S, int Hq, int Hkv, int d, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldgho,
                                 int64_t ldo, int64_t lddq, float scale, int causal, int window, const int* row_lo, const int* row_hi,
                                 const float* cos_t, const float* sin_t, int dtype, void* stream) {
    if ((cos_t == nullptr) != (sin_t == nullptr)) return LRP_EINVAL;
    if (cos_t && d != 64 && d != 128) return LRP_ESHAPE;                // the rotation's partner column d/2 away must be a whole 32-column block
    if ((row_lo == nullptr) != (row_hi == nullptr)) return LRP_EINVAL;
    if (!q || !k || !v || !Gho || !o || !lse || !D || !dq) return LRP_EINVAL;
    int rc = attn_common_check(B, S, Hq, Hkv, d, dtype, scale);
    if (rc) return rc;
    if (!lrp_attn_bwd_dq_d_ok(dtype, d)) return LRP_ESHAPE;
    if (B == 0 || S == 0) return LRP_OK;
    if (!al16(q) || !al16(k) || !al16(v) || !al16(Gho) || !al16(o) || !al16(dq) || (ldq % 8) || (ldk % 8) || (ldv % 8) || (ldgho % 8) ||
        (ldo % 8) || (lddq % 4)) return LRP_EALIGN;
    return lrp_attn32_dq(q, k, v, Gho, lse, nullptr, dq, B, S, Hq, Hkv, d, ldq, ldk, ldv, ldgho, lddq, scale, 0.f, 0.f, causal, window, 0,
                         row_lo, row_hi, (hipStream_t)stream, o, ldo, D, cos_t, sin_t);
}

template <typename T>
static int attn_dkv_t(const void* q, const void* k, const void* v, const void* qt, const void* gho, const void* ghot,
                      const float* lse, const float* D, void* dk, void* dv, int B, int S, int Hq, int Hkv, int d, int64_t ldq,
                      int64_t ldk, int64_t ldv, int64_t ldt, int64_t ldg, int64_t lddk, int64_t lddv, float scale,
                      float eps_mask, float eps_qk, int causal, int window, int q_begin, const int* row_lo, const int* row_hi, hipStream_t st) {
    constexpr int SZ = sizeof(T), CT = 128 / SZ;
    if ((ldt % CT) == 0 && ((size_t)d * SZ >= 64) && d <= 128 && S >= 1) {
        ATT_DISPATCH_D(T, d, {
            if constexpr (DD <= 128) {
                const size_t lds = 2 * (4 * (size_t)128 * DD + 512);
                dim3 grid(xcd_group_grid(B * Hq, (S + 127) / 128));
                if (eps_mask != 0.f || eps_qk != 0.f) {
                    auto kern = attn_bwd_dkv_v2_kernel<T, DD, true>;
                    set_lds(kern, lds);
                    hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, (const T*)q, (const T*)k, (const T*)v, (const T*)qt,
                                       (const T*)gho, (const T*)ghot, lse, D, (T*)dk, (T*)dv, S, Hq, Hkv, ldq, ldk, ldv, ldt,
                                       ldg, lddk, lddv, scale, eps_mask, eps_qk, causal, window, B, q_begin, row_lo, row_hi);
                } else {
                    auto kern = attn_bwd_dkv_v2_kernel<T, DD, false>;
                    set_lds(kern, lds);
                    hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, (const T*)q, (const T*)k, (const T*)v, (const T*)qt,
                                       (const T*)gho, (const T*)ghot, lse, D, (T*)dk, (T*)dv, S, Hq, Hkv, ldq, ldk, ldv, ldt,
                                       ldg, lddk, lddv, scale, eps_mask, eps_qk, causal, window, B, q_begin, row_lo, row_hi);
                }
            }
        })
        return lrp_check_launch();
    }
    ATT_DISPATCH_D(T, d, {
        const size_t lds = 2 * (size_t)CT * DD * SZ + 2 * (size_t)DD * 128;
        auto kern = attn_bwd_dkv_kernel<T, DD>;
        set_lds(kern, lds);
        dim3 grid((S + 63) / 64, Hq, B);
        hipLaunchKernelGGL(kern, grid, dim3(ANT), lds, st, (const T*)q, (const T*)k, (const T*)v, (const T*)qt, (const T*)gho,
                           (const T*)ghot, lse, D, (T*)dk, (T*)dv, S, Hq, Hkv, ldq, ldk, ldv, ldt, ldg, lddk, lddv, scale,
                           eps_mask, eps_qk, causal, window, q_begin, row_lo, row_hi);
    })
    return lrp_check_launch();
}

extern "C" int lrp_attn_bwd_dkv(const void* q, const void* k, const void* v, const void* q_t, const void* Gho,
                                const void* Gho_t, const float* lse, const float* D, void* dk_h, void* dv_h, int B, int S,
                                int Hq, int Hkv, int d, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldt, int64_t ldgho,
                                int64_t lddk, int64_t lddv, float scale, float eps_mask, float eps_qk, int causal, int window,
                                int q_begin, const int* row_lo, const int* row_hi, int dtype, void* stream) {
    if ((row_lo == nullptr) != (row_hi == nullptr)) return LRP_EINVAL;
    if (!q || !k || !v || !Gho || !lse || !D || !dk_h || !dv_h) return LRP_EINVAL;
    int rc = attn_common_check(B, S, Hq, Hkv, d, dtype, scale);
    if (rc) return rc;
    if (B == 0 || S == 0) return LRP_OK;
    const int epc = dtype == LRP_F32 ? 4 : 8;
    if (!al16(q) || !al16(k) || !al16(v) || !al16(Gho) || !al16(dk_h) || !al16(dv_h) ||
        (ldq % epc) || (ldk % epc) || (ldv % epc) || (ldgho % epc) || (lddk % 4) || (lddv % 4)) return LRP_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    if (use_attn32(dtype, d) && d == 256)
        return lrp_attn32_dkv_d256(q, k, v, Gho, lse, D, dk_h, dv_h, B, S, Hq, Hkv, ldq, ldk, ldv, ldgho, lddk, lddv, scale, eps_mask, eps_qk, causal, window, q_begin, row_lo, row_hi, st);
    if (use_attn32(dtype, d))
        return lrp_attn32_dkv(q, k, v, Gho, lse, D, dk_h, dv_h, B, S, Hq, Hkv, d, ldq, ldk, ldv, ldgho, lddk, lddv, scale, eps_mask, eps_qk, causal, window, q_begin, row_lo, row_hi, st);
    if (!q_t || !Gho_t) return LRP_EINVAL;
    if (!al16(q_t) || !al16(Gho_t) || (ldt % epc) || ldt < S) return LRP_EALIGN;
    if (dtype == LRP_F32) return attn_dkv_t<float>(q, k, v, q_t, Gho, Gho_t, lse, D, dk_h, dv_h, B, S, Hq, Hkv, d, ldq, ldk, ldv, ldt, ldgho, lddk, lddv, scale, eps_mask, eps_qk, causal, window, q_begin, row_lo, row_hi, st);
    return attn_dkv_t<bf16_t>(q, k, v, q_t, Gho, Gho_t, lse, D, dk_h, dv_h, B, S, Hq, Hkv, d, ldq, ldk, ldv, ldt, ldgho, lddk, lddv, scale, eps_mask, eps_qk, causal, window, q_begin, row_lo, row_hi, st);
}

extern "C" int lrp_transpose_heads(const void* x, void* xt, int B, int S, int H, int d, int64_t ldx, int64_t ldt,
                                   int dtype, void* stream) {
    if (!x || !xt || B < 0 || S < 0 || H < 1 || d < 1 || ldt < S) return LRP_EINVAL;
    if (B == 0 || S == 0) return LRP_OK;
    if ((int64_t)B * H > 65535 || (S + 63) / 64 > 65535) return LRP_ESHAPE;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((d + 63) / 64, (S + 63) / 64, B * H), block(256);
    const bool vec = dtype == LRP_BF16 && (d % 8) == 0 && (S % 8) == 0 && (ldx % 8) == 0 && (ldt % 8) == 0 && al16(x) && al16(xt);
    if (vec) hipLaunchKernelGGL(transpose_heads_vec_kernel, grid, block, 0, st, (const bf16_t*)x, (bf16_t*)xt, S, H, d, ldx, ldt);
    else if (dtype == LRP_F32) hipLaunchKernelGGL((transpose_heads_kernel<float>), grid, block, 0, st, (const float*)x, (float*)xt, S, H, d, ldx, ldt);
    else if (dtype == LRP_BF16) hipLaunchKernelGGL((transpose_heads_kernel<bf16_t>), grid, block, 0, st, (const bf16_t*)x, (bf16_t*)xt, S, H, d, ldx, ldt);
    else return LRP_EINVAL;
    return lrp_check_launch();
}

extern "C" int lrp_gqa_reduce(const void* in, void* out, int64_t rows, int Hkv, int rep, int d, int64_t ld_in,
                              int64_t ld_out, int dtype, void* stream) {
    if (!in || !out || rows < 0 || Hkv < 1 || rep < 1 || d < 1) return LRP_EINVAL;
    if (rows == 0) return LRP_OK;
    const int epc = dtype == LRP_F32 ? 4 : 8;
    if (dtype != LRP_F32 && dtype != LRP_BF16) return LRP_EINVAL;
    if (!al16(in) || !al16(out) || (d % epc) || (ld_in % epc) || (ld_out % epc)) return LRP_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    const int64_t work = rows * Hkv * (d / epc);
    int64_t nb = (work + 255) / 256;
    if (nb > 2048) nb = 2048;
    if (dtype == LRP_F32) hipLaunchKernelGGL((gqa_reduce_kernel<float>), dim3((unsigned)nb), dim3(256), 0, st, (const float*)in, (float*)out, rows, Hkv, rep, d, ld_in, ld_out);
    else hipLaunchKernelGGL((gqa_reduce_kernel<bf16_t>), dim3((unsigned)nb), dim3(256), 0, st, (const bf16_t*)in, (bf16_t*)out, rows, Hkv, rep, d, ld_in, ld_out);
    return lrp_check_launch();
}

extern "C" int lrp_gqa_reduce_rope(const void* in, void* out, int64_t rows, int seq, int Hkv, int rep, int d, int64_t ld_in, int64_t ld_out,
                                   const float* cos_t, const float* sin_t, int dtype, void* stream) {
    if (!in || !out || !cos_t || !sin_t || rows < 0 || seq < 1 || Hkv < 1 || rep < 1 || d < 2 || (d & 1)) return LRP_EINVAL;
    if (rows == 0) return LRP_OK;
    if (dtype != LRP_F32 && dtype != LRP_BF16) return LRP_EINVAL;
    const int epc = dtype == LRP_F32 ? 4 : 8;
    if (!al16(in) || !al16(out) || ((d / 2) % epc) || (ld_in % epc) || (ld_out % epc)) return LRP_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    const int64_t work = rows * Hkv * (d / 2 / epc);
    int64_t nb = (work + 255) / 256;
    if (nb > 2048) nb = 2048;
    if (dtype == LRP_F32) hipLaunchKernelGGL((gqa_reduce_rope_kernel<float>), dim3((unsigned)nb), dim3(256), 0, st, (const float*)in, (float*)out, rows, seq, Hkv, rep, d, ld_in, ld_out, cos_t, sin_t);
    else hipLaunchKernelGGL((gqa_reduce_rope_kernel<bf16_t>), dim3((unsigned)nb), dim3(256), 0, st, (const bf16_t*)in, (bf16_t*)out, rows, seq, Hkv, rep, d, ld_in, ld_out, cos_t, sin_t);
    return lrp_check_launch();
}
