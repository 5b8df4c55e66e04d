// attention32.hip -- K4 for the headline shape (bf16, head_dim 128): forward, dQ and dK/dV on v_mfma_f32_32x32x16_bf16.
//
// Same orientation trick as attention.hip (scores are computed TRANSPOSED w.r.t. the register-resident "row side", so a
// row's softmax statistics are per-lane scalars and P / dS in the accumulator registers ARE the next MFMA's operand), with
// three differences that the instruction mix of the 16x16 kernels asked for (profiles/r01):
//   * 32 row-side rows per wave on 32x32x16 MFMAs: one 16-byte LDS fragment feeds 2x the FLOPs of a 16x16x32 fragment
//     (the 16-row kernels were LDS-bandwidth-bound: one ds_read_b128 per MFMA);
//   * the transposed operand of the third contraction (K^T for dQ, Q^T / Gho^T for dK / dV, V^T for O) is read straight out
//     of the ROW-MAJOR tile with ds_read_b64_tr_b16 -- no head-transposed copies in HBM, no transposed tiles in LDS (two
//     tiles per stage instead of three / four), no lrp_transpose_heads launches in front of these kernels.  The contraction
//     slot order of the register-resident operand (accumulator register r of lane-half hi holds column-side row
//     (r&3) + 8 (r>>2) + 4 hi) is matched by WHICH four rows each transpose read fetches;
//   * every LDS address is a per-lane base register (hoisted out of the tile loop) + an immediate, and a 32-row block that is
//     entirely visible / entirely masked takes a wave-uniform branch: no per-element mask arithmetic on interior tiles.
// LDS tile: [64 rows][256 B], 16-byte chunk c of row r stored at chunk c ^ rot(r), rot(r) = ((r&3)<<2)|((r>>2)&3): the
// ds_read_b128 row fragments (16 lanes = 16 distinct rows) and the transpose reads (4 consecutive rows x 64 B per 32 lanes)
// are both conflict-free under it.  Direct-to-LDS staging (global_load_lds_dwordx4), swizzle on the source address, two
// stages, one barrier per 64-row tile.
#include "common.hpp"

namespace attn32 {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_t;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

constexpr int NW = 8;              // waves per workgroup of the dK/dV kernel (256 keys)
#ifndef A32_NWQ
#define A32_NWQ 4
#endif
constexpr int NWQ = A32_NWQ;       // waves per workgroup of the forward / dQ kernels: 128 queries, TWO workgroups per CU, so that
                                   // one workgroup's prologue / store tail and barrier waits run under the other's MFMAs
constexpr int CT = 64;             // column-side rows per tile
constexpr int KP = 256;            // bytes per LDS tile row: the head-dim-128 image for EVERY head dim DH in {64, 96, 128} (16-byte chunks
                                   // >= DH / 8 of a row are never staged and never read: one swizzle, one set of lane addresses)
constexpr int TILE = CT * KP;      // 16 KiB
constexpr int NKX = 8, NDX = 4;    // array bounds (DH = 128); an instantiation for DH runs NK = DH / 16 MFMA k-steps over the head dim and
                                   // ND32 = DH / 32 32-wide head-dim blocks per out^T accumulator
#define LRP_LOG2E 1.4426950408889634f

LRP_DEVICE int rot4(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }
LRP_DEVICE float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

LRP_DEVICE bool xcd_group_decode(int L, int ngroups, int per_group, int& group, int& item) {
    const int xcd = L & 7, i = L >> 3;
    item = i % per_group;
    group = xcd + 8 * (i / per_group);
    return group < ngroups;
}
inline int xcd_group_grid(int ngroups, int per_group) { return ((ngroups + 7) / 8) * 8 * per_group; }
// the same grid walked ITEM-major within a XCD: all groups' item 0 first, then item 1, ... -- for the dK / dV kernel, whose items (key blocks of a
// head) differ 8 : 1 in work (causal) and whose 128-KiB workgroups run ONE per CU: with the group-major order the last head's heaviest key
// block starts when the other CUs are already draining (list-scheduling simulation of 16 heads x 8 key blocks on a XCD's 32 CUs: makespan 88 tile
// steps against an ideal of 72; item-major -- heaviest first over ALL heads -- reaches 72).  Price: a head's key blocks no longer run at the same
// time on one XCD, so its Q / Gho tiles are re-read from the Infinity Cache instead of the XCD's L2.
LRP_DEVICE bool xcd_item_major_decode(int L, int ngroups, int per_group, int& group, int& item) {
    const int xcd = L & 7, i = L >> 3, gpx = (ngroups + 7) >> 3;
    item = i / gpx;
    group = xcd + 8 * (i % gpx);
    return group < ngroups;
}

// stage a [64 rows][256 B] tile of a token-major operand: 16 one-KiB groups of 4 rows, 16 / NWV per wave.
// A32_BUFFER_STAGING (default): buffer_load_dwordx4 .. lds -- ONE per-lane byte offset (row (l >> 4) of the group, swizzled chunk; the
// group index only enters the chunk through grp & 3 = wave & 3), the group's first row in the scalar offset, rows >= S read as zero
// through num_records (a zero K / V / Q / Gho row contributes nothing: masked keys, and dS^T Q = P^T Gho = 0 for a zero query row).
// A global_load_lds piece with its 64-bit per-lane address costs ~80-120 cycles at issue, a buffer piece ~35 (profiles/r03_gemm_experiments.txt).
#ifndef A32_BUFFER_STAGING
#define A32_BUFFER_STAGING 1
#endif
// dK / dV kernel (d <= 128) build switches, kept for A/B builds (tools/ab): A32_DKV_P1 = fragment read-ahead of the S^T / dP^T phase in
// steps (0: the round 2-4 form), A32_DKV_EW = 1: one mask branch per 4-query group in the element-wise phase
#ifndef A32_DKV_P1
#define A32_DKV_P1 2
#endif
#ifndef A32_DKV_EW
#define A32_DKV_EW 1
#endif
// A32_DKV_ORDER = 1: workgroups walk the (head, key block) grid key-block-major within a XCD (heaviest key blocks of ALL heads first)
#ifndef A32_DKV_ORDER
#define A32_DKV_ORDER 1
#endif

// Head dims below 128: a lane whose source chunk lies past the head (chunk >= DH / 8) gets an offset beyond num_records -- the buffer
// unit returns zero without a memory request; its LDS slot is never read.
constexpr int A32_OOB = 0x40000000;
template <int NWV = NW, int DH = 128>
LRP_DEVICE void stage_tile(const bf16_t* base, int64_t ld, int row0, int S, char* lds, int wave, int lane) {
#if A32_BUFFER_STAGING
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(((int64_t)(S - 1) * ld + DH) * 2), 0x00020000);
    const int rl = lane >> 4, slot = lane & 15;
    const int chunk_ = slot ^ ((rl << 2) | (wave & 3));
    const int voff = (DH == 128 || chunk_ < DH / 8) ? (int)(rl * ld * 2) + (chunk_ << 4) : A32_OOB;
#pragma unroll
    for (int g = 0; g < 16 / NWV; ++g) {
        const int grp = g * NWV + wave;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(lds + grp * 1024), 16, voff, (int)((int64_t)(row0 + grp * 4) * ld * 2), 0, 0);
    }
#else
#pragma unroll
    for (int g = 0; g < 16 / NWV; ++g) {
        const int grp = g * NWV + wave;
        const int row = grp * 4 + (lane >> 4), slot = lane & 15;
        const int chunk = slot ^ rot4(row);
        int gr = row0 + row;
        gr = gr < S ? gr : S - 1;
        if (chunk * 8 < DH)
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(base + (int64_t)gr * ld + chunk * 8), (lds_ptr_t)(lds + grp * 1024), 16, 0, 0);
    }
#endif
}
// 64 fp32 row statistics -> lds[0..63]
LRP_DEVICE void stage_stats(const float* base, int r0, int S, char* lds, int lane) {
    int r = r0 + lane;
    r = r < S ? r : S - 1;
    __builtin_amdgcn_global_load_lds((glb_ptr_t)(base + r), (lds_ptr_t)lds, 4, 0, 0);
}

// per-lane LDS byte offsets (relative to a tile base), hoisted out of the tile loops
struct LaneAddr {
    uint32_t rm[NKX];         // row fragment: row (lane & 31) of a 32-row block, head-dim chunk 2 ks + hi
    uint32_t tr[NDX][2];      // transpose read: head-dim block db, half (rows +0..3 / +8..11 of a 16-row group, + 4 hi)
    LRP_DEVICE void init(int lane) {
        const int l31 = lane & 31, hi = lane >> 5, i16 = lane & 15;
#pragma unroll
        for (int ks = 0; ks < NKX; ++ks) rm[ks] = l31 * KP + (((ks * 2 + hi) ^ rot4(l31)) << 4);
#pragma unroll
        for (int db = 0; db < NDX; ++db)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int r = half * 8 + 4 * hi + (i16 >> 2);                       // row within its 16-row group
                const int chunk = db * 4 + ((l31 >> 4) << 1) + ((i16 & 3) >> 1);
                tr[db][half] = r * KP + ((chunk ^ rot4(r)) << 4) + 8 * (i16 & 1);
            }
    }
};

LRP_DEVICE bf16x8 lds_frag(const char* tile, uint32_t off) { return *reinterpret_cast<const bf16x8*>(tile + off); }
// 8 contraction slots of one head-dim column: rows {4hi..4hi+3} and {8+4hi..8+4hi+3} of the 16-row group at `tile`
LRP_DEVICE bf16x8 lds_frag_tr(const char* tile, uint32_t off0, uint32_t off1) {
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(tile + off0));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(tile + off1));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, v);
}
LRP_DEVICE f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
LRP_DEVICE f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}
// accumulator registers 8j .. 8j+7 as the next MFMA's operand
LRP_DEVICE bf16x8 pack8(const f32x16& x, int j) {
    bf16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (bf16_t)x[8 * j + e];
    return r;
}
// column-side row of accumulator register r in lane-half hi
LRP_DEVICE int crow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// ---- hand-issued LDS reads ---------------------------------------------------------------------------------------------
// The compiler's own bookkeeping puts s_waitcnt lgkmcnt(0) behind every fresh fragment read, i.e. {read, wait the full LDS
// latency, MFMA} per contraction step.  The LDS returns in order, so "the fragment I need" = "all but the n younger reads":
// reads are issued a few steps ahead as inline asm and waited for with hand-counted s_waitcnt lgkmcnt(n); sched_barrier(0)
// keeps the groups in program order.  (Outstanding scalar loads or compiler-issued LDS operations can only make a counted
// wait more conservative.)  A side effect: the compiler no longer sees LDS reads that might alias the in-flight
// direct-to-LDS loads of the next tile and stops draining vmcnt in the middle of the tile.
#define A32_RD128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(dst) : "v"(addr), "n"(off))
#define A32_RDTR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=&v"(dst) : "v"(addr), "n"(off))
#define A32_WAIT(n, ...) asm volatile("s_waitcnt lgkmcnt(" #n ")" : __VA_ARGS__)
#define A32_FENCE() __builtin_amdgcn_sched_barrier(0)
// the 2 ND transpose reads (ND head-dim blocks x 2 halves) of one 16-row group, and the counted waits that name them as results
template <int ND, int OFF> LRP_DEVICE void tr_group(u32x2 (&dst)[NDX][2], const uint32_t (&adr)[NDX][2]) {
    A32_RDTR(dst[0][0], adr[0][0], OFF); A32_RDTR(dst[0][1], adr[0][1], OFF);
    A32_RDTR(dst[1][0], adr[1][0], OFF); A32_RDTR(dst[1][1], adr[1][1], OFF);
    if constexpr (ND > 2) { A32_RDTR(dst[2][0], adr[2][0], OFF); A32_RDTR(dst[2][1], adr[2][1], OFF); }
    if constexpr (ND > 3) { A32_RDTR(dst[3][0], adr[3][0], OFF); A32_RDTR(dst[3][1], adr[3][1], OFF); }
}
#define A32_TRV2(t) "+v"(t[0][0]), "+v"(t[0][1]), "+v"(t[1][0]), "+v"(t[1][1])
#define A32_TRV3(t) A32_TRV2(t), "+v"(t[2][0]), "+v"(t[2][1])
#define A32_TRV4(t) A32_TRV3(t), "+v"(t[3][0]), "+v"(t[3][1])
template <int ND, int N> LRP_DEVICE void wait_tr(u32x2 (&t)[NDX][2]) {
    if constexpr (ND == 4) asm volatile("s_waitcnt lgkmcnt(%[n])" : A32_TRV4(t) : [n] "n"(N));
    else if constexpr (ND == 3) asm volatile("s_waitcnt lgkmcnt(%[n])" : A32_TRV3(t) : [n] "n"(N));
    else asm volatile("s_waitcnt lgkmcnt(%[n])" : A32_TRV2(t) : [n] "n"(N));
}
template <int ND, int N> LRP_DEVICE void wait_tr2(u32x2 (&t)[NDX][2], u32x2 (&u)[NDX][2]) {
    if constexpr (ND == 4) asm volatile("s_waitcnt lgkmcnt(%[n])" : A32_TRV4(t), A32_TRV4(u) : [n] "n"(N));
    else if constexpr (ND == 3) asm volatile("s_waitcnt lgkmcnt(%[n])" : A32_TRV3(t), A32_TRV3(u) : [n] "n"(N));
    else asm volatile("s_waitcnt lgkmcnt(%[n])" : A32_TRV2(t), A32_TRV2(u) : [n] "n"(N));
}
// Measured and NOT adopted (profiles/r02_attention_experiments.txt): s_setprio schemes that favour one of the two waves of a
// SIMD (static, or high priority in the MFMA phases only) change nothing -- tools/attn_timeline.py shows the MFMA phases of a
// wave already running at close to their single-wave time, i.e. the two waves of a SIMD are complementary; what is left is the
// per-wave cost of the element-wise phase (VALU issue beside the other wave's MFMAs), the LDS-DMA issue cost of the staging
// loads and the per-workgroup prologue / store tail.
// dev builds only (-DA32_TIMELINE, tools/attn_timeline.py): per-wave shader-clock totals of the dQ kernel's loop segments, written
// over the first words of the wave's first output row
#ifdef A32_TIMELINE
#define A32_TS(slot)                                                          \
    {                                                                         \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                    \
        const uint32_t now_ = (uint32_t)__builtin_readcyclecounter();         \
        tl_acc[slot] += now_ - tl_prev;                                       \
        tl_prev = now_;                                                       \
    }
#else
#define A32_TS(slot)
#endif
template <int V> struct IC { static constexpr int value = V; };
template <int I, int N, typename F> LRP_DEVICE void sfor(F&& f) {
    if constexpr (I < N) {
        f(IC<I>{});
        sfor<I + 1, N>(static_cast<F&&>(f));
    }
}
// value of the other lane half (lane ^ 32) combined with one's own: v_permlane32_swap (upper 32 lanes of the first operand
// <-> lower 32 lanes of the second) instead of an LDS round trip.  Inline asm: the builtin's second result was folded away
// by the compiler when both operands are the same value; s_nop 1 = the VALU-write -> permlane-read wait states.
LRP_DEVICE void half_swap(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
LRP_DEVICE float half_max(float x) {
    float a = x, b = x;
    half_swap(a, b);
    return fmaxf(a, b);
}
LRP_DEVICE float half_sum(float x) {
    float a = x, b = x;
    half_swap(a, b);
    return a + b;
}
// row-fragment address of contraction step ks from the step-0 address: chunk (2 ks + hi) ^ rot = (2 ks) ^ (hi ^ rot), so the
// eight addresses differ by an XOR of ks << 5 (tile rows are 256-byte aligned).  Kept opaque (asm volatile) so that the
// compiler recomputes it next to the read instead of keeping eight address registers live across the kernel.
template <int KS> LRP_DEVICE uint32_t frag_addr(uint32_t a0) {
    if constexpr (KS == 0) return a0;
    else {
        uint32_t r;
        asm volatile("v_xor_b32 %0, %1, %2" : "=v"(r) : "n"(KS << 5), "v"(a0));
        return r;
    }
}
LRP_DEVICE uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)(lds_ptr_t)p; }
// the two halves of a transpose-read fragment as one MFMA operand
LRP_DEVICE bf16x8 join_tr(u32x2 a, u32x2 b) {
    const u32x4 v = {a[0], a[1], b[0], b[1]};
    return __builtin_bit_cast(bf16x8, v);
}

template <int DH>
LRP_DEVICE void load_row_frags(bf16x8* f, const bf16_t* base, int64_t ld, int row, int S, int hi) {
    const bool ok = row < S;
#pragma unroll
    for (int ks = 0; ks < DH / 16; ++ks) {
        if (ok) f[ks] = *reinterpret_cast<const bf16x8*>(base + (int64_t)row * ld + ks * 16 + hi * 8);
        else {
            u32x4 z = {0, 0, 0, 0};
            f[ks] = __builtin_bit_cast(bf16x8, z);
        }
    }
}
// out^T accumulators -> token-major rows: lane (row, hi) holds head-dim columns db*32 + 8 i + 4 hi + 0..3
// Round 5: 16-byte stores.  A row's 8-column groups are split between its two lanes (hi = 0: columns 8 i .. + 3, hi = 1: 8 i + 4 .. + 7), so the
// natural store is 4 x 8 bytes per 32-column block and lane -- 32 store instructions per lane for dK + dV, and the store tail of a workgroup is
// ISSUE-bound (MI355X_MICROARCH.md: ~9.3k cycles per 16 dwordx2 stores; cdna_hip_programming.md T21).  v_permlane32_swap between the packed
// registers of groups i and i + 1 gives the lower lane columns 8 i .. 8 i + 7 and the upper lane 8 (i + 1) .. + 7: one dwordx4 store per pair.
template <int DH>
LRP_DEVICE void store_rows(bf16_t* base, int64_t ld, int row, int S, const f32x16* acc, float mul, int hi) {
    const bool wide = (((reinterpret_cast<uintptr_t>(base) | (uintptr_t)(ld * 2)) & 15) == 0);        // wave-uniform (kernel arguments)
    if (!wide) {
        if (row >= S) return;
#pragma unroll
        for (int db = 0; db < DH / 32; ++db)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                bf16x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (bf16_t)(acc[db][4 * i + e] * mul);
                *reinterpret_cast<bf16x4*>(base + (int64_t)row * ld + db * 32 + 8 * i + 4 * hi) = v;
            }
        return;
    }
    bf16_t* const rb = base + (int64_t)row * ld + 8 * hi;               // upper lane: the second 8-column half of a pair
#pragma unroll
    for (int db = 0; db < DH / 32; ++db)
#pragma unroll
        for (int i = 0; i < 4; i += 2) {
            bf16x4 va, vb;
#pragma unroll
            for (int e = 0; e < 4; ++e) { va[e] = (bf16_t)(acc[db][4 * i + e] * mul); vb[e] = (bf16_t)(acc[db][4 * (i + 1) + e] * mul); }
            u32x2 a = __builtin_bit_cast(u32x2, va), b = __builtin_bit_cast(u32x2, vb);
            auto r0 = __builtin_amdgcn_permlane32_swap(a[0], b[0], false, false);      // every lane takes part (rows >= S only skip the store)
            auto r1 = __builtin_amdgcn_permlane32_swap(a[1], b[1], false, false);
            const u32x4 o = {r0[0], r1[0], r0[1], r1[1]};
            if (row < S) *reinterpret_cast<u32x4*>(rb + db * 32 + 8 * i) = o;
        }
}

LRP_DEVICE bool visible(int q, int key, int S, int causal, int window, int lo, int hi) {
    return (key < S) & (!causal | (key <= q)) & ((window <= 0) | (key > q - window)) & (key >= lo) & (key < hi);   // no branches
}
// explicit stabilisers folded into ONE reciprocal (as attention.hip: lrp_ds2)
template <bool EXPL>
LRP_DEVICE float lrp_ds(float s_raw, float p, float dp, float Dq, float scale, float eps_mask, float eps_qk) {
    if constexpr (!EXPL) return p * (dp - Dq);                 // * scale / 2 folded into the final store
    else {
        const float s2 = s_raw * scale;
        return p * (dp - Dq) * scale * (s2 * s_raw) * __builtin_amdgcn_rcpf((s2 + eps_mask) * (2.f * s_raw + eps_qk));
    }
}

// ---- per-row key intervals: what a workgroup / a wave can see at all -------------------------------------------------------------------
// Callers that express their whole mask as intervals (causal = 0: Gemma-3 image + text prompts, packed sequences) would otherwise stream
// EVERY key tile past every query block and mask element by element.  The kernels derive the tile range from the intervals themselves:
// forward / dQ: the union [min lo, max hi) over the workgroup's query rows bounds the key loop, the wave's own union skips dead tiles, and a
// tile inside the INTERSECTION [max lo, min hi) of the wave's rows needs no per-element interval mask; dK / dV: the query rows whose interval
// meets the workgroup's keys bound the query loop.  No assumption on the intervals (monotone or not); one LDS round trip per workgroup.
struct IvWave { int lo_min, hi_max, lo_max, hi_min; };
LRP_DEVICE int wave_min(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
    return v;
}
LRP_DEVICE int wave_max(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
    return v;
}
// lane's row interval [lo, hi) (rows >= S: in_range = false) -> the wave's bounds; sh[2 * wave], sh[2 * wave + 1] <- its union (the caller
// barriers and folds the waves' unions with iv_fold)
LRP_DEVICE IvWave iv_wave(int lo, int hi, bool in_range, int wave, int* sh) {
    const bool sees = in_range && hi > lo;
    IvWave w;
    w.lo_min = wave_min(sees ? lo : 0x7fffffff);
    w.hi_max = wave_max(sees ? hi : 0);
    w.lo_max = wave_max(in_range ? lo : 0);
    w.hi_min = wave_min(in_range ? hi : 0x7fffffff);
    sh[2 * wave] = w.lo_min;
    sh[2 * wave + 1] = w.hi_max;
    return w;
}
template <int NWAVES> LRP_DEVICE void iv_fold(const int* sh, int& lo, int& hi) {
    lo = sh[0];
    hi = sh[1];
#pragma unroll
    for (int w = 1; w < NWAVES; ++w) { lo = min(lo, sh[2 * w]); hi = max(hi, sh[2 * w + 1]); }
}
// dK / dV (the row side is keys; queries stream): a table [S / 32][4] = {lo_min, hi_max, lo_max, hi_min} of every 32-query block of batch entry b
// behind the kernel's tile buffers (host: + 16 bytes per block of dynamic LDS), built once per workgroup; from it the query range whose
// intervals meet the workgroup's keys [k0, k0 + BK), and per (query block, wave) "dead" / "no interval mask needed" as for forward / dQ
// one table row, read with a hand-issued ds_read_b128 (a compiler-visible LDS read in the tile loop would make the compiler drain the
// in-flight direct-to-LDS loads of the next tile first) and moved to scalar registers (wave-uniform branches)
LRP_DEVICE void iv_row(const int* tab, int blk, int& lo_min, int& hi_max, int& lo_max, int& hi_min) {
    u32x4 t;
    const uint32_t a = (uint32_t)(uintptr_t)(lds_ptr_t)tab + 16u * (uint32_t)blk;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(t) : "v"(a));
    lo_min = __builtin_amdgcn_readfirstlane((int)t[0]);
    hi_max = __builtin_amdgcn_readfirstlane((int)t[1]);
    lo_max = __builtin_amdgcn_readfirstlane((int)t[2]);
    hi_min = __builtin_amdgcn_readfirstlane((int)t[3]);
}
template <int NWAVES> LRP_DEVICE void iv_block_table(const int* rlo, const int* rhi, int S, int k0, int BK, int wave, int lane, int* tab, int* sh,
                                                     int& qlo, int& qhi) {
    const int nblk = (S + 31) >> 5;
    for (int blk = wave; blk < nblk; blk += NWAVES) {
        const int r = blk * 32 + (lane & 31);
        const bool in = r < S;
        const int lo = in ? rlo[r] : 0, hi = in ? rhi[r] : 0;
        const bool sees = in && hi > lo;
        const int a = wave_min(sees ? lo : 0x7fffffff), e = wave_max(sees ? hi : 0), c = wave_max(in ? lo : 0), d_ = wave_min(in ? hi : 0x7fffffff);
        if (lane == 0) { tab[4 * blk] = a; tab[4 * blk + 1] = e; tab[4 * blk + 2] = c; tab[4 * blk + 3] = d_; }
    }
    __syncthreads();
    int a = 0x7fffffff, e = 0;
    for (int blk = threadIdx.x; blk < nblk; blk += NWAVES * 64)
        if (tab[4 * blk] < k0 + BK && tab[4 * blk + 1] > k0) { a = min(a, blk * 32); e = max(e, blk * 32 + 32); }
    a = wave_min(a);
    e = wave_max(e);
    sh[2 * wave] = a;
    sh[2 * wave + 1] = e;
    __syncthreads();
    iv_fold<NWAVES>(sh, qlo, qhi);
    __syncthreads();                                              // the scratch words are free again (they lie in the first tile buffer)
}

// =====================================================================================================================
// forward
// =====================================================================================================================
template <int DH>
__global__ __launch_bounds__(NWQ * 64, 2) void fwd_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, bf16_t* __restrict__ o,
    float* __restrict__ lse, int S, int Hq, int Hkv, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, float scale,
    int causal, int window, int B, int q_begin, const int* __restrict__ row_lo, const int* __restrict__ row_hi) {
    constexpr int BQ = NWQ * 32, STAGE = 2 * TILE, NK = DH / 16, ND32 = DH / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rep = Hq / Hkv, nqb = (S + BQ - 1) / BQ;
    int bhk, item;
    if (!xcd_group_decode(blockIdx.x, B * Hkv, rep * nqb, bhk, item)) return;
    const int b = bhk / Hkv, hk = bhk % Hkv, h = hk * rep + item % rep;
    const int qblk = nqb - 1 - item / rep;                       // heavy (late) causal blocks first
    const int q0 = qblk * BQ, qw = q0 + wave * 32, qi = qw + l31;
    if (q0 + BQ <= q_begin) return;
    const bf16_t* kb_ = k + (int64_t)b * S * ldk + (int64_t)hk * DH;
    const bf16_t* vb_ = v + (int64_t)b * S * ldv + (int64_t)hk * DH;

    bf16x8 qf[NK];
    load_row_frags<DH>(qf, q + (int64_t)b * S * ldq + (int64_t)h * DH, ldq, qi, S, hi);
    int ivlo = 0, ivhi = S;
    if (row_lo != nullptr && qi < S) { ivlo = row_lo[(int64_t)b * S + qi]; ivhi = row_hi[(int64_t)b * S + qi]; }
    f32x16 oacc[ND32];
#pragma unroll
    for (int db = 0; db < ND32; ++db) oacc[db] = zero16();
    float m_run = -INFINITY, l_run = 0.f;
    const float c1 = scale * LRP_LOG2E;
    int kend = S;
    if (causal) kend = min(S, q0 + BQ);
    int kbeg = 0;
    if (window > 0) { kbeg = q0 - window + 1; kbeg = kbeg < 0 ? 0 : (kbeg / CT) * CT; }
    IvWave ivw = {0, S, 0, S};
    if (row_lo != nullptr) {
        int* ivsh = reinterpret_cast<int*>(smem);                 // the tile buffers are still unused (a static __shared__ array would move
        ivw = iv_wave(ivlo, ivhi, qi < S, wave, ivsh);            // the dynamic base off the alignment the swizzled addresses rely on)
        __syncthreads();
        int glo, ghi;
        iv_fold<NWQ>(ivsh, glo, ghi);
        __syncthreads();                                          // everyone has read the scratch words before the first tile lands on them
        kbeg = max(kbeg, (min(glo, S) / CT) * CT);
        kend = min(kend, ghi);
    }

    auto stage = [&](int kt0, int buf) {
        char* sb = smem + buf * STAGE;
        stage_tile<NWQ, DH>(kb_, ldk, kt0, S, sb, wave, lane);
        stage_tile<NWQ, DH>(vb_, ldv, kt0, S, sb + TILE, wave, lane);
    };
    if (kbeg < kend) stage(kbeg, 0);
    __syncthreads();
    // absolute LDS addresses of the current tile's fragments (K tile; V tile = + TILE; 32-row block kb = + kb * 8192)
    uint32_t arm[NK], atr[NDX][2];
    {
        const uint32_t sbase = lds_addr(smem);
        LaneAddr la;
        la.init(lane);
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) arm[ks] = sbase + la.rm[ks];
#pragma unroll
        for (int db = 0; db < ND32; ++db) { atr[db][0] = sbase + la.tr[db][0]; atr[db][1] = sbase + la.tr[db][1]; }
    }
    int cur = 0;
    for (int kt0 = kbeg; kt0 < kend; kt0 += CT) {
        if (kt0 + CT < kend) stage(kt0 + CT, cur ^ 1);
        // this wave's rows see nothing of the tile (causal: every key beyond the wave's last query): skip its arithmetic
        const bool dead = (causal && kt0 > qw + 31) || kt0 >= ivw.hi_max || kt0 + CT <= ivw.lo_min;
        if (!dead) {
            // ---- S^T of both 32-key blocks: 8 steps x 2 K fragments, read two steps ahead
            f32x16 st[2] = {zero16(), zero16()};
            bf16x8 f0[3], f1[3];
            A32_RD128(f0[0], arm[0], 0); A32_RD128(f1[0], arm[0], 32 * KP);
            A32_RD128(f0[1], arm[1], 0); A32_RD128(f1[1], arm[1], 32 * KP);
            sfor<0, NK>([&](auto ksc) {
                constexpr int ks = decltype(ksc)::value, cu = ks % 3, nx = (ks + 2) % 3;
                if constexpr (ks + 2 < NK) {
                    A32_RD128(f0[nx], arm[ks + 2], 0); A32_RD128(f1[nx], arm[ks + 2], 32 * KP);
                    A32_WAIT(4, "+v"(f0[cu]), "+v"(f1[cu]));
                } else if constexpr (ks + 1 < NK) {
                    A32_WAIT(2, "+v"(f0[cu]), "+v"(f1[cu]));
                } else {
                    A32_WAIT(0, "+v"(f0[cu]), "+v"(f1[cu]));
                }
                st[0] = mfma32(f0[cu], qf[ks], st[0]);
                st[1] = mfma32(f1[cu], qf[ks], st[1]);
                A32_FENCE();
            });
            // ---- V^T fragments of the first key block: in flight under the softmax
            u32x2 tv0[2][NDX][2], tv1[2][NDX][2];
            tr_group<ND32, TILE>(tv0[0], atr);
            tr_group<ND32, TILE + 16 * KP>(tv0[1], atr);
            A32_FENCE();
            const bool need_mask = (kt0 + CT > S) || (causal && kt0 + CT - 1 > qw) || (window > 0) ||
                                   (row_lo != nullptr && !(kt0 >= ivw.lo_max && kt0 + CT <= ivw.hi_min));
            if (need_mask) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        st[kb][r] = visible(qi, kt0 + kb * 32 + crow(r, hi), S, causal, window, ivlo, ivhi) ? st[kb][r] : -INFINITY;
            }
            float mx = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[kb][r]);
            mx = half_max(mx);
            // LAZY running maximum (as the d = 256 forward): the reference point m_run moves only when some row's maximum grew by more than
            // 2^8 in the exponent's units -- with 32 rows per wave "some row's maximum moved" is true on most tiles, and every move costs a
            // rescale of the DH / 2 accumulator registers.  Between moves p = exp2((s - m_run) c1) <= 256 (exact in fp32, the same relative
            // rounding in bf16); l_run uses the same reference, so o = acc / l and lse = m_run scale + log l are unchanged up to rounding.
            float m_new = m_run;
            if (__any((mx - m_run) * c1 > 8.f || (m_run == -INFINITY && mx != -INFINITY))) m_new = fmaxf(m_run, mx);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            const float nm2 = -m_use * c1;
            float rs = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = fast_exp2(__builtin_fmaf(st[kb][r], c1, nm2));
                    st[kb][r] = p;
                    rs += p;
                }
            rs = half_sum(rs);
            if (__any(m_new != m_run)) {
                const float alpha = fast_exp2((m_run - m_use) * c1);
                l_run = l_run * alpha + rs;
#pragma unroll
                for (int db = 0; db < ND32; ++db) oacc[db] *= alpha;
            } else l_run += rs;
            m_run = m_new;
            const bf16x8 p00 = pack8(st[0], 0), p01 = pack8(st[0], 1), p10 = pack8(st[1], 0), p11 = pack8(st[1], 1);
            A32_FENCE();
            // (lgkmcnt is a 4-bit counter on gfx9: never count on more than 15 outstanding reads)
            tr_group<ND32, TILE + 32 * KP>(tv1[0], atr);
            wait_tr2<ND32, 2 * ND32>(tv0[0], tv0[1]);
#pragma unroll
            for (int db = 0; db < ND32; ++db) oacc[db] = mfma32(join_tr(tv0[0][db][0], tv0[0][db][1]), p00, oacc[db]);
            A32_FENCE();
            tr_group<ND32, TILE + 48 * KP>(tv1[1], atr);
#pragma unroll
            for (int db = 0; db < ND32; ++db) oacc[db] = mfma32(join_tr(tv0[1][db][0], tv0[1][db][1]), p01, oacc[db]);
            A32_FENCE();
            wait_tr<ND32, 2 * ND32>(tv1[0]);
#pragma unroll
            for (int db = 0; db < ND32; ++db) oacc[db] = mfma32(join_tr(tv1[0][db][0], tv1[0][db][1]), p10, oacc[db]);
            A32_FENCE();
            wait_tr<ND32, 0>(tv1[1]);
#pragma unroll
            for (int db = 0; db < ND32; ++db) oacc[db] = mfma32(join_tr(tv1[1][db][0], tv1[1][db][1]), p11, oacc[db]);
            A32_FENCE();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the next tile has landed (this wave's pieces)
        __syncthreads();
        {
            const uint32_t delta = cur ? (uint32_t)-STAGE : (uint32_t)STAGE;
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) arm[ks] += delta;
#pragma unroll
            for (int db = 0; db < ND32; ++db) { atr[db][0] += delta; atr[db][1] += delta; }
        }
        cur ^= 1;
    }
    const float inv = (l_run > 0.f) ? 1.f / l_run : 0.f;
    store_rows<DH>(o + (int64_t)b * S * ldo + (int64_t)h * DH, ldo, qi, S, oacc, inv, hi);
    if (hi == 0 && qi < S) lse[((int64_t)b * Hq + h) * S + qi] = m_run * scale + __logf(l_run);
}

// =====================================================================================================================
// dQ: row side = 32 queries per wave; K and V tiles stream
// =====================================================================================================================
template <bool EXPL, int DH>
__global__ __launch_bounds__(NWQ * 64, 2) void dq_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, const bf16_t* __restrict__ gho,
    const float* __restrict__ lse, const float* __restrict__ Dd, bf16_t* __restrict__ dq, int S, int Hq, int Hkv, int64_t ldq,
    int64_t ldk, int64_t ldv, int64_t ldg, int64_t lddq, float scale, float eps_mask, float eps_qk, int causal, int window,
    int B, int q_begin, const int* __restrict__ row_lo, const int* __restrict__ row_hi, const bf16_t* __restrict__ ofw, int64_t ldo,
    float* __restrict__ Dout, const float* __restrict__ rope_cos, const float* __restrict__ rope_sin) {
    constexpr int BQ = NWQ * 32, STAGE = 2 * TILE, NK = DH / 16, ND32 = DH / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rep = Hq / Hkv, nqb = (S + BQ - 1) / BQ;
    int bhk, item;
    if (!xcd_group_decode(blockIdx.x, B * Hkv, rep * nqb, bhk, item)) return;
    const int b = bhk / Hkv, hk = bhk % Hkv, h = hk * rep + item % rep;
    const int qblk = nqb - 1 - item / rep;
    const int q0 = qblk * BQ, qw = q0 + wave * 32, qi = qw + l31;
    if (q0 + BQ <= q_begin) return;
    const bf16_t* kb_ = k + (int64_t)b * S * ldk + (int64_t)hk * DH;
    const bf16_t* vb_ = v + (int64_t)b * S * ldv + (int64_t)hk * DH;

    bf16x8 qf[NK], gf[NK];
    load_row_frags<DH>(qf, q + (int64_t)b * S * ldq + (int64_t)h * DH, ldq, qi, S, hi);
    load_row_frags<DH>(gf, gho + (int64_t)b * S * ldg + (int64_t)h * DH, ldg, qi, S, hi);
    const float lse2 = ((qi < S) ? lse[((int64_t)b * Hq + h) * S + qi] : 0.f) * LRP_LOG2E;
    float Dq;
    if (ofw != nullptr) {
        // D_i = sum_d Gho_i o_i from the rows themselves: the wave holds its 32 Gho rows as fragments anyway (two lanes per row, 64 columns each),
        // so the stand-alone attn_bwd_prep pass (3 x 67 MB per layer at B 4 / S 2048) reduces to one more row read here; the dK / dV kernel, which
        // runs after this one, takes D from Dout
        bf16x8 of[NK];
        load_row_frags<DH>(of, ofw + (int64_t)b * S * ldo + (int64_t)h * DH, ldo, qi, S, hi);
        float sd = 0.f;
#pragma unroll
        for (int ks = 0; ks < NK; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) sd += (float)gf[ks][e] * (float)of[ks][e];
        sd += __shfl_xor(sd, 32);
        Dq = sd;
        if (hi == 0 && qi < S) Dout[((int64_t)b * Hq + h) * S + qi] = sd;
    } else Dq = (qi < S) ? Dd[((int64_t)b * Hq + h) * S + qi] : 0.f;
    int ivlo = 0, ivhi = S;
    if (row_lo != nullptr && qi < S) { ivlo = row_lo[(int64_t)b * S + qi]; ivhi = row_hi[(int64_t)b * S + qi]; }
    f32x16 acc[ND32];
#pragma unroll
    for (int db = 0; db < ND32; ++db) acc[db] = zero16();
    const float c1 = scale * LRP_LOG2E;
    int kend = S;
    if (causal) kend = min(S, q0 + BQ);
    int kbeg = 0;
    if (window > 0) { kbeg = q0 - window + 1; kbeg = kbeg < 0 ? 0 : (kbeg / CT) * CT; }
    IvWave ivw = {0, S, 0, S};
    if (row_lo != nullptr) {
        int* ivsh = reinterpret_cast<int*>(smem);                 // the tile buffers are still unused (a static __shared__ array would move
        ivw = iv_wave(ivlo, ivhi, qi < S, wave, ivsh);            // the dynamic base off the alignment the swizzled addresses rely on)
        __syncthreads();
        int glo, ghi;
        iv_fold<NWQ>(ivsh, glo, ghi);
        __syncthreads();                                          // everyone has read the scratch words before the first tile lands on them
        kbeg = max(kbeg, (min(glo, S) / CT) * CT);
        kend = min(kend, ghi);
    }

    auto stage = [&](int kt0, int buf) {
        char* sb = smem + buf * STAGE;
        stage_tile<NWQ, DH>(kb_, ldk, kt0, S, sb, wave, lane);
        stage_tile<NWQ, DH>(vb_, ldv, kt0, S, sb + TILE, wave, lane);
    };
    if (kbeg < kend) stage(kbeg, 0);
    __syncthreads();
    // absolute LDS addresses of the current tile's fragments (K tile; V tile = + TILE; 32-row block kb = + kb * 8192), moved
    // from stage to stage with the loop
    uint32_t arm[NK], atr[NDX][2];
    {
        const uint32_t sbase = lds_addr(smem);
        LaneAddr la;
        la.init(lane);
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) arm[ks] = sbase + la.rm[ks];
#pragma unroll
        for (int db = 0; db < ND32; ++db) { atr[db][0] = sbase + la.tr[db][0]; atr[db][1] = sbase + la.tr[db][1]; }
    }
#ifdef A32_TIMELINE
    uint32_t tl_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl_prev = (uint32_t)__builtin_readcyclecounter();
    const uint32_t tl_start = tl_prev;
#endif
    int cur = 0;
    for (int kt0 = kbeg; kt0 < kend; kt0 += CT) {
        if (kt0 + CT < kend) stage(kt0 + CT, cur ^ 1);
        A32_TS(0);                                                       // 0: loop overhead + staging issue
        sfor<0, 2>([&](auto kbc) {
            constexpr int kb = decltype(kbc)::value, KO = kb * 32 * KP;
            const int kk0 = kt0 + kb * 32;
            if ((causal && kk0 > qw + 31) || kk0 >= S || kk0 >= ivw.hi_max || kk0 + 32 <= ivw.lo_min) return;   // block invisible to every row of this wave
            // ---- S^T and dP^T: 8 steps x {K fragment, V fragment}, read two steps ahead
            f32x16 st = zero16(), dp = zero16();
            bf16x8 fk[3], fv[3];
            A32_RD128(fk[0], arm[0], KO); A32_RD128(fv[0], arm[0], KO + TILE);
            A32_RD128(fk[1], arm[1], KO); A32_RD128(fv[1], arm[1], KO + TILE);
            sfor<0, NK>([&](auto ksc) {
                constexpr int ks = decltype(ksc)::value, cu = ks % 3, nx = (ks + 2) % 3;
                if constexpr (ks + 2 < NK) {
                    A32_RD128(fk[nx], arm[ks + 2], KO); A32_RD128(fv[nx], arm[ks + 2], KO + TILE);
                    A32_WAIT(4, "+v"(fk[cu]), "+v"(fv[cu]));
                } else if constexpr (ks + 1 < NK) {
                    A32_WAIT(2, "+v"(fk[cu]), "+v"(fv[cu]));
                } else {
                    A32_WAIT(0, "+v"(fk[cu]), "+v"(fv[cu]));
                }
                st = mfma32(fk[cu], qf[ks], st);
                dp = mfma32(fv[cu], gf[ks], dp);
                A32_FENCE();
            });
            A32_TS(1);                                                   // 1: S / dP phase
            // ---- K^T fragments of the dQ contraction: issued before the element-wise work that hides their latency
            u32x2 tk[2][NDX][2];
            tr_group<ND32, KO>(tk[0], atr);
            tr_group<ND32, KO + 16 * KP>(tk[1], atr);
            A32_FENCE();
            const bool masked = (kk0 + 32 > S) || (causal && kk0 + 31 > qw) || (window > 0) ||
                                (row_lo != nullptr && !(kk0 >= ivw.lo_max && kk0 + 32 <= ivw.hi_min));
            if (masked) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float s_raw = st[r];
                    float p = fast_exp2(__builtin_fmaf(s_raw, c1, -lse2));
                    if (!visible(qi, kk0 + crow(r, hi), S, causal, window, ivlo, ivhi)) p = 0.f;
                    st[r] = lrp_ds<EXPL>(s_raw, p, dp[r], Dq, scale, eps_mask, eps_qk);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float s_raw = st[r];
                    const float p = fast_exp2(__builtin_fmaf(s_raw, c1, -lse2));
                    st[r] = lrp_ds<EXPL>(s_raw, p, dp[r], Dq, scale, eps_mask, eps_qk);
                }
            }
            const bf16x8 df0 = pack8(st, 0), df1 = pack8(st, 1);
            A32_FENCE();
            A32_TS(2);                                                   // 2: element-wise
            wait_tr<ND32, 2 * ND32>(tk[0]);
#pragma unroll
            for (int db = 0; db < ND32; ++db) acc[db] = mfma32(join_tr(tk[0][db][0], tk[0][db][1]), df0, acc[db]);
            A32_FENCE();
            wait_tr<ND32, 0>(tk[1]);
#pragma unroll
            for (int db = 0; db < ND32; ++db) acc[db] = mfma32(join_tr(tk[1][db][0], tk[1][db][1]), df1, acc[db]);
            A32_FENCE();
            A32_TS(3);                                                   // 3: dQ phase (issue)
        });
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the next tile has landed (this wave's pieces)
        A32_TS(4);                                                       // 4: wait for the next tile
        __syncthreads();
        A32_TS(5);                                                       // 5: barrier
        {
            const uint32_t delta = cur ? (uint32_t)-STAGE : (uint32_t)STAGE;
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) arm[ks] += delta;
#pragma unroll
            for (int db = 0; db < ND32; ++db) { atr[db][0] += delta; atr[db][1] += delta; }
        }
        cur ^= 1;
    }
    const float omul = EXPL ? 1.f : 0.5f * scale;
    if constexpr (!EXPL && (DH == 64 || DH == 128)) {
        if (rope_cos != nullptr) {
            // RoPE's backward on the way out (round 5: no rope_bwd pass over dQ): the rotate-half partner of column c is c + DH/2 = the SAME lane's
            // accumulator block db + ND32/2 -- a1 = g1 cos(c) + g2 sin(c + DH/2), a2 = g2 cos(c + DH/2) - g1 sin(c) at the row's position, in fp32
            // on the un-rounded gradients (the formula of rope_bwd_kernel with both stabilisers 0)
            const int pos = qi < S ? qi : 0;
            const float* pc = rope_cos + (int64_t)pos * DH;
            const float* ps = rope_sin + (int64_t)pos * DH;
#pragma unroll
            for (int db = 0; db < ND32 / 2; ++db)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = db * 32 + 8 * i + 4 * hi;
                    // (the tables' two halves are EQUAL in the rotate-half convention -- emb = cat(freqs, freqs) -- and only the first is read:
                    // every workgroup pulls its rows of both tables through L2, 2048 workgroups x 128 rows x 2 x 256 B per launch)
                    const f32x4 c1 = *reinterpret_cast<const f32x4*>(pc + c), s1 = *reinterpret_cast<const f32x4*>(ps + c);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float g1 = acc[db][4 * i + e], g2 = acc[db + ND32 / 2][4 * i + e];
                        acc[db][4 * i + e] = g1 * c1[e] + g2 * s1[e];
                        acc[db + ND32 / 2][4 * i + e] = g2 * c1[e] - g1 * s1[e];
                    }
                }
        }
    }
    store_rows<DH>(dq + (int64_t)b * S * lddq + (int64_t)h * DH, lddq, qi, S, acc, omul, hi);
#ifdef A32_TIMELINE
    if (lane == 0 && qw < S) {
        uint32_t* w = reinterpret_cast<uint32_t*>(dq + ((int64_t)b * S + qw) * lddq + (int64_t)h * DH);
#pragma unroll
        for (int i = 0; i < 6; ++i) w[i] = tl_acc[i];
        w[6] = (uint32_t)__builtin_readcyclecounter() - tl_start;
        w[7] = (uint32_t)((kend - kbeg + CT - 1) / CT);
    }
#endif
}

// =====================================================================================================================
// dK / dV per query head: row side = 32 keys per wave; Q and Gho tiles (+ lse, D) stream
// =====================================================================================================================
// IV: the caller handed per-row key intervals (row_lo / row_hi).  A compile-time switch because the interval loads of the masked element-wise path
// sit behind a compiler-placed s_waitcnt vmcnt(0), which also drains the in-flight direct-to-LDS loads of the next tile -- in EVERY diagonal block
// of a plain causal call, intervals or not.
template <bool EXPL, int DH, bool IV>
__global__ __launch_bounds__(512, 2) void dkv_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, const bf16_t* __restrict__ gho,
    const float* __restrict__ lse, const float* __restrict__ Dd, bf16_t* __restrict__ dk, bf16_t* __restrict__ dv, int S, int Hq,
    int Hkv, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldg, int64_t lddk, int64_t lddv, float scale, float eps_mask,
    float eps_qk, int causal, int window, int B, int q_begin, const int* __restrict__ row_lo, const int* __restrict__ row_hi) {
    constexpr int BK = NW * 32, STAGE = 2 * TILE + 512, VROWS = 32 * KP, NK = DH / 16, ND32 = DH / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int bh, kblk;
#if A32_DKV_ORDER
    if (!xcd_item_major_decode(blockIdx.x, B * Hq, (S + BK - 1) / BK, bh, kblk)) return;
#else
    if (!xcd_group_decode(blockIdx.x, B * Hq, (S + BK - 1) / BK, bh, kblk)) return;
#endif
    const int b = bh / Hq, h = bh % Hq, hk = h / (Hq / Hkv);
    const int k0 = kblk * BK, kw = k0 + wave * 32, ki = kw + l31;
    const bf16_t* qb_ = q + (int64_t)b * S * ldq + (int64_t)h * DH;
    const bf16_t* gb_ = gho + (int64_t)b * S * ldg + (int64_t)h * DH;
    const float* lse_b = lse + ((int64_t)b * Hq + h) * S;
    const float* D_b = Dd + ((int64_t)b * Hq + h) * S;
    const int* rlo_b = (IV && row_lo) ? row_lo + (int64_t)b * S : nullptr;
    const int* rhi_b = (IV && row_lo) ? row_hi + (int64_t)b * S : nullptr;
    int* const ivtab = reinterpret_cast<int*>(smem + 2 * STAGE + NW * VROWS);    // per-32-query-block interval bounds (only with row intervals)

    // K fragments of the wave's 32 keys live in registers; the V fragments (32 more registers: with two 128-register
    // accumulators the kernel would spill) live in a per-wave [32 rows][256 B] LDS block in the tile layout
    bf16x8 kf[NK];
    load_row_frags<DH>(kf, k + (int64_t)b * S * ldk + (int64_t)hk * DH, ldk, ki, S, hi);
    char* sVw = smem + 2 * STAGE + wave * VROWS;
    {
        const bf16_t* vbase = v + (int64_t)b * S * ldv + (int64_t)hk * DH;
#if A32_BUFFER_STAGING
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)vbase, 0, (int)(((int64_t)(S - 1) * ldv + DH) * 2), 0x00020000);
        const int rl = lane >> 4, slot = lane & 15;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int chunk_ = slot ^ ((rl << 2) | (g & 3));
            const int voff = (DH == 128 || chunk_ < DH / 8) ? (int)(rl * ldv * 2) + (chunk_ << 4) : A32_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(sVw + g * 1024), 16, voff, (int)((int64_t)(kw + g * 4) * ldv * 2), 0, 0);
        }
#else
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int row = g * 4 + (lane >> 4), slot = lane & 15;
            int gr = kw + row;
            gr = gr < S ? gr : S - 1;
            if ((slot ^ rot4(row)) * 8 < DH)
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(vbase + (int64_t)gr * ldv + ((slot ^ rot4(row)) * 8)), (lds_ptr_t)(sVw + g * 1024), 16, 0, 0);
        }
#endif
    }
    f32x16 dkacc[ND32], dvacc[ND32];
#pragma unroll
    for (int db = 0; db < ND32; ++db) { dkacc[db] = zero16(); dvacc[db] = zero16(); }
    const float c1 = scale * LRP_LOG2E;
    int qbeg = 0, qend = S;
    if (causal) qbeg = (k0 / CT) * CT;
    if (window > 0) qend = min(S, k0 + BK - 1 + window);
    if (q_begin > qbeg) qbeg = (q_begin / CT) * CT;              // queries below q_begin carry no relevance
    if (rlo_b != nullptr) {
        int qlo, qhi;
        iv_block_table<NW>(rlo_b, rhi_b, S, k0, BK, wave, lane, ivtab, reinterpret_cast<int*>(smem), qlo, qhi);
        qbeg = max(qbeg, (min(qlo, S) / CT) * CT);
        qend = min(qend, qhi);
    }

    auto stage = [&](int qt0, int buf) {
        char* sb = smem + buf * STAGE;
        stage_tile<NW, DH>(qb_, ldq, qt0, S, sb, wave, lane);
        stage_tile<NW, DH>(gb_, ldg, qt0, S, sb + TILE, wave, lane);
        if (wave == 0) stage_stats(lse_b, qt0, S, sb + 2 * TILE, lane);
        if (wave == 1) stage_stats(D_b, qt0, S, sb + 2 * TILE + 256, lane);
    };
    if (qbeg < qend) stage(qbeg, 0);
    __syncthreads();
    // Measured and NOT adopted (round 5, profiles/r05_attention_experiments.txt): running the two 4-wave groups of the workgroup (waves w and w + 4
    // share a SIMD) HALF A BLOCK APART -- S^T / dP^T of one beside {element-wise, dV / dK} of the other, raw s_barrier between the half blocks,
    // one extra barrier up front for group 1, the GEMM's recipe -- is SLOWER (413 -> 471 us): beside a partner that streams MFMAs the element-wise
    // phase takes 1190 cycles per block instead of 890 beside a partner that is in its own element-wise phase, the S^T / dP^T phase does not get
    // faster (1150 cycles for 16 MFMAs either way), and four barriers per tile cost 2200 cycles of waiting instead of 1260.
    // absolute LDS addresses of the current stage (Q tile; Gho tile = + TILE; 32-row block qb = + qb * 8192; statistics at
    // + 2 TILE), moved from stage to stage with the loop; the wave's V block sits at a fixed address
    uint32_t arm0, avw0, atr[ND32][2], ast;
    {
        const uint32_t sbase = lds_addr(smem);
        LaneAddr la;
        la.init(lane);
        arm0 = sbase + la.rm[0];
        avw0 = arm0 + 2 * STAGE + wave * VROWS;                  // the wave's V block (fixed)
#pragma unroll
        for (int db = 0; db < ND32; ++db) { atr[db][0] = sbase + la.tr[db][0]; atr[db][1] = sbase + la.tr[db][1]; }
        ast = sbase + 2 * TILE + hi * 16;
    }
#ifdef A32_TIMELINE
    uint32_t tl_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl_prev = (uint32_t)__builtin_readcyclecounter();
    const uint32_t tl_start = tl_prev;
#endif
    int cur = 0;
    for (int qt0 = qbeg; qt0 < qend; qt0 += CT) {
        if (qt0 + CT < qend) stage(qt0 + CT, cur ^ 1);
        A32_TS(0);                                                       // 0: loop overhead + staging issue
        sfor<0, 2>([&](auto qbc) {
            constexpr int qb = decltype(qbc)::value, QO = qb * 32 * KP, SO = qb * 128;
            const int qq0 = qt0 + qb * 32;
            bool live = !((causal && qq0 + 31 < kw) || qq0 >= S);       // false: every query of the block precedes every key of the wave
            bool iv_mask = false;
            if (live && rlo_b != nullptr) {
                int t0_, t1_, t2_, t3_;
                iv_row(ivtab, qq0 >> 5, t0_, t1_, t2_, t3_);
                if (t1_ <= kw || t0_ >= kw + 32) live = false;           // no query of the block sees a key of this wave
                iv_mask = !(t2_ <= kw && t3_ >= kw + 32);                // some row's interval ends inside the wave's keys
            }
            if (!live) return;
            // ---- S^T and dP^T
            f32x16 st = zero16(), dp = zero16();
            f32x4 sl[2], sd[2];                                         // lse / D of the queries 8 i + 4 hi + 0..3, double-buffered
#if A32_DKV_P1 == 0
            // (round 2-4 form, kept for A/B: the contractions one after the other, fragment reads ONE step = one MFMA = 32 pipe cycles ahead)
            bf16x8 fq[2], fg[2], fv[2];
            A32_RD128(fq[0], arm0, QO);
            sfor<0, NK>([&](auto ksc) {
                constexpr int ks = decltype(ksc)::value, cu = ks & 1, nx = cu ^ 1;
                if constexpr (ks + 1 < NK) {
                    { const uint32_t aq = frag_addr<ks + 1>(arm0); A32_RD128(fq[nx], aq, QO); }
                    A32_WAIT(1, "+v"(fq[cu]));
                } else {
                    A32_RD128(fg[0], arm0, QO + TILE);
                    A32_RD128(fv[0], avw0, 0);
                    A32_WAIT(2, "+v"(fq[cu]));
                }
                st = mfma32(fq[cu], kf[ks], st);
                A32_FENCE();
            });
            sfor<0, NK>([&](auto ksc) {
                constexpr int ks = decltype(ksc)::value, cu = ks & 1, nx = cu ^ 1;
                if constexpr (ks + 1 < NK) {
                    { const uint32_t ag = frag_addr<ks + 1>(arm0); A32_RD128(fg[nx], ag, QO + TILE); }
                    { const uint32_t av = frag_addr<ks + 1>(avw0); A32_RD128(fv[nx], av, 0); }
                    A32_WAIT(2, "+v"(fg[cu]), "+v"(fv[cu]));
                } else {
                    A32_RD128(sl[0], ast, SO); A32_RD128(sd[0], ast, SO + 256);
                    A32_WAIT(2, "+v"(fg[cu]), "+v"(fv[cu]));
                }
                dp = mfma32(fg[cu], fv[cu], dp);
                A32_FENCE();
            });
#else
            // round 5: S^T and dP^T step by step side by side (two independent accumulator chains: the MFMAs of a step issue back to back) with
            // the step's three fragments -- Q, Gho, V -- read A32_DKV_P1 steps = 2 A32_DKV_P1 MFMAs = 64 A32_DKV_P1 pipe cycles ahead through a
            // ring of A32_DKV_P1 + 1 slots.  With one MFMA of cover (32 cycles) every contraction step waited out the LDS latency: ~100+ cycles
            // per MFMA in this phase, on BOTH waves of the SIMD at once (ISA: ds_read, s_waitcnt lgkmcnt(1), v_mfma per step).  The ring costs
            // no registers at the kernel's peak: the transpose-read groups, P / dS operands and statistics of phase C are dead here.
            constexpr int PD = A32_DKV_P1, RS = PD + 1;
            bf16x8 fq[RS], fg[RS], fv[RS];
            static_assert(PD < NK, "read-ahead deeper than the contraction");
#define A32_P1_PRO(pk)                                                                                          \
    if constexpr (PD > (pk)) {                                                                                  \
        const uint32_t aq = frag_addr<(pk)>(arm0), av = frag_addr<(pk)>(avw0);                                  \
        A32_RD128(fq[pk], aq, QO); A32_RD128(fg[pk], aq, QO + TILE); A32_RD128(fv[pk], av, 0);                  \
    }
            A32_P1_PRO(0) A32_P1_PRO(1) A32_P1_PRO(2) A32_P1_PRO(3)
#undef A32_P1_PRO
            sfor<0, NK>([&](auto ksc) {
                constexpr int ks = decltype(ksc)::value, cu = ks % RS, nx = (ks + PD) % RS;
                // reads younger than step ks's at the wait: the steps ks + 1 .. min(ks + PD, NK - 1) (3 each) and, from step NK - 2 on, the
                // first pair of statistics (2)
                constexpr int ahead = (ks + PD < NK ? PD : NK - 1 - ks);
                if constexpr (ks + PD < NK) {
                    const uint32_t aq = frag_addr<ks + PD>(arm0), av = frag_addr<ks + PD>(avw0);
                    A32_RD128(fq[nx], aq, QO); A32_RD128(fg[nx], aq, QO + TILE); A32_RD128(fv[nx], av, 0);
                }
                if constexpr (ks == NK - 2) { A32_RD128(sl[0], ast, SO); A32_RD128(sd[0], ast, SO + 256); }
                constexpr int nw = 3 * ahead + (ks >= NK - 2 ? 2 : 0);
                static_assert(nw <= 14, "lgkmcnt is a 4-bit counter");
                asm volatile("s_waitcnt lgkmcnt(%[n])" : "+v"(fq[cu]), "+v"(fg[cu]), "+v"(fv[cu]) : [n] "n"(nw));
                st = mfma32(fq[cu], kf[ks], st);
                dp = mfma32(fg[cu], fv[cu], dp);
                A32_FENCE();
            });
#endif
            A32_TS(1);                                                   // 1: S^T / dP^T phase
            // ---- element-wise work and the dV / dK contractions, one 16-query group j at a time:
            //   B(j): P and dS of the group's 8 accumulator registers (statistics double-buffered, one read pair ahead);
            //   C(j): dV^T += Gho^T P^T, dK^T += Q^T dS^T over the four head-dim blocks, transpose-read group g = (j, db)
            //         issued one group ahead (two 128-register accumulators + the K fragments leave room for two groups).
            // The first group of C(j) is in flight under B(j).
            u32x2 tg[2][2], tq[2][2];
#define A32_TRG(slot, jj, dbb)                                                                                        \
    A32_RDTR(tg[slot][0], atr[dbb][0], QO + TILE + (jj) * 16 * KP); A32_RDTR(tg[slot][1], atr[dbb][1], QO + TILE + (jj) * 16 * KP); \
    A32_RDTR(tq[slot][0], atr[dbb][0], QO + (jj) * 16 * KP); A32_RDTR(tq[slot][1], atr[dbb][1], QO + (jj) * 16 * KP)
#define A32_TIE(sl_) "+v"(tg[sl_][0]), "+v"(tg[sl_][1]), "+v"(tq[sl_][0]), "+v"(tq[sl_][1])
#define A32_DKV_STEP(g, wait_stmt)                                                                     \
    {                                                                                                  \
        constexpr int j_ = (g) / ND32, db_ = (g) % ND32, sl_ = (g)&1;                                  \
        wait_stmt;                                                                                     \
        dvacc[db_] = mfma32(join_tr(tg[sl_][0], tg[sl_][1]), pf[j_], dvacc[db_]);                      \
        dkacc[db_] = mfma32(join_tr(tq[sl_][0], tq[sl_][1]), df[j_], dkacc[db_]);                      \
        A32_FENCE();                                                                                   \
    }
            // lane (key, hi) holds queries qq0 + 8 i + 4 hi + e (i = r >> 2, e = r & 3)
            const bool masked = (qq0 + 32 > S) || (causal && qq0 < kw + 31) || (window > 0) || iv_mask;
            bf16x8 pf[2], df[2];
            auto elementwise = [&](auto ic) {
                constexpr int i = decltype(ic)::value, cu = i & 1, nx = cu ^ 1;
                // reads younger than the statistics of group i at this point: see the issue order below
                if constexpr (i == 0) {
                    A32_RD128(sl[nx], ast, SO + 32); A32_RD128(sd[nx], ast, SO + 32 + 256);
                    A32_WAIT(6, "+v"(sl[cu]), "+v"(sd[cu]));
                } else if constexpr (i == 1) {
                    A32_RD128(sl[nx], ast, SO + 64); A32_RD128(sd[nx], ast, SO + 64 + 256);
                    A32_WAIT(2, "+v"(sl[cu]), "+v"(sd[cu]), A32_TIE(0));
                } else if constexpr (i == 2) {
                    A32_RD128(sl[nx], ast, SO + 96); A32_RD128(sd[nx], ast, SO + 96 + 256);
                    A32_WAIT(6, "+v"(sl[cu]), "+v"(sd[cu]));
                } else if constexpr (ND32 == 3) {
                    A32_WAIT(0, "+v"(sl[cu]), "+v"(sd[cu]), A32_TIE(1));        // group (1, 0) is g = ND32: slot g & 1
                } else {
                    A32_WAIT(0, "+v"(sl[cu]), "+v"(sd[cu]), A32_TIE(0));
                }
#if A32_DKV_EW == 0
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * i + e;
                    const float s_raw = st[r];
                    float p = fast_exp2(__builtin_fmaf(s_raw, c1, -(sl[cu][e] * LRP_LOG2E)));
                    if (masked) {
                        const int qi = qq0 + 8 * i + 4 * hi + e;
                        int ivlo = 0, ivhi = S;
                        if constexpr (IV) { if (rlo_b != nullptr && qi < S) { ivlo = rlo_b[qi]; ivhi = rhi_b[qi]; } }
                        p = ((qi < S) & visible(qi, ki, S, causal, window, ivlo, ivhi)) ? p : 0.f;
                    }
                    pf[i >> 1][4 * (i & 1) + e] = (bf16_t)p;
                    df[i >> 1][4 * (i & 1) + e] = (bf16_t)lrp_ds<EXPL>(s_raw, p, dp[r], sd[cu][e], scale, eps_mask, eps_qk);
                }
#else
                // round 5: ONE wave-uniform branch per 4-query group instead of one per element (the per-element form compiled to 16 taken
                // s_cbranch per 32 x 32 block on the interior fast path, each between an exp2 and its consumers)
                float pe[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) pe[e] = fast_exp2(__builtin_fmaf(st[4 * i + e], c1, -(sl[cu][e] * LRP_LOG2E)));
                if (masked) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int qi = qq0 + 8 * i + 4 * hi + e;
                        int ivlo = 0, ivhi = S;
                        if constexpr (IV) { if (rlo_b != nullptr && qi < S) { ivlo = rlo_b[qi]; ivhi = rhi_b[qi]; } }
                        pe[e] = ((qi < S) & visible(qi, ki, S, causal, window, ivlo, ivhi)) ? pe[e] : 0.f;
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * i + e;
                    pf[i >> 1][4 * (i & 1) + e] = (bf16_t)pe[e];
                    df[i >> 1][4 * (i & 1) + e] = (bf16_t)lrp_ds<EXPL>(st[r], pe[e], dp[r], sd[cu][e], scale, eps_mask, eps_qk);
                }
#endif
                A32_FENCE();
            };
            // group g = (j, db) = (g / ND32, g % ND32) uses register slot g & 1; its reads are issued one group ahead
            A32_TRG(0, 0, 0);                                           // g = 0
            A32_FENCE();
            elementwise(IC<0>{});
            elementwise(IC<1>{});
            // (Measured and not kept, round 5: the second group's element-wise work BETWEEN the first group's dV / dK steps, so that the wave's own
            // asynchronous MFMAs run under its exp2 / dS instructions: 412 vs 410 us -- nothing.)
            if constexpr (ND32 == 4) {
                A32_TS(2);                                              // 2: element-wise
                A32_TRG(1, 0, 1);
                A32_DKV_STEP(0, (void)0);
                A32_TRG(0, 0, 2);
                A32_DKV_STEP(1, A32_WAIT(4, A32_TIE(1)));
                A32_TRG(1, 0, 3);
                A32_DKV_STEP(2, A32_WAIT(4, A32_TIE(0)));
                A32_TRG(0, 1, 0);                                       // g = 4: in flight under B(1)
                A32_DKV_STEP(3, A32_WAIT(4, A32_TIE(1)));
                A32_TS(3);                                              // 3: dV / dK contractions
                elementwise(IC<2>{});
                elementwise(IC<3>{});
                A32_TS(2);
                A32_TRG(1, 1, 1);
                A32_DKV_STEP(4, (void)0);
                A32_TRG(0, 1, 2);
                A32_DKV_STEP(5, A32_WAIT(4, A32_TIE(1)));
                A32_TRG(1, 1, 3);
                A32_DKV_STEP(6, A32_WAIT(4, A32_TIE(0)));
                A32_DKV_STEP(7, A32_WAIT(0, A32_TIE(1)));
                A32_TS(3);
            } else if constexpr (ND32 == 3) {
                A32_TRG(1, 0, 1);
                A32_DKV_STEP(0, (void)0);
                A32_TRG(0, 0, 2);
                A32_DKV_STEP(1, A32_WAIT(4, A32_TIE(1)));
                A32_TRG(1, 1, 0);                                       // g = 3: in flight under B(1)
                A32_DKV_STEP(2, A32_WAIT(4, A32_TIE(0)));
                elementwise(IC<2>{});
                elementwise(IC<3>{});
                A32_TRG(0, 1, 1);
                A32_DKV_STEP(3, (void)0);
                A32_TRG(1, 1, 2);
                A32_DKV_STEP(4, A32_WAIT(4, A32_TIE(0)));
                A32_DKV_STEP(5, A32_WAIT(0, A32_TIE(1)));
            } else {
                A32_TRG(1, 0, 1);
                A32_DKV_STEP(0, (void)0);
                A32_TRG(0, 1, 0);                                       // g = 2: in flight under B(1)
                A32_DKV_STEP(1, A32_WAIT(4, A32_TIE(1)));
                elementwise(IC<2>{});
                elementwise(IC<3>{});
                A32_TRG(1, 1, 1);
                A32_DKV_STEP(2, (void)0);
                A32_DKV_STEP(3, A32_WAIT(0, A32_TIE(1)));
            }
#undef A32_DKV_STEP
#undef A32_TIE
#undef A32_TRG
        });
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the next tile has landed (this wave's pieces)
        A32_TS(4);                                                       // 4: wait for the next tile
        __syncthreads();
        A32_TS(5);                                                       // 5: barrier
        {
            const uint32_t delta = cur ? (uint32_t)-STAGE : (uint32_t)STAGE;
            arm0 += delta;
#pragma unroll
            for (int db = 0; db < ND32; ++db) { atr[db][0] += delta; atr[db][1] += delta; }
            ast += delta;
        }
        cur ^= 1;
    }
    store_rows<DH>(dk + (int64_t)b * S * lddk + (int64_t)h * DH, lddk, ki, S, dkacc, EXPL ? 1.f : 0.5f * scale, hi);
    store_rows<DH>(dv + (int64_t)b * S * lddv + (int64_t)h * DH, lddv, ki, S, dvacc, 1.f, hi);
#ifdef A32_TIMELINE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0 && kw < S) {
        uint32_t* w = reinterpret_cast<uint32_t*>(dk + ((int64_t)b * S + kw) * lddk + (int64_t)h * DH);
#pragma unroll
        for (int i = 0; i < 6; ++i) w[i] = tl_acc[i];
        w[6] = (uint32_t)__builtin_readcyclecounter() - tl_start;
        w[7] = (uint32_t)((qend - qbeg + CT - 1) / CT);
    }
#endif
}


// =====================================================================================================================
// head_dim 256 (Gemma-3: 8 query / 4 kv heads of d = 256, sliding window 1024 on 5 of 6 layers; ref lxt/efficient/models/gemma3.py:11-19,
// lxt/efficient/patches.py:193-203) on the same 32x32x16 / transpose-read form -- round 4.  What changes against d = 128:
//   * a tile row is 512 B (32 chunks of 16 B; the swizzle c ^ rot(r) acts on the low four chunk bits, so both read kinds stay
//     conflict-free), a tile is 32 column-side rows (16 KiB as before), a staging piece (1 KiB) is 2 rows;
//   * 16 contraction steps over the head dim and 8 head-dim blocks per out^T accumulator: 128 accumulator registers per output, so the
//     kernels run ONE wave per SIMD (4-wave workgroups, up to 512 registers per lane) and hide LDS latency inside the wave: fragment reads
//     run a 3-deep ring ahead of their MFMAs, transpose reads are issued one 8-read unit (4 head-dim blocks of one 16-row group) ahead;
//   * addresses: ONE per-lane base per read kind, the step / head-dim block enters through an XOR (chunk bits) and an immediate.
// No head-transposed copies in HBM for d = 256 any more (lrp_attn_needs_transposed(bf16, 256) = 0).
// =====================================================================================================================
namespace d256 {

constexpr int D2 = 256, KP2 = 512, CT2 = 32, TILE2 = CT2 * KP2, NK2 = 16, ND2 = 8, NW2 = 4;

// stage a [32 rows][512 B] tile: 16 pieces of 2 rows, 4 per wave (4-wave workgroups); rows >= S read as zero
LRP_DEVICE void stage_tile2(const bf16_t* base, int64_t ld, int row0, int S, char* lds, int wave, int lane) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(((int64_t)(S - 1) * ld + D2) * 2), 0x00020000);
    const int rl = lane >> 5, slot = lane & 31;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int grp = g * NW2 + wave;                               // piece: rows 2 grp, 2 grp + 1
        const int rot = ((2 * (wave & 1) + rl) << 2) | ((2 * g + (wave >> 1)) & 3);      // rot4(2 grp + rl)
        const int voff = (int)(rl * ld * 2) + ((slot ^ rot) << 4);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(lds + grp * 1024), 16, voff, (int)((int64_t)(row0 + 2 * grp) * ld * 2), 0, 0);
    }
}
// the same image for a wave-private 32-row block (all 16 pieces by one wave)
LRP_DEVICE void stage_block2(const bf16_t* base, int64_t ld, int row0, int S, char* lds, int lane) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(((int64_t)(S - 1) * ld + D2) * 2), 0x00020000);
    const int rl = lane >> 5, slot = lane & 31;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        const int rot = ((2 * (g & 1) + rl) << 2) | ((g >> 1) & 3);
        const int voff = (int)(rl * ld * 2) + ((slot ^ rot) << 4);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(lds + g * 1024), 16, voff, (int)((int64_t)(row0 + 2 * g) * ld * 2), 0, 0);
    }
}
// per-lane bases (relative to a tile): row fragment of step 0, transpose read of head-dim block 0 (two halves)
LRP_DEVICE uint32_t rm_base(int lane) {
    const int l31 = lane & 31, hi = lane >> 5;
    return (uint32_t)(l31 * KP2 + ((hi ^ rot4(l31)) << 4));
}
LRP_DEVICE uint32_t tr_base(int lane, int half) {
    const int l31 = lane & 31, hi = lane >> 5, i16 = lane & 15;
    const int r = half * 8 + 4 * hi + (i16 >> 2);
    const int chunk = ((l31 >> 4) << 1) + ((i16 & 3) >> 1);
    return (uint32_t)(r * KP2 + ((chunk ^ rot4(r)) << 4) + 8 * (i16 & 1));
}
// step ks of the row fragment: chunk 2 ks + hi -> XOR ks << 5; head-dim block db of a transpose read: chunk 4 db + .. -> XOR db << 6
template <int X> LRP_DEVICE uint32_t xor_addr(uint32_t a0) {
    if constexpr (X == 0) return a0;
    else {
        uint32_t r;
        asm volatile("v_xor_b32 %0, %1, %2" : "=v"(r) : "n"(X), "v"(a0));
        return r;
    }
}
LRP_DEVICE void load_row_frags2(bf16x8* f, const bf16_t* base, int64_t ld, int row, int S, int hi) {
    const bool ok = row < S;
#pragma unroll
    for (int ks = 0; ks < NK2; ++ks) {
        if (ok) f[ks] = *reinterpret_cast<const bf16x8*>(base + (int64_t)row * ld + ks * 16 + hi * 8);
        else {
            u32x4 z = {0, 0, 0, 0};
            f[ks] = __builtin_bit_cast(bf16x8, z);
        }
    }
}
LRP_DEVICE void store_rows2(bf16_t* base, int64_t ld, int row, int S, const f32x16* acc, float mul, int hi) {
    attn32::store_rows<D2>(base, ld, row, S, acc, mul, hi);           // the 16-byte-store form of the d <= 128 kernels (same register layout)
}
// one unit = the 8 transpose reads (4 head-dim blocks x 2 halves) of 16-row group J, head-dim blocks 4 DBH .. 4 DBH + 3, tile byte offset OFF
#define A2_UNIT(dst, a0, a1, J, DBH, OFF)                                                                            \
    {                                                                                                                \
        const uint32_t b0_ = xor_addr<((DBH) * 4 + 0) << 6>(a0), c0_ = xor_addr<((DBH) * 4 + 0) << 6>(a1);           \
        const uint32_t b1_ = xor_addr<((DBH) * 4 + 1) << 6>(a0), c1_ = xor_addr<((DBH) * 4 + 1) << 6>(a1);           \
        const uint32_t b2_ = xor_addr<((DBH) * 4 + 2) << 6>(a0), c2_ = xor_addr<((DBH) * 4 + 2) << 6>(a1);           \
        const uint32_t b3_ = xor_addr<((DBH) * 4 + 3) << 6>(a0), c3_ = xor_addr<((DBH) * 4 + 3) << 6>(a1);           \
        A32_RDTR(dst[0][0], b0_, (OFF) + (J) * 16 * KP2); A32_RDTR(dst[0][1], c0_, (OFF) + (J) * 16 * KP2);          \
        A32_RDTR(dst[1][0], b1_, (OFF) + (J) * 16 * KP2); A32_RDTR(dst[1][1], c1_, (OFF) + (J) * 16 * KP2);          \
        A32_RDTR(dst[2][0], b2_, (OFF) + (J) * 16 * KP2); A32_RDTR(dst[2][1], c2_, (OFF) + (J) * 16 * KP2);          \
        A32_RDTR(dst[3][0], b3_, (OFF) + (J) * 16 * KP2); A32_RDTR(dst[3][1], c3_, (OFF) + (J) * 16 * KP2);          \
    }
#define A2_UNITV(t) "+v"(t[0][0]), "+v"(t[0][1]), "+v"(t[1][0]), "+v"(t[1][1]), "+v"(t[2][0]), "+v"(t[2][1]), "+v"(t[3][0]), "+v"(t[3][1])

// ---------------------------------------------------------------------------------------------------------------------
// forward, d = 256: row side = 32 queries per wave, 32-key K / V tiles stream
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NW2 * 64, 1) void fwd256_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, bf16_t* __restrict__ o,
    float* __restrict__ lse, int S, int Hq, int Hkv, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, float scale,
    int causal, int window, int B, int q_begin, const int* __restrict__ row_lo, const int* __restrict__ row_hi) {
    constexpr int BQ = NW2 * 32, STAGE = 2 * TILE2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rep = Hq / Hkv, nqb = (S + BQ - 1) / BQ;
    int bhk, item;
    if (!xcd_group_decode(blockIdx.x, B * Hkv, rep * nqb, bhk, item)) return;
    const int b = bhk / Hkv, hk = bhk % Hkv, h = hk * rep + item % rep;
    const int qblk = nqb - 1 - item / rep;
    const int q0 = qblk * BQ, qw = q0 + wave * 32, qi = qw + l31;
    if (q0 + BQ <= q_begin) return;
    const bf16_t* kb_ = k + (int64_t)b * S * ldk + (int64_t)hk * D2;
    const bf16_t* vb_ = v + (int64_t)b * S * ldv + (int64_t)hk * D2;

    bf16x8 qf[NK2];
    load_row_frags2(qf, q + (int64_t)b * S * ldq + (int64_t)h * D2, ldq, qi, S, hi);
    int ivlo = 0, ivhi = S;
    if (row_lo != nullptr && qi < S) { ivlo = row_lo[(int64_t)b * S + qi]; ivhi = row_hi[(int64_t)b * S + qi]; }
    f32x16 oacc[ND2];
#pragma unroll
    for (int db = 0; db < ND2; ++db) oacc[db] = zero16();
    float m_run = -INFINITY, l_run = 0.f;
    const float c1 = scale * LRP_LOG2E;
    int kend = S;
    if (causal) kend = min(S, q0 + BQ);
    int kbeg = 0;
    if (window > 0) { kbeg = q0 - window + 1; kbeg = kbeg < 0 ? 0 : (kbeg / CT2) * CT2; }
    IvWave ivw = {0, S, 0, S};
    if (row_lo != nullptr) {
        int* ivsh = reinterpret_cast<int*>(smem);                 // the tile buffers are still unused (a static __shared__ array would move
        ivw = iv_wave(ivlo, ivhi, qi < S, wave, ivsh);            // the dynamic base off the alignment the swizzled addresses rely on)
        __syncthreads();
        int glo, ghi;
        iv_fold<NW2>(ivsh, glo, ghi);
        __syncthreads();                                          // everyone has read the scratch words before the first tile lands on them
        kbeg = max(kbeg, (min(glo, S) / CT2) * CT2);
        kend = min(kend, ghi);
    }

    auto stage = [&](int kt0, int buf) {
        char* sb = smem + buf * STAGE;
        stage_tile2(kb_, ldk, kt0, S, sb, wave, lane);
        stage_tile2(vb_, ldv, kt0, S, sb + TILE2, wave, lane);
    };
    if (kbeg < kend) stage(kbeg, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    uint32_t arm = lds_addr(smem) + rm_base(lane), at0 = lds_addr(smem) + tr_base(lane, 0), at1 = lds_addr(smem) + tr_base(lane, 1);
    int cur = 0;
    for (int kt0 = kbeg; kt0 < kend; kt0 += CT2) {
        if (kt0 + CT2 < kend) stage(kt0 + CT2, cur ^ 1);
        const bool dead = (causal && kt0 > qw + 31) || (window > 0 && kt0 + CT2 - 1 <= qw - window) || kt0 >= ivw.hi_max || kt0 + CT2 <= ivw.lo_min;
        if (!dead) {
            // ---- S^T of the 32 keys: 16 steps, fragments read two steps ahead
            f32x16 st = zero16();
            bf16x8 fk[3];
            A32_RD128(fk[0], arm, 0);
            { const uint32_t a_ = xor_addr<1 << 5>(arm); A32_RD128(fk[1], a_, 0); }
            sfor<0, NK2>([&](auto ksc) {
                constexpr int ks = decltype(ksc)::value, cu = ks % 3, nx = (ks + 2) % 3;
                (void)&fk; (void)&arm;
                if constexpr (ks + 2 < NK2) {
                    { const uint32_t a_ = xor_addr<(ks + 2) << 5>(arm); A32_RD128(fk[nx], a_, 0); }
                    A32_WAIT(2, "+v"(fk[cu]));
                } else if constexpr (ks + 1 < NK2) {
                    A32_WAIT(1, "+v"(fk[cu]));
                } else {
                    A32_WAIT(0, "+v"(fk[cu]));
                }
                st = mfma32(fk[cu], qf[ks], st);
                A32_FENCE();
            });
            // ---- first V^T unit in flight under the softmax
            u32x2 tv[2][4][2];
            A2_UNIT(tv[0], at0, at1, 0, 0, TILE2);
            A32_FENCE();
            // (a sliding window only needs per-element masks where the tile crosses the window's lower edge for some row of the wave)
            const bool need_mask = (kt0 + CT2 > S) || (causal && kt0 + CT2 - 1 > qw) || (window > 0 && kt0 <= qw + 31 - window) ||
                                   (row_lo != nullptr && !(kt0 >= ivw.lo_max && kt0 + CT2 <= ivw.hi_min));
            if (need_mask) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    st[r] = visible(qi, kt0 + crow(r, hi), S, causal, window, ivlo, ivhi) ? st[r] : -INFINITY;
            }
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[r]);
            mx = half_max(mx);
            // LAZY running maximum: the reference point m_run only moves when some row's maximum grew by more than 2^LAZY (in the exponent's
            // units) -- rescaling 128 accumulator registers through the accumulator file costs ~320 VALU operations, more than the tile's 32
            // MFMAs, and with 32 rows per wave "some row's maximum moved" is true on most tiles.  Between moves p = exp2((s - m_run) c1) may
            // exceed 1 (by at most 2^LAZY = 256: exact in fp32, same relative rounding in bf16); l_run uses the same reference, so o = acc / l
            // and lse = m_run scale + log l are unchanged up to rounding.
            constexpr float LAZY = 8.f;
            float m_new = m_run;
            if (__any((mx - m_run) * c1 > LAZY || (m_run == -INFINITY && mx != -INFINITY))) m_new = fmaxf(m_run, mx);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            const float nm2 = -m_use * c1;
            float rs = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = fast_exp2(__builtin_fmaf(st[r], c1, nm2));
                st[r] = p;
                rs += p;
            }
            rs = half_sum(rs);
            if (__any(m_new != m_run)) {
                const float alpha = fast_exp2((m_run - m_use) * c1);
                l_run = l_run * alpha + rs;
#pragma unroll
                for (int db = 0; db < ND2; ++db) oacc[db] *= alpha;
            } else l_run += rs;
            m_run = m_new;
            const bf16x8 p0 = pack8(st, 0), p1 = pack8(st, 1);
            A32_FENCE();
            // ---- O^T += V^T P^T: units (j, dbh) = (0,0) (0,1) (1,0) (1,1), next unit's reads issued before this unit's MFMAs
            A2_UNIT(tv[1], at0, at1, 0, 1, TILE2);
            A32_WAIT(8, A2_UNITV(tv[0]));
#pragma unroll
            for (int d4 = 0; d4 < 4; ++d4) oacc[d4] = mfma32(join_tr(tv[0][d4][0], tv[0][d4][1]), p0, oacc[d4]);
            A32_FENCE();
            A2_UNIT(tv[0], at0, at1, 1, 0, TILE2);
            A32_WAIT(8, A2_UNITV(tv[1]));
#pragma unroll
            for (int d4 = 0; d4 < 4; ++d4) oacc[4 + d4] = mfma32(join_tr(tv[1][d4][0], tv[1][d4][1]), p0, oacc[4 + d4]);
            A32_FENCE();
            A2_UNIT(tv[1], at0, at1, 1, 1, TILE2);
            A32_WAIT(8, A2_UNITV(tv[0]));
#pragma unroll
            for (int d4 = 0; d4 < 4; ++d4) oacc[d4] = mfma32(join_tr(tv[0][d4][0], tv[0][d4][1]), p1, oacc[d4]);
            A32_FENCE();
            A32_WAIT(0, A2_UNITV(tv[1]));
#pragma unroll
            for (int d4 = 0; d4 < 4; ++d4) oacc[4 + d4] = mfma32(join_tr(tv[1][d4][0], tv[1][d4][1]), p1, oacc[4 + d4]);
            A32_FENCE();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        {
            const uint32_t delta = cur ? (uint32_t)-STAGE : (uint32_t)STAGE;
            arm += delta; at0 += delta; at1 += delta;
        }
        cur ^= 1;
    }
    const float inv = (l_run > 0.f) ? 1.f / l_run : 0.f;
    store_rows2(o + (int64_t)b * S * ldo + (int64_t)h * D2, ldo, qi, S, oacc, inv, hi);
    if (hi == 0 && qi < S) lse[((int64_t)b * Hq + h) * S + qi] = m_run * scale + __logf(l_run);
}

// ---------------------------------------------------------------------------------------------------------------------
// dQ, d = 256: row side = 32 queries per wave; 32-key K / V tiles stream
// ---------------------------------------------------------------------------------------------------------------------
template <bool EXPL>
__global__ __launch_bounds__(NW2 * 64, 1) void dq256_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, const bf16_t* __restrict__ gho,
    const float* __restrict__ lse, const float* __restrict__ Dd, bf16_t* __restrict__ dq, int S, int Hq, int Hkv, int64_t ldq,
    int64_t ldk, int64_t ldv, int64_t ldg, int64_t lddq, float scale, float eps_mask, float eps_qk, int causal, int window,
    int B, int q_begin, const int* __restrict__ row_lo, const int* __restrict__ row_hi) {
    constexpr int BQ = NW2 * 32, STAGE = 2 * TILE2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rep = Hq / Hkv, nqb = (S + BQ - 1) / BQ;
    int bhk, item;
    if (!xcd_group_decode(blockIdx.x, B * Hkv, rep * nqb, bhk, item)) return;
    const int b = bhk / Hkv, hk = bhk % Hkv, h = hk * rep + item % rep;
    const int qblk = nqb - 1 - item / rep;
    const int q0 = qblk * BQ, qw = q0 + wave * 32, qi = qw + l31;
    if (q0 + BQ <= q_begin) return;
    const bf16_t* kb_ = k + (int64_t)b * S * ldk + (int64_t)hk * D2;
    const bf16_t* vb_ = v + (int64_t)b * S * ldv + (int64_t)hk * D2;

    bf16x8 qf[NK2], gf[NK2];
    load_row_frags2(qf, q + (int64_t)b * S * ldq + (int64_t)h * D2, ldq, qi, S, hi);
    load_row_frags2(gf, gho + (int64_t)b * S * ldg + (int64_t)h * D2, ldg, qi, S, hi);
    const float lse2 = ((qi < S) ? lse[((int64_t)b * Hq + h) * S + qi] : 0.f) * LRP_LOG2E;
    const float Dq = (qi < S) ? Dd[((int64_t)b * Hq + h) * S + qi] : 0.f;
    int ivlo = 0, ivhi = S;
    if (row_lo != nullptr && qi < S) { ivlo = row_lo[(int64_t)b * S + qi]; ivhi = row_hi[(int64_t)b * S + qi]; }
    f32x16 acc[ND2];
#pragma unroll
    for (int db = 0; db < ND2; ++db) acc[db] = zero16();
    const float c1 = scale * LRP_LOG2E;
    int kend = S;
    if (causal) kend = min(S, q0 + BQ);
    int kbeg = 0;
    if (window > 0) { kbeg = q0 - window + 1; kbeg = kbeg < 0 ? 0 : (kbeg / CT2) * CT2; }
    IvWave ivw = {0, S, 0, S};
    if (row_lo != nullptr) {
        int* ivsh = reinterpret_cast<int*>(smem);                 // the tile buffers are still unused (a static __shared__ array would move
        ivw = iv_wave(ivlo, ivhi, qi < S, wave, ivsh);            // the dynamic base off the alignment the swizzled addresses rely on)
        __syncthreads();
        int glo, ghi;
        iv_fold<NW2>(ivsh, glo, ghi);
        __syncthreads();                                          // everyone has read the scratch words before the first tile lands on them
        kbeg = max(kbeg, (min(glo, S) / CT2) * CT2);
        kend = min(kend, ghi);
    }

    auto stage = [&](int kt0, int buf) {
        char* sb = smem + buf * STAGE;
        stage_tile2(kb_, ldk, kt0, S, sb, wave, lane);
        stage_tile2(vb_, ldv, kt0, S, sb + TILE2, wave, lane);
    };
    if (kbeg < kend) stage(kbeg, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    uint32_t arm = lds_addr(smem) + rm_base(lane), at0 = lds_addr(smem) + tr_base(lane, 0), at1 = lds_addr(smem) + tr_base(lane, 1);
    int cur = 0;
    for (int kt0 = kbeg; kt0 < kend; kt0 += CT2) {
        if (kt0 + CT2 < kend) stage(kt0 + CT2, cur ^ 1);
        const bool dead = (causal && kt0 > qw + 31) || kt0 >= S || (window > 0 && kt0 + CT2 - 1 <= qw - window) || kt0 >= ivw.hi_max ||
                          kt0 + CT2 <= ivw.lo_min;
        if (!dead) {
            // ---- S^T and dP^T: 16 steps x {K fragment, V fragment}, read two steps ahead
            f32x16 st = zero16(), dp = zero16();
            bf16x8 fk[3], fv[3];
            A32_RD128(fk[0], arm, 0); A32_RD128(fv[0], arm, TILE2);
            { const uint32_t a_ = xor_addr<1 << 5>(arm); A32_RD128(fk[1], a_, 0); A32_RD128(fv[1], a_, TILE2); }
            sfor<0, NK2>([&](auto ksc) {
                constexpr int ks = decltype(ksc)::value, cu = ks % 3, nx = (ks + 2) % 3;
                (void)&fk; (void)&fv; (void)&arm;
                if constexpr (ks + 2 < NK2) {
                    { const uint32_t a_ = xor_addr<(ks + 2) << 5>(arm); A32_RD128(fk[nx], a_, 0); A32_RD128(fv[nx], a_, TILE2); }
                    A32_WAIT(4, "+v"(fk[cu]), "+v"(fv[cu]));
                } else if constexpr (ks + 1 < NK2) {
                    A32_WAIT(2, "+v"(fk[cu]), "+v"(fv[cu]));
                } else {
                    A32_WAIT(0, "+v"(fk[cu]), "+v"(fv[cu]));
                }
                st = mfma32(fk[cu], qf[ks], st);
                dp = mfma32(fv[cu], gf[ks], dp);
                A32_FENCE();
            });
            // ---- first K^T unit in flight under the element-wise work
            u32x2 tk[2][4][2];
            A2_UNIT(tk[0], at0, at1, 0, 0, 0);
            A32_FENCE();
            const bool masked = (kt0 + 32 > S) || (causal && kt0 + 31 > qw) || (window > 0 && kt0 <= qw + 31 - window) ||
                                (row_lo != nullptr && !(kt0 >= ivw.lo_max && kt0 + 32 <= ivw.hi_min));
            if (masked) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float s_raw = st[r];
                    float p = fast_exp2(__builtin_fmaf(s_raw, c1, -lse2));
                    if (!visible(qi, kt0 + crow(r, hi), S, causal, window, ivlo, ivhi)) p = 0.f;
                    st[r] = lrp_ds<EXPL>(s_raw, p, dp[r], Dq, scale, eps_mask, eps_qk);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float s_raw = st[r];
                    const float p = fast_exp2(__builtin_fmaf(s_raw, c1, -lse2));
                    st[r] = lrp_ds<EXPL>(s_raw, p, dp[r], Dq, scale, eps_mask, eps_qk);
                }
            }
            const bf16x8 df0 = pack8(st, 0), df1 = pack8(st, 1);
            A32_FENCE();
            // ---- dQ^T += K^T dS^T
            A2_UNIT(tk[1], at0, at1, 0, 1, 0);
            A32_WAIT(8, A2_UNITV(tk[0]));
#pragma unroll
            for (int d4 = 0; d4 < 4; ++d4) acc[d4] = mfma32(join_tr(tk[0][d4][0], tk[0][d4][1]), df0, acc[d4]);
            A32_FENCE();
            A2_UNIT(tk[0], at0, at1, 1, 0, 0);
            A32_WAIT(8, A2_UNITV(tk[1]));
#pragma unroll
            for (int d4 = 0; d4 < 4; ++d4) acc[4 + d4] = mfma32(join_tr(tk[1][d4][0], tk[1][d4][1]), df0, acc[4 + d4]);
            A32_FENCE();
            A2_UNIT(tk[1], at0, at1, 1, 1, 0);
            A32_WAIT(8, A2_UNITV(tk[0]));
#pragma unroll
            for (int d4 = 0; d4 < 4; ++d4) acc[d4] = mfma32(join_tr(tk[0][d4][0], tk[0][d4][1]), df1, acc[d4]);
            A32_FENCE();
            A32_WAIT(0, A2_UNITV(tk[1]));
#pragma unroll
            for (int d4 = 0; d4 < 4; ++d4) acc[4 + d4] = mfma32(join_tr(tk[1][d4][0], tk[1][d4][1]), df1, acc[4 + d4]);
            A32_FENCE();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        {
            const uint32_t delta = cur ? (uint32_t)-STAGE : (uint32_t)STAGE;
            arm += delta; at0 += delta; at1 += delta;
        }
        cur ^= 1;
    }
    store_rows2(dq + (int64_t)b * S * lddq + (int64_t)h * D2, lddq, qi, S, acc, EXPL ? 1.f : 0.5f * scale, hi);
}

// ---------------------------------------------------------------------------------------------------------------------
// dK / dV per query head, d = 256: row side = 32 keys per wave (128 keys per workgroup); 32-query Q / Gho tiles (+ lse, D) stream.
// TWO PASSES over the query tiles with ONE 128-register out^T accumulator: pass 0  dV^T += Gho^T P^T  (S^T, P),  pass 1  dK^T += Q^T dS^T
// (S^T, dP^T, dS).  dK^T and dV^T together are 256 accumulator registers -- all of the accumulator file -- and S^T / dP^T need 32 more: the
// fused form made the compiler shuttle accumulators between the two register classes every tile (~500 v_accvgpr moves against 64 MFMAs) and
// keep the K fragments in scratch.  The second S^T costs 16 of 80 MFMAs per tile.
// ---------------------------------------------------------------------------------------------------------------------
template <bool EXPL>
__global__ __launch_bounds__(NW2 * 64, 1) void dkv256_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, const bf16_t* __restrict__ gho,
    const float* __restrict__ lse, const float* __restrict__ Dd, bf16_t* __restrict__ dk, bf16_t* __restrict__ dv, int S, int Hq,
    int Hkv, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldg, int64_t lddk, int64_t lddv, float scale, float eps_mask,
    float eps_qk, int causal, int window, int B, int q_begin, const int* __restrict__ row_lo, const int* __restrict__ row_hi) {
    constexpr int BK = NW2 * 32, STAGE = 2 * TILE2 + 512, VROWS = 32 * KP2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int bh, kblk;
#if A32_DKV_ORDER
    if (!xcd_item_major_decode(blockIdx.x, B * Hq, (S + BK - 1) / BK, bh, kblk)) return;       // heaviest key blocks of all heads first (see above)
#else
    if (!xcd_group_decode(blockIdx.x, B * Hq, (S + BK - 1) / BK, bh, kblk)) return;
#endif
    const int b = bh / Hq, h = bh % Hq, hk = h / (Hq / Hkv);
    const int k0 = kblk * BK, kw = k0 + wave * 32, ki = kw + l31;
    const bf16_t* qb_ = q + (int64_t)b * S * ldq + (int64_t)h * D2;
    const bf16_t* gb_ = gho + (int64_t)b * S * ldg + (int64_t)h * D2;
    const float* lse_b = lse + ((int64_t)b * Hq + h) * S;
    const float* D_b = Dd + ((int64_t)b * Hq + h) * S;
    const int* rlo_b = row_lo ? row_lo + (int64_t)b * S : nullptr;
    const int* rhi_b = row_lo ? row_hi + (int64_t)b * S : nullptr;

    // K fragments of the wave's 32 keys in registers, its V rows in a wave-private LDS block in the tile layout (read in pass 1)
    bf16x8 kf[NK2];
    load_row_frags2(kf, k + (int64_t)b * S * ldk + (int64_t)hk * D2, ldk, ki, S, hi);
    char* sVw = smem + 2 * STAGE + wave * VROWS;
    int* const ivtab = reinterpret_cast<int*>(smem + 2 * STAGE + NW2 * VROWS);   // per-32-query-block interval bounds (only with row intervals)
    stage_block2(v + (int64_t)b * S * ldv + (int64_t)hk * D2, ldv, kw, S, sVw, lane);
    const float c1 = scale * LRP_LOG2E;
    int qbeg = 0, qend = S;
    if (causal) qbeg = (k0 / CT2) * CT2;
    if (window > 0) qend = min(S, k0 + BK - 1 + window);
    if (q_begin > qbeg) qbeg = (q_begin / CT2) * CT2;
    if (rlo_b != nullptr) {
        int qlo, qhi;
        iv_block_table<NW2>(rlo_b, rhi_b, S, k0, BK, wave, lane, ivtab, reinterpret_cast<int*>(smem), qlo, qhi);
        qbeg = max(qbeg, (min(qlo, S) / CT2) * CT2);
        qend = min(qend, qhi);
    }

    auto stage = [&](int qt0, int buf) {
        char* sb = smem + buf * STAGE;
        stage_tile2(qb_, ldq, qt0, S, sb, wave, lane);
        stage_tile2(gb_, ldg, qt0, S, sb + TILE2, wave, lane);
        if (wave == 0) stage_stats(lse_b, qt0, S, sb + 2 * TILE2, lane);
        if (wave == 1) stage_stats(D_b, qt0, S, sb + 2 * TILE2 + 256, lane);
    };
    const uint32_t avw = lds_addr(sVw) + rm_base(lane);              // the wave's V block (fixed)

    sfor<0, 2>([&](auto passc) {
        constexpr int PASS = decltype(passc)::value;                   // 0: dV, 1: dK
        f32x16 acc[ND2];
#pragma unroll
        for (int db = 0; db < ND2; ++db) acc[db] = zero16();
        if (qbeg < qend) stage(qbeg, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        uint32_t arm = lds_addr(smem) + rm_base(lane), at0 = lds_addr(smem) + tr_base(lane, 0), at1 = lds_addr(smem) + tr_base(lane, 1);
        uint32_t ast = lds_addr(smem) + 2 * TILE2 + hi * 16;
        int cur = 0;
        for (int qt0 = qbeg; qt0 < qend; qt0 += CT2) {
            if (qt0 + CT2 < qend) stage(qt0 + CT2, cur ^ 1);
            bool dead = (causal && qt0 + 31 < kw) || qt0 >= S || (window > 0 && qt0 - window >= kw + 31);
            bool iv_mask = false;
            if (rlo_b != nullptr && !dead) {
                int t0_, t1_, t2_, t3_;
                iv_row(ivtab, qt0 >> 5, t0_, t1_, t2_, t3_);
                dead = t1_ <= kw || t0_ >= kw + 32;                       // no query of the tile sees a key of this wave
                iv_mask = !(t2_ <= kw && t3_ >= kw + 32);                // some row's interval ends inside the wave's keys
            }
            if (!dead) {
                // ---- S^T (16 steps x Q fragment); pass 1: dP^T as well (16 steps x {Gho fragment, V fragment})
                f32x16 st = zero16(), dp = zero16();
                bf16x8 fq[3];
                A32_RD128(fq[0], arm, 0);
                { const uint32_t a_ = xor_addr<1 << 5>(arm); A32_RD128(fq[1], a_, 0); }
                sfor<0, NK2>([&](auto ksc) {
                    constexpr int ks = decltype(ksc)::value, cu = ks % 3, nx = (ks + 2) % 3;
                    (void)&fq; (void)&arm;
                    if constexpr (ks + 2 < NK2) {
                        { const uint32_t a_ = xor_addr<(ks + 2) << 5>(arm); A32_RD128(fq[nx], a_, 0); }
                        A32_WAIT(2, "+v"(fq[cu]));
                    } else if constexpr (ks + 1 < NK2) {
                        A32_WAIT(1, "+v"(fq[cu]));
                    } else {
                        A32_WAIT(0, "+v"(fq[cu]));
                    }
                    st = mfma32(fq[cu], kf[ks], st);
                    A32_FENCE();
                });
                if constexpr (PASS == 1) {
                    bf16x8 fg[3], fv[3];
                    A32_RD128(fg[0], arm, TILE2); A32_RD128(fv[0], avw, 0);
                    { const uint32_t a_ = xor_addr<1 << 5>(arm), b_ = xor_addr<1 << 5>(avw); A32_RD128(fg[1], a_, TILE2); A32_RD128(fv[1], b_, 0); }
                    sfor<0, NK2>([&](auto ksc) {
                        constexpr int ks = decltype(ksc)::value, cu = ks % 3, nx = (ks + 2) % 3;
                        (void)&fg; (void)&fv; (void)&arm; (void)&avw;
                        if constexpr (ks + 2 < NK2) {
                            { const uint32_t a_ = xor_addr<(ks + 2) << 5>(arm), b_ = xor_addr<(ks + 2) << 5>(avw); A32_RD128(fg[nx], a_, TILE2); A32_RD128(fv[nx], b_, 0); }
                            A32_WAIT(4, "+v"(fg[cu]), "+v"(fv[cu]));
                        } else if constexpr (ks + 1 < NK2) {
                            A32_WAIT(2, "+v"(fg[cu]), "+v"(fv[cu]));
                        } else {
                            A32_WAIT(0, "+v"(fg[cu]), "+v"(fv[cu]));
                        }
                        dp = mfma32(fg[cu], fv[cu], dp);
                        A32_FENCE();
                    });
                }
                // ---- first transpose-read unit (pass 0: Gho tile, pass 1: Q tile) in flight under the element-wise work
                constexpr int TOFF = (PASS == 0) ? TILE2 : 0;
                u32x2 tt[2][4][2];
                A2_UNIT(tt[0], at0, at1, 0, 0, TOFF);
                A32_FENCE();
                const bool masked = (qt0 + 32 > S) || (causal && qt0 < kw + 31) || (window > 0 && qt0 + 31 - window >= kw) || iv_mask;
                bf16x8 xf[2];                                             // pass 0: P, pass 1: dS (per 16-query group)
                f32x4 sl[2], sd[2];
                sfor<0, 2>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    (void)&sl; (void)&sd; (void)&ast;
                    A32_RD128(sl[0], ast, 64 * j); A32_RD128(sl[1], ast, 64 * j + 32);
                    if constexpr (PASS == 1) {
                        A32_RD128(sd[0], ast, 64 * j + 256); A32_RD128(sd[1], ast, 64 * j + 32 + 256);
                        A32_WAIT(0, "+v"(sl[0]), "+v"(sl[1]), "+v"(sd[0]), "+v"(sd[1]));
                    } else {
                        A32_WAIT(0, "+v"(sl[0]), "+v"(sl[1]));
                    }
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int i = 2 * j + ii, r = 4 * i + e;
                            const float s_raw = st[r];
                            float p = fast_exp2(__builtin_fmaf(s_raw, c1, -(sl[ii][e] * LRP_LOG2E)));
                            if (masked) {
                                const int qi = qt0 + 8 * i + 4 * hi + e;
                                int ivlo = 0, ivhi = S;
                                if (rlo_b != nullptr && qi < S) { ivlo = rlo_b[qi]; ivhi = rhi_b[qi]; }
                                p = ((qi < S) & visible(qi, ki, S, causal, window, ivlo, ivhi)) ? p : 0.f;
                            }
                            if constexpr (PASS == 0) xf[j][4 * ii + e] = (bf16_t)p;
                            else xf[j][4 * ii + e] = (bf16_t)lrp_ds<EXPL>(s_raw, p, dp[r], sd[ii][e], scale, eps_mask, eps_qk);
                        }
                    A32_FENCE();
                });
                // ---- out^T += (Gho | Q)^T x^T: units (j, dbh) = (0,0) (0,1) (1,0) (1,1), next unit's reads issued before this unit's MFMAs
                A2_UNIT(tt[1], at0, at1, 0, 1, TOFF);
                A32_WAIT(8, A2_UNITV(tt[0]));
#pragma unroll
                for (int d4 = 0; d4 < 4; ++d4) acc[d4] = mfma32(join_tr(tt[0][d4][0], tt[0][d4][1]), xf[0], acc[d4]);
                A32_FENCE();
                A2_UNIT(tt[0], at0, at1, 1, 0, TOFF);
                A32_WAIT(8, A2_UNITV(tt[1]));
#pragma unroll
                for (int d4 = 0; d4 < 4; ++d4) acc[4 + d4] = mfma32(join_tr(tt[1][d4][0], tt[1][d4][1]), xf[0], acc[4 + d4]);
                A32_FENCE();
                A2_UNIT(tt[1], at0, at1, 1, 1, TOFF);
                A32_WAIT(8, A2_UNITV(tt[0]));
#pragma unroll
                for (int d4 = 0; d4 < 4; ++d4) acc[d4] = mfma32(join_tr(tt[0][d4][0], tt[0][d4][1]), xf[1], acc[d4]);
                A32_FENCE();
                A32_WAIT(0, A2_UNITV(tt[1]));
#pragma unroll
                for (int d4 = 0; d4 < 4; ++d4) acc[4 + d4] = mfma32(join_tr(tt[1][d4][0], tt[1][d4][1]), xf[1], acc[4 + d4]);
                A32_FENCE();
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            {
                const uint32_t delta = cur ? (uint32_t)-STAGE : (uint32_t)STAGE;
                arm += delta; at0 += delta; at1 += delta; ast += delta;
            }
            cur ^= 1;
        }
        if constexpr (PASS == 0) store_rows2(dv + (int64_t)b * S * lddv + (int64_t)h * D2, lddv, ki, S, acc, 1.f, hi);
        else store_rows2(dk + (int64_t)b * S * lddk + (int64_t)h * D2, lddk, ki, S, acc, EXPL ? 1.f : 0.5f * scale, hi);
    });
}

}  // namespace d256

constexpr size_t A32_LDS_MAX = 160 * 1024;       // gfx950: 160 KiB per workgroup (the dK / dV kernels' interval table grows with S: S <= ~60000 with row intervals)
template <typename K> void set_lds(K kern, size_t bytes) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

}  // namespace attn32

// ---- entry points used by the dispatchers of attention.hip (bf16, d == 256) ------------------------------------------
int lrp_attn32_fwd_d256(const void* q, const void* k, const void* v, void* o, float* lse, int B, int S, int Hq, int Hkv, int64_t ldq,
                        int64_t ldk, int64_t ldv, int64_t ldo, float scale, int causal, int window, int q_begin, const int* row_lo,
                        const int* row_hi, hipStream_t st) {
    using namespace attn32;
    using namespace attn32::d256;
    const size_t lds = 2 * (2 * (size_t)TILE2);
    auto kern = fwd256_kernel;
    LRP_SET_MAX_LDS(kern, lds);
    dim3 grid(xcd_group_grid(B * Hkv, (Hq / Hkv) * ((S + NW2 * 32 - 1) / (NW2 * 32))));
    hipLaunchKernelGGL(kern, grid, dim3(NW2 * 64), lds, st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, lse, S, Hq,
                       Hkv, ldq, ldk, ldv, ldo, scale, causal, window, B, q_begin, row_lo, row_hi);
    return lrp_check_launch();
}

int lrp_attn32_dq_d256(const void* q, const void* k, const void* v, const void* gho, const float* lse, const float* D_, void* dq, int B,
                       int S, int Hq, int Hkv, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldg, int64_t lddq, float scale,
                       float eps_mask, float eps_qk, int causal, int window, int q_begin, const int* row_lo, const int* row_hi,
                       hipStream_t st) {
    using namespace attn32;
    using namespace attn32::d256;
    const size_t lds = 2 * (2 * (size_t)TILE2);
    dim3 grid(xcd_group_grid(B * Hkv, (Hq / Hkv) * ((S + NW2 * 32 - 1) / (NW2 * 32))));
    if (eps_mask != 0.f || eps_qk != 0.f) {
        auto kern = dq256_kernel<true>;
        LRP_SET_MAX_LDS(kern, lds);
        hipLaunchKernelGGL(kern, grid, dim3(NW2 * 64), lds, st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)gho,
                           lse, D_, (bf16_t*)dq, S, Hq, Hkv, ldq, ldk, ldv, ldg, lddq, scale, eps_mask, eps_qk, causal, window, B,
                           q_begin, row_lo, row_hi);
    } else {
        auto kern = dq256_kernel<false>;
        LRP_SET_MAX_LDS(kern, lds);
        hipLaunchKernelGGL(kern, grid, dim3(NW2 * 64), lds, st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)gho,
                           lse, D_, (bf16_t*)dq, S, Hq, Hkv, ldq, ldk, ldv, ldg, lddq, scale, eps_mask, eps_qk, causal, window, B,
                           q_begin, row_lo, row_hi);
    }
    return lrp_check_launch();
}

int lrp_attn32_dkv_d256(const void* q, const void* k, const void* v, const void* gho, const float* lse, const float* D_, void* dk,
                        void* dv, int B, int S, int Hq, int Hkv, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldg, int64_t lddk,
                        int64_t lddv, float scale, float eps_mask, float eps_qk, int causal, int window, int q_begin,
                        const int* row_lo, const int* row_hi, hipStream_t st) {
    using namespace attn32;
    using namespace attn32::d256;
    const size_t lds = 2 * (2 * (size_t)TILE2 + 512) + (size_t)NW2 * 32 * KP2 + (row_lo ? (size_t)((S + 31) / 32) * 16 : 0);   // + the interval table
    if (lds > A32_LDS_MAX) return LRP_ESHAPE;
    dim3 grid(xcd_group_grid(B * Hq, (S + NW2 * 32 - 1) / (NW2 * 32)));
    if (eps_mask != 0.f || eps_qk != 0.f) {
        auto kern = dkv256_kernel<true>;
        LRP_SET_MAX_LDS(kern, A32_LDS_MAX);
        hipLaunchKernelGGL(kern, grid, dim3(NW2 * 64), lds, st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)gho,
                           lse, D_, (bf16_t*)dk, (bf16_t*)dv, S, Hq, Hkv, ldq, ldk, ldv, ldg, lddk, lddv, scale, eps_mask, eps_qk,
                           causal, window, B, q_begin, row_lo, row_hi);
    } else {
        auto kern = dkv256_kernel<false>;
        LRP_SET_MAX_LDS(kern, A32_LDS_MAX);
        hipLaunchKernelGGL(kern, grid, dim3(NW2 * 64), lds, st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)gho,
                           lse, D_, (bf16_t*)dk, (bf16_t*)dv, S, Hq, Hkv, ldq, ldk, ldv, ldg, lddk, lddv, scale, eps_mask, eps_qk,
                           causal, window, B, q_begin, row_lo, row_hi);
    }
    return lrp_check_launch();
}

// ---- entry points used by the dispatchers of attention.hip (bf16, d in {64, 96, 128}) ----------------------------------
#define A32_FOR_DH(d, ...)                                     \
    switch (d) {                                               \
        case 64: { constexpr int DH = 64; __VA_ARGS__ } break; \
        case 96: { constexpr int DH = 96; __VA_ARGS__ } break; \
        case 128: { constexpr int DH = 128; __VA_ARGS__ } break; \
        default: return LRP_ESHAPE;                            \
    }

int lrp_attn32_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int B, int S, int Hq, int Hkv, int d, int64_t ldq,
                   int64_t ldk, int64_t ldv, int64_t ldo, float scale, int causal, int window, int q_begin, const int* row_lo,
                   const int* row_hi, hipStream_t st) {
    using namespace attn32;
    const size_t lds = 2 * (2 * (size_t)TILE);
    dim3 grid(xcd_group_grid(B * Hkv, (Hq / Hkv) * ((S + NWQ * 32 - 1) / (NWQ * 32))));
    A32_FOR_DH(d, {
        auto kern = fwd_kernel<DH>;
        LRP_SET_MAX_LDS(kern, lds);
        hipLaunchKernelGGL(kern, grid, dim3(NWQ * 64), lds, st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, lse, S, Hq,
                           Hkv, ldq, ldk, ldv, ldo, scale, causal, window, B, q_begin, row_lo, row_hi);
    })
    return lrp_check_launch();
}

// o != NULL: D is COMPUTED from (gho, o) and written to Dout (then D_ is not read)
int lrp_attn32_dq(const void* q, const void* k, const void* v, const void* gho, const float* lse, const float* D_, void* dq, int B,
                  int S, int Hq, int Hkv, int d, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldg, int64_t lddq, float scale,
                  float eps_mask, float eps_qk, int causal, int window, int q_begin, const int* row_lo, const int* row_hi,
                  hipStream_t st, const void* o, int64_t ldo, float* Dout, const float* cos_t, const float* sin_t) {
    using namespace attn32;
    const size_t lds = 2 * (2 * (size_t)TILE);
    dim3 grid(xcd_group_grid(B * Hkv, (Hq / Hkv) * ((S + NWQ * 32 - 1) / (NWQ * 32))));
    const bool expl = eps_mask != 0.f || eps_qk != 0.f;
#define A32_LAUNCH_DQ(EX)                                                                                                                  \
    {                                                                                                                                       \
        auto kern = dq_kernel<EX, DH>;                                                                                                      \
        LRP_SET_MAX_LDS(kern, lds);                                                                                                         \
        hipLaunchKernelGGL(kern, grid, dim3(NWQ * 64), lds, st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)gho, \
                           lse, D_, (bf16_t*)dq, S, Hq, Hkv, ldq, ldk, ldv, ldg, lddq, scale, eps_mask, eps_qk, causal, window, B,         \
                           q_begin, row_lo, row_hi, (const bf16_t*)o, ldo, Dout, cos_t, sin_t);                                            \
    }
    A32_FOR_DH(d, { if (expl) A32_LAUNCH_DQ(true) else A32_LAUNCH_DQ(false) })
#undef A32_LAUNCH_DQ
    return lrp_check_launch();
}

int lrp_attn32_dkv(const void* q, const void* k, const void* v, const void* gho, const float* lse, const float* D_, void* dk,
                   void* dv, int B, int S, int Hq, int Hkv, int d, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldg, int64_t lddk,
                   int64_t lddv, float scale, float eps_mask, float eps_qk, int causal, int window, int q_begin,
                   const int* row_lo, const int* row_hi, hipStream_t st) {
    using namespace attn32;
    const size_t lds = 2 * (2 * (size_t)TILE + 512) + (size_t)NW * 32 * KP + (row_lo ? (size_t)((S + 31) / 32) * 16 : 0);   // + the interval table
    if (lds > A32_LDS_MAX) return LRP_ESHAPE;
    dim3 grid(xcd_group_grid(B * Hq, (S + 255) / 256));
    const bool expl = eps_mask != 0.f || eps_qk != 0.f;
#define A32_LAUNCH_DKV(EX, IVF)                                                                                                        \
    {                                                                                                                                   \
        auto kern = dkv_kernel<EX, DH, IVF>;                                                                                            \
        LRP_SET_MAX_LDS(kern, A32_LDS_MAX);                                                                                             \
        hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)gho,   \
                           lse, D_, (bf16_t*)dk, (bf16_t*)dv, S, Hq, Hkv, ldq, ldk, ldv, ldg, lddk, lddv, scale, eps_mask, eps_qk,     \
                           causal, window, B, q_begin, row_lo, row_hi);                                                                \
    }
    const bool iv = row_lo != nullptr;
    A32_FOR_DH(d, {
        if (expl) { if (iv) A32_LAUNCH_DKV(true, true) else A32_LAUNCH_DKV(true, false) }
        else { if (iv) A32_LAUNCH_DKV(false, true) else A32_LAUNCH_DKV(false, false) }
    })
#undef A32_LAUNCH_DKV
    return lrp_check_launch();
}
