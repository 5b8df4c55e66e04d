// gemm_pp.hip -- the big bf16 NT GEMMs of the Linear eps-rule (K1 at M = B*S rows): C[M,N] = A[M,K] . B[N,K]^T (+bias),
// fp32 accumulate, as an 8-wave PING-PONG kernel on v_mfma_f32_16x16x32_bf16.
//
// What the measurements of rounds 1-3 say (profiles/r03_gemm_experiments.txt):
//   * Every one of these GEMMs runs at the 1400 W package power cap; the clock the chip settles at is the free variable.  What counts is
//     (matrix-pipe occupancy) x (clock), i.e. ENERGY per FLOP, not latency hiding alone.
//   * A pure v_mfma_f32_16x16x32_bf16 stream on random operands settles at 2.13 GHz / 2.06 PFLOP/s, a v_mfma_f32_32x32x16_bf16 stream at
//     1.84 GHz / 1.82 PFLOP/s (tools/micro/mfma_power.hip): the 16x16 form is 13 % cheaper per FLOP.  LDS fragment reads cost ~8 % per
//     0.75 KiB read per 32 KFLOP (tools/micro/lds_power.hip); the staging traffic (L2 -> LDS, fabric, HBM) another 12-25 %.
//   * Lock-step forms (4, 8 or 16 waves that all read LDS, then all issue MFMAs) stop at ~63 % occupancy.  Here the 8 waves of a
//     workgroup form two groups of four (one wave of each group per SIMD) that run HALF A PHASE APART: while group X issues 32
//     back-to-back MFMAs (a bare stream: one extra VALU / branch state between MFMAs costs 40-90 cycles), group Y reads its next
//     fragments out of LDS and issues the direct-to-LDS loads of a later K tile; then they swap.  The skeleton {barrier, MFMA burst,
//     barrier} alone runs at 100 % of the pipe (tools/micro/mfma32_pp.hip); this kernel reaches 94-96 % inside the K loop (the
//     single-wave issue limit of the 16x16 form) and 88-91 % over a whole tile.
//
// Geometry: 256 x 256 tile, K tile = 64 elements (128 B per row: full cache lines per row visit), 512 threads.  Wave w: group g = w >> 2
// (rows g*128 .. +127 of the tile), column block wc = w & 3 (columns wc*64 .. +63): 128 x 64 per wave = 2 row halves (a) x {4 x 4 MFMA
// tiles of 16 x 16} = 128 accumulator registers.  One phase = one row half a of one K tile: L = 8 A-fragment reads (+ 8 B-fragment
// reads when a = 0; the B fragments are kept for a = 1), M = 32 MFMAs (2 k-steps x 4 x 4 tiles).
//
// LDS (128 KiB): [A buf0][A buf1][B buf0][B buf1], 32 KiB each = 256 rows x 128 B.  Row r holds its eight 16-byte chunks at position
// chunk ^ (r & 7): a 16-row fragment read (lane l: row l & 15, chunk 4 ks + (l >> 4)) is conflict-free under the ds_read_b128 lane
// groups of the hardware (tests/test_layout_maps_cpu.py).  Staging pieces are buffer_load_dwordx4 .. lds: one wave instruction writes a
// lane-linear 1-KiB piece (8 rows), so the swizzle is applied to the per-lane SOURCE chunk; the row and K position sit in the scalar
// offset (one per-lane offset register per operand, no 64-bit address arithmetic: a global_load_lds piece costs ~80 cycles of the L
// phase, a buffer piece ~35); rows past M / N read as zero through num_records.
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
// Operand forms.  NT: B is [N, K], K contiguous (a weight in its forward layout).  NN: B is [K, N], N contiguous -- the SAME stored weight
// W [out, in] used as the dgrad operand (c = s W contracts over W's rows), so no W^T copy exists: a staging piece is 2 contraction rows x
// 512 B of the tile's 256 output columns, the LDS image is [64 contraction rows][512 B] with 16-byte chunk ^ 2 ((row & 3) + 4 ((row >> 3) & 1)),
// and the MFMA operand (8 consecutive contraction indices of one output column) is gathered by two ds_read_b64_tr_b16 (4 rows each).
// Split-K: blockIdx.y owns a contiguous range of K tiles and writes an fp32 partial slab; used for skinny problems (M <= 256 rows), where
// the weight is streamed exactly once and the tile count alone would leave most CUs idle (lrp_gemm_skinny, reduced by a second kernel).
// Barriers: L | barrier | M | barrier per phase; group 1 executes ONE extra barrier up front, which puts it half a phase behind
// for the whole kernel (and group 0 one at the very end to balance the count).
// Round 5 -- PERSISTENT tile walk (PP_PERSIST, default on): the launch is min(tiles, CUs) workgroups; workgroup w computes the tiles w, w + grid,
// w + 2 grid, ... (the same tile every CU got from the dispatcher before: block b runs on XCD b % 8 and the XCD remap keeps the 32 tiles a XCD
// works on at any time a compact group).  What it buys: the K loop's prologue loads of the NEXT tile (14 LDS-DMA pieces per wave into the two
// buffers the finished K loop no longer reads) are issued BEFORE the epilogue's stores, so their HBM / L2 latency -- paid by all 256 CUs at once
// at every round boundary of the non-persistent launch, together with the dispatch of a fresh 512-thread, 128-KiB workgroup -- runs under the
// store phase (and, in the fused down-projection dgrad, under its gu reads).  LDS safety: group 0 reaches that point one interval ahead of group 1,
// which by then has finished every read of the shared B rows (B fragments are read in phase (t, 0) only) and otherwise reads only its OWN A rows,
// which only its own waves re-stage.  vmcnt counts loads and stores in issue order on gfx9, so "the first two staging units of the new tile have
// landed" is vmcnt(8 + stores issued since) -- the stores themselves are never waited for.
// Measured (profiles/r05_gemm_experiments.txt): +0.8 % on the seven Llama shapes, +1.1 % on the short-K SigLIP / Gemma-3 shapes in the isolated A/B,
// within noise in situ -- the per-tile prologue / dispatch cost the round-4 notes hoped to recover (3-15 %) is ~1 %.  A staggered first round
// (workgroups starting up to 17 us apart to de-phase the CUs' epilogue bursts) changed nothing and was removed: the fused epilogues are bound by
// their own VALU issue (see gated_bwd_pair_bf16 in common.hpp), not by a shared HBM burst.
#include "common.hpp"

#ifndef PP_PERSIST
#define PP_PERSIST 1
#endif

namespace {

constexpr int PP_KT = 64;                         // K elements per K tile
constexpr int PP_OPND = 256 * 128;                // one operand of one buffer (bytes)

typedef __attribute__((address_space(3))) void* pp_lds_ptr_t;

// Counted waits on the vector-memory counter as the BUILTIN, not as inline asm: the compiler's own wait insertion (SIInsertWaitcnts) reads an
// s_waitcnt it finds in the instruction stream and updates its picture of what is still in flight; an inline-asm wait is opaque to it, so
// every load the epilogues issue (row scales, residual / coefficient tiles) stayed "pending" in its model across the persistent tile loop and
// it protected the K loop's fragment registers -- which those loads had re-used -- with a vmcnt(0) at the TOP OF THE K LOOP, draining the
// staging pipeline every second K tile (round 6: found in the EPI 4 instantiation, -4 % on the K = 28672 gate/up dgrad; round 5's EPI 2 had the
// same).  gfx9 encoding: vmcnt = imm[3:0] | imm[15:14] << 4, expcnt imm[6:4] = 7 and lgkmcnt imm[11:8] = 15 (no wait on those).
#define PP_VMWAIT(N)                                                                              \
    do {                                                                                          \
        asm volatile("" ::: "memory");                                                            \
        __builtin_amdgcn_s_waitcnt(((N) & 15) | 0x70 | 0xF00 | (((N) >> 4) << 14));               \
        asm volatile("" ::: "memory");                                                            \
    } while (0)
#define PP_DSRD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define PP_DSTR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))

// fused gated-MLP epilogues (EPI 1 / 2, see the end of the kernel); EPI 0 = plain (+ bias)
struct PPEpi {
    bf16_t* c2;            // EPI 1: m [M, I]  (C is then the COEFFICIENT stash [M, 2 I], see below)
    int64_t ldc2;
    const bf16_t* gu;      // EPI 2: the coefficient stash [M, 2 I] the gate/up forward (EPI 1) left
    int64_t ldgu;
    float eps_g, eps_lin;
    int act;
    // round 5, RMSNorm folded into the GEMMs around it (lrp_gemm_*_rs / _res entry points below):
    const float* rs;       // RS: per-row scale of the accumulators (rstd of the consumer's input row), applied before everything else
    const bf16_t* res;     // EPI 3 / 4: residual [M, N] added to the (scaled) accumulators
    int64_t ldres;
    float* ssq;            // EPI 3: partial sums of squares of the bf16-rounded output rows, [N / 64][ldssq] (one 64-column block per wave)
    int64_t ldssq;
    // round 6, EPI 5: RoPE on the q / k columns of the fused QKV forward (lrp_gemm_nt_rs_rope)
    const float* cos;      // fp32 [>= seq, 128] tables (rotate-half convention: both halves of a row equal; only the first is read)
    const float* sin;
    int seq;               // rows per prompt: the position of output row m is m % seq
    int rope_cols;         // columns [0, rope_cols) are q / k heads of 128; the rest (v) pass through
};

// SK ("skinny"): the problem has ONE row of tiles and fewer than 241 rows (the HBM-bound regime of the Linear eps-rule, M <~ 160): 16-row
// blocks of the 256-row tile that lie past M are not multiplied -- the full tile's 2 x 256 x 256 x 64 FLOP per K tile would make a
// 16-row problem MATRIX-pipe-bound (13.7 us per workgroup for a [14336, 4096] weight) under the 15-us weight stream it serves.
// EPI 3: C = bf16(acc + res), ssq partials (the residual add + the sum of squares of RMSNorm in the producing GEMM);  EPI 4: C = bf16(rs acc + res)
// (RMSNorm's identity-rule backward + the residual gradient in the dgrad GEMM);  RS with EPI 0 / 1: acc scaled by rs[row] first (the norm's
// 1 / rms applied to the consumer's OUTPUT rows: (rstd x) W^T = rstd (x W^T); the norm's weight is folded into W by the host).
// EPI 5 (NT, RS): the fused QKV forward with RoPE in the epilogue (K5 of SURVEY.md 2.3; HF apply_rotary_pos_emb, ref
// lxt/explicit/models/llama.py:226-260 -- constant cos / sin: the reference leaves it un-patched).  A rotate-half pair (c, c + 64) of a 128-wide
// head sits 64 columns apart, i.e. in TWO waves of the standard tile mapping; here the wave reads its B fragments from the rows
// {32 w .. 32 w + 31} and {64 + 32 w .. 64 + 32 w + 31} of its head (w = wave & 1) -- two immediates of the fragment read change, nothing else --
// so that column tiles j and j + 2 of ONE lane are a rotation pair, and stores its two 32-column segments where they belong.  Weights,
// activations and every other kernel keep the standard head-dim order.
template <typename TO, bool NN, int EPI, int ACT, bool SK = false, bool LEAN = false, bool RS = false>
__global__ __launch_bounds__(512, 2) void gemm_pp_kernel(
    const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, TO* __restrict__ C, const bf16_t* __restrict__ bias,
    int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, int tiles_m, int tiles_n, int kt_per_split, int64_t slab_stride, PPEpi ep) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave >> 2, wc = wave & 3;
    // this workgroup's range of K tiles (split-K: blockIdx.y), host guarantees >= 2 tiles per split
    const int nkt_all = K / PP_KT;
    const int kt0 = blockIdx.y * kt_per_split;
    const int nkt = min(kt_per_split, nkt_all - kt0);
    C += (int64_t)blockIdx.y * slab_stride;
    const int ntiles = tiles_m * tiles_n;
    int tile = blockIdx.x;
    int m0, n0;                                                        // the tile whose K loop runs / ran last (epilogue coordinates)
    {
        int tm, tn;
        grouped_tile(xcd_remap(tile, ntiles), tiles_m, tiles_n, tm, tn);
        m0 = tm * 256; n0 = tn * 256;
    }
    // ---- staging (buffer_load_dwordx4 .. lds, 1 KiB per wave instruction; rows / columns past the operand read as zero)
    // A (both forms): piece = 8 rows x 128 B; lane l -> row (l >> 3), LDS position (l & 7), source chunk position ^ (row & 7).  A pieces of
    // this wave: rows g*128 + a*64 + wc*16 + 8 p (a = unit half, p = 0, 1).
    const int prow = lane >> 3, pslot = lane & 7;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)((int64_t)M * lda * 2), 0x00020000);
    const int voA = (int)(prow * lda * 2) + ((pslot ^ prow) << 4);
    int soA[2][2];
    auto set_soA = [&](int m0_) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int p = 0; p < 2; ++p) soA[a][p] = (int)((int64_t)(m0_ + g * 128 + a * 64 + wc * 16 + 8 * p) * lda * 2) + kt0 * 128;
    };
    set_soA(m0);
    // B, NT form: piece = 8 rows of B x 128 B, rows wave*32 + 8 p (p = 0..3), same swizzle as A.
    // B, NN form: piece = 2 contraction rows x 512 B (the tile's 256 output columns), contraction rows 8 wave + 2 p + (l >> 5),
    //             LDS position (l & 31), source chunk position ^ f(row), f(row) = 2 ((row & 3) + 4 ((row >> 3) & 1)).
    const int64_t brows = NN ? (int64_t)K : (int64_t)N;
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (int)(brows * ldb * 2), 0x00020000);
    int voB[2], soB[4];
    if constexpr (NN) {
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            const int r = 2 * pp + (lane >> 5);                          // row & 3 of the piece's row (pieces p and p + 2 agree)
            const int f = 2 * ((r & 3) + 4 * (wave & 1));
            voB[pp] = (int)((lane >> 5) * ldb * 2) + (((lane & 31) ^ f) << 4);
        }
    } else {
        voB[0] = voB[1] = (int)(prow * ldb * 2) + ((pslot ^ prow) << 4);
    }
    auto set_soB = [&](int n0_) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
            soB[p] = NN ? (int)(((int64_t)(kt0 * PP_KT + 8 * wave + 2 * p) * ldb + n0_) * 2) : (int)((int64_t)(n0_ + wave * 32 + 8 * p) * ldb * 2) + kt0 * 128;
    };
    set_soB(n0);
    const int bstep = NN ? (int)(PP_KT * ldb * 2) : 128;                // bytes per K tile along B
    char* const ldsA = smem + (g * 128 + wc * 16) * 128;               // + buf*PP_OPND + a*8192 + p*1024
    char* const ldsB = smem + 2 * PP_OPND + (NN ? wave * 8 * 512 : wave * 32 * 128);      // + buf*PP_OPND + p*1024
    auto stage_A = [&](int a, int kt, int buf) {
#pragma unroll
        for (int p = 0; p < 2; ++p)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (pp_lds_ptr_t)(ldsA + buf * PP_OPND + a * 8192 + p * 1024), 16, voA,
                                                     soA[a][p] + kt * 128, 0, 0);
    };
    auto stage_B = [&](int kt, int buf) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (pp_lds_ptr_t)(ldsB + buf * PP_OPND + p * 1024), 16, voB[p & 1],
                                                     soB[p] + kt * bstep, 0, 0);
    };

    // ---- fragment addresses.  A (and NT B): row (l & 15) of a 16-row block, chunk (4 ks + (l >> 4)) ^ (l & 7).
    const int hi = lane >> 4;
    const unsigned rowA = (unsigned)((g * 128 + (lane & 15)) * 128);
    unsigned cA[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) cA[ks] = rowA + (unsigned)(((4 * ks + hi) ^ (lane & 7)) << 4);
    // NT: cB[ks] as cA over the B rows wc*64 + 16 j.  NN: cB[j] = address of this lane's 8-byte segment for the transpose read of column tile
    // j: contraction row 8 hi + (i16 >> 2) (+ 32 ks + 4 h through immediates), 16-byte chunk (8 wc + 2 j + ((i16 & 3) >> 1)) ^ f, byte 8 (i16 & 1)
    unsigned cB[4];
    if constexpr (NN) {
        const int i16 = lane & 15;
        const int f = 2 * (((i16 >> 2) & 3) + 4 * (hi & 1));
#pragma unroll
        for (int j = 0; j < 4; ++j)
            cB[j] = (unsigned)(2 * PP_OPND + (8 * hi + (i16 >> 2)) * 512 + (((8 * wc + 2 * j + ((i16 & 3) >> 1)) ^ f) << 4) + 8 * (i16 & 1));
    } else {
        const unsigned rowB = (unsigned)(2 * PP_OPND + ((EPI == 5 ? (wc >> 1) * 128 + (wc & 1) * 32 : wc * 64) + (lane & 15)) * 128);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) cB[ks] = rowB + (unsigned)(((4 * ks + hi) ^ (lane & 7)) << 4);
        cB[2] = cB[3] = 0;
    }

    // SK: live 16-row blocks of this wave's two row halves (wave-uniform: g comes from readfirstlane)
    int nb[2] = {4, 4};
    if constexpr (SK) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int r = (M - (m0 + g * 128 + a * 64) + 15) >> 4;
            nb[a] = r < 0 ? 0 : (r > 4 ? 4 : r);
        }
    }
    f32x4 acc[2][4][4];
    bf16x8 fa[4][2];                                                    // [16-row block][k-step]
    union BFrag { bf16x8 v; bf16x4 h[2]; } fb[4][2];                    // [16-column tile][k-step]; NN: two 4-row transpose reads each

    // ---- prologue of a tile: V0(0) V1(0) V2(0) V0(1) V1(1); V0(0), V1(0) landed = all but the newest 8 pieces
    auto issue_prologue = [&]() { stage_A(0, 0, 0); stage_B(0, 0); stage_A(1, 0, 0); stage_A(0, 1, 1); stage_B(1, 1); };
    issue_prologue();
    PP_VMWAIT(8);
    // persistent walk: one iteration per tile (PP_PERSIST 0, split-K slabs and the timeline build: exactly one)
    for (;;) {
    const int em0 = m0, en0 = n0;                                     // this tile = the epilogue's tile (m0 / n0 move on to the next one after the K loop)
    // epilogue coordinates.  D = mfma(Bfrag, Afrag): lane l holds C[m = .. + 16 i + (l & 15)][n = .. + 16 j + 4 (l >> 4) + e] in acc[a][i][j][e]
    const int mrow = em0 + g * 128 + (lane & 15);
    const int ncol = en0 + wc * 64;
    bool full = (em0 + 256 <= M) && (en0 + 256 <= N) && ((ldc & 7) == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0);
    if constexpr (EPI == 1) full = full && ((ep.ldc2 & 7) == 0) && ((reinterpret_cast<uintptr_t>(ep.c2) & 15) == 0);
    if constexpr (EPI == 2) full = full && ((ep.ldgu & 7) == 0) && ((reinterpret_cast<uintptr_t>(ep.gu) & 15) == 0);
    if constexpr (EPI == 3 || EPI == 4) full = full && ((ep.ldres & 7) == 0) && ((reinterpret_cast<uintptr_t>(ep.res) & 15) == 0);
    // the epilogue's first operand loads go out NOW, a whole K loop ahead of their use: the row scales (8 registers) and, EPI 3 / 4, the residual
    // of the first two row blocks (16 registers).  Requested at the start of the epilogue they cost their full latency per tile (~2 us, measured
    // +29 us on the 14-tile-per-CU gate/up forward); behind the next tile's staging pieces they would not return before those have landed.
    float rsv[8];
    auto load_rs = [&]() {
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const int gm = mrow + (b >> 2) * 64 + (b & 3) * 16;
            rsv[b] = ep.rs[gm < M ? gm : M - 1];             // (rows past M are never stored; an unconditional load keeps the control flow flat)
        }
    };
    if constexpr (RS && EPI != 4) load_rs();
    const int epoff = 16 * (hi & 1) + 8 * (hi >> 1);
    u32x4 rpre[4][2];                                                  // residual of row block k in rpre[k & 3]
    auto issue_res = [&](int b, u32x4 (&d)[2]) {
        const bf16_t* p = ep.res + (int64_t)(mrow + (b >> 2) * 64 + (b & 3) * 16) * ep.ldres + ncol + epoff;
        d[0] = *reinterpret_cast<const u32x4*>(p);
        d[1] = *reinterpret_cast<const u32x4*>(p + 32);
    };
    if constexpr (EPI == 3) {
        if (full) { issue_res(0, rpre[0]); issue_res(1, rpre[1]); }
    }
    // EPI 5: cos / sin of row block k in tpre[k & 1] = {cos tile 0, cos tile 1, sin tile 0, sin tile 1} of the lane's 2 x 4 table columns
    const int w5 = wc & 1, hb5 = (wc >> 1) * 128;
    const bool rot5 = (EPI == 5) && en0 < ep.rope_cols;              // q / k tile (tile-uniform); v tiles only take the row scale
    f32x4 tpre[2][4];
    // (seq is a multiple of 16 -- lrp_gemm_nt_rs_rope_ok -- so the 16 rows of a row block share one prompt and their positions are
    // consecutive: the block's first position is WAVE-UNIFORM, scalar arithmetic incl. the modulo, and the lane part of the address one register)
    const int lane_tab = (lane & 15) * 128 + 32 * w5 + 4 * hi;
    auto issue_tab = [&](int b, f32x4 (&d)[4]) {
        const int pos = (em0 + g * 128 + (b >> 2) * 64 + (b & 3) * 16) % ep.seq;
        const float* pc = ep.cos + (int64_t)pos * 128;
        const float* ps = ep.sin + (int64_t)pos * 128;
        d[0] = *reinterpret_cast<const f32x4*>(pc + lane_tab);
        d[1] = *reinterpret_cast<const f32x4*>(pc + lane_tab + 16);
        d[2] = *reinterpret_cast<const f32x4*>(ps + lane_tab);
        d[3] = *reinterpret_cast<const f32x4*>(ps + lane_tab + 16);
    };
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[a][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_s_barrier();
    if (g == 1) __builtin_amdgcn_s_barrier();                          // the half-phase offset between the two groups

#define PP_READ_A(BUF, AH)                                                                        \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                            \
        PP_DSRD(fa[0][ks], cA[ks], (BUF) * PP_OPND + (AH) * 8192);                                \
        PP_DSRD(fa[1][ks], cA[ks], (BUF) * PP_OPND + (AH) * 8192 + 2048);                         \
        PP_DSRD(fa[2][ks], cA[ks], (BUF) * PP_OPND + (AH) * 8192 + 4096);                         \
        PP_DSRD(fa[3][ks], cA[ks], (BUF) * PP_OPND + (AH) * 8192 + 6144);                         \
    }
#define PP_READ_B(BUF)                                                                            \
    if constexpr (NN) {                                                                           \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                           \
            PP_DSTR(fb[j][0].h[0], cB[j], (BUF) * PP_OPND);                                       \
            PP_DSTR(fb[j][0].h[1], cB[j], (BUF) * PP_OPND + 2048);                                \
            PP_DSTR(fb[j][1].h[0], cB[j], (BUF) * PP_OPND + 16384);                               \
            PP_DSTR(fb[j][1].h[1], cB[j], (BUF) * PP_OPND + 16384 + 2048);                        \
        }                                                                                         \
    } else {                                                                                      \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                        \
            PP_DSRD(fb[0][ks].v, cB[ks], (BUF) * PP_OPND);                                        \
            PP_DSRD(fb[1][ks].v, cB[ks], (BUF) * PP_OPND + 2048);                                 \
            PP_DSRD(fb[2][ks].v, cB[ks], (BUF) * PP_OPND + (EPI == 5 ? 8192 : 4096));             \
            PP_DSRD(fb[3][ks].v, cB[ks], (BUF) * PP_OPND + (EPI == 5 ? 10240 : 6144));            \
        }                                                                                         \
    }
#define PP_WAIT_A() asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[1][0]), "+v"(fa[1][1]),   \
                                 "+v"(fa[2][0]), "+v"(fa[2][1]), "+v"(fa[3][0]), "+v"(fa[3][1]) :: "memory")
#define PP_WAIT_B() asm volatile("" : "+v"(fb[0][0].v), "+v"(fb[0][1].v), "+v"(fb[1][0].v), "+v"(fb[1][1].v),                \
                                 "+v"(fb[2][0].v), "+v"(fb[2][1].v), "+v"(fb[3][0].v), "+v"(fb[3][1].v))
#define PP_MMA(AH)                                                                                \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                              \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                             \
            if (!SK || i < nb[AH])                                                                \
                _Pragma("unroll") for (int j = 0; j < 4; ++j)                                     \
                    acc[AH][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j][ks].v, fa[i][ks], acc[AH][i][j], 0, 0, 0);
#define PP_FENCE() __builtin_amdgcn_sched_barrier(0)

    // one K tile out of buffer BUF (compile-time); t is the running K-tile index.  The M phases are BARE MFMA streams.
#define PP_KTILE(BUF)                                                                             \
    {                                                                                             \
        /* ---- phase (t, 0): L */                                                                \
        PP_READ_B(BUF) PP_READ_A(BUF, 0)                                                          \
        if (t + 1 < nkt) { stage_A(1, t + 1, (BUF) ^ 1); PP_VMWAIT(8); }   \
        else PP_VMWAIT(0);                                     \
        PP_WAIT_A(); PP_WAIT_B(); PP_FENCE();                                                     \
        __builtin_amdgcn_s_barrier(); PP_FENCE();                                                 \
        /* ---- M */                                                                              \
        __builtin_amdgcn_s_setprio(1); PP_MMA(0) __builtin_amdgcn_s_setprio(0); PP_FENCE();       \
        __builtin_amdgcn_s_barrier(); PP_FENCE();                                                 \
        /* ---- phase (t, 1): L */                                                                \
        PP_READ_A(BUF, 1)                                                                         \
        if (t + 2 < nkt) { stage_A(0, t + 2, BUF); stage_B(t + 2, BUF); PP_VMWAIT(8); }   \
        else if (t + 1 < nkt) PP_VMWAIT(2);                    \
        PP_WAIT_A(); PP_FENCE();                                                                  \
        __builtin_amdgcn_s_barrier(); PP_FENCE();                                                 \
        /* ---- M */                                                                              \
        __builtin_amdgcn_s_setprio(1); PP_MMA(1) __builtin_amdgcn_s_setprio(0); PP_FENCE();       \
        __builtin_amdgcn_s_barrier(); PP_FENCE();                                                 \
    }

    int t = 0;
    for (; t + 1 < nkt; t += 2) {
        PP_KTILE(0)
        ++t;
        PP_KTILE(1)
        --t;
    }
    if (t < nkt) PP_KTILE(0)
    // ---- the next tile of this workgroup: its first staging units go out NOW, ahead of the epilogue's memory traffic
    bool has_next = false;
    // the epilogue's first operands were requested at the top of the tile (below); make them architecturally "used" HERE, ahead of the next
    // tile's staging pieces: every load of the K loop has been waited for by now, so the compiler's own wait at this point is free, and
    // none is left to place behind the pieces
    if constexpr (RS && EPI != 4) {
#pragma unroll
        for (int b = 0; b < 8; ++b) asm volatile("" : "+v"(rsv[b]));
    }
    if constexpr (EPI == 3) {
        asm volatile("" : "+v"(rpre[0][0]), "+v"(rpre[0][1]), "+v"(rpre[1][0]), "+v"(rpre[1][1]));
    }
    if constexpr (EPI == 5) {        // (the tables of the first two row blocks: requested here, ahead of the next tile's staging pieces, like EPI 4's operands)
        issue_tab(0, tpre[0]); issue_tab(1, tpre[1]);              // (v tiles load them too and ignore them: flat control flow, no spills)
    }
    // EPI 2: the coefficient loads of the first row block, likewise ahead of the staging pieces.  Lane (row l & 15, hi) reads, for column tile j,
    // the 16 bytes {cg x 4 | cu x 4} of ITS four intermediate indices ncol + 16 j + 4 hi .. + 3: the stash is in accumulator order (no cross-lane
    // exchange on either side), 64 contiguous bytes per row and wave instruction
    u32x4 gpre[4][4];                                                  // coefficient quads of row block k in gpre[k & 3]
    auto issue_gu = [&](int b, u32x4 (&d)[4]) {
        const bf16_t* p = ep.gu + (int64_t)(mrow + (b >> 2) * 64 + (b & 3) * 16) * ep.ldgu + 2 * (int64_t)ncol + 8 * hi;
#pragma unroll
        for (int k = 0; k < 4; ++k) d[k] = *reinterpret_cast<const u32x4*>(p + 32 * k);
    };
    if constexpr (EPI == 2) {
        if (full) issue_gu(0, gpre[0]);                           // (both blocks: 32 registers more than the kernel has at this point -- spills)
    }
    if constexpr (EPI == 4) {       // (NN form: no 24 registers to spare across the K loop -- requested here, ahead of the staging pieces; two tiles per CU)
        load_rs();
        if (full) { issue_res(0, rpre[0]); issue_res(1, rpre[1]); }
    }
#if PP_PERSIST
    if (tile + (int)gridDim.x < ntiles) {
        tile += gridDim.x;
        int tm, tn;
        grouped_tile(xcd_remap(tile, ntiles), tiles_m, tiles_n, tm, tn);
        m0 = tm * 256; n0 = tn * 256;
        has_next = true;
        set_soA(m0);
        set_soB(n0);
        issue_prologue();
    }
#endif
    {
    // ---- epilogue
    const bool vec4 = ((ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0);
    bool done = false;
    if constexpr (EPI == 2) {
        // ---- gated-MLP backward rule in the down-projection dgrad's epilogue, full tiles (round 6: COEFFICIENT form).  The accumulators are Gm
        // for 64 intermediate indices; the gate/up forward (EPI 1) left, per index, cg = 1/2 u act(g) / (g + eps_g) and cu = 1/2 act(g) u / (u + eps_lin)
        // -- the identity rule on the activation, the uniform rule on the product and both Linears' stabilisers (ref lxt/efficient/patches.py:145-157,
        // lxt/efficient/rules.py:88-100, lxt/explicit/models/llama.py:84-86,273-281) -- so the rule here is Agu = Gm (*) (cg | cu): one multiply and
        // one pack per element, no v_exp / v_rcp (the round-5 epilogue recomputed act(g) from the stashed g: ~310 VALU instructions per 16 pairs,
        // 20 of the 26 us a tile's epilogue took).  Agu keeps the GEMM-operand layout [32 gate | 32 up] per 64 columns (v_permlane16_swap pairing).
        // Software-pipelined over the 8 row blocks (a, i), DEEPER as the accumulators drain (as EPI 3 / 4 below): block 0 was requested ahead
        // of the staging pieces, block 1 goes out first thing here; behind block 0's stores go the loads of blocks 2 AND 3 (into the registers
        // block 0's accumulators and coefficients just released), behind block 1's those of 4 and 5, then one per block: up to four blocks =
        // 16 x 16 B per lane in flight.  (With the rule reduced to a multiply the epilogue is bound by the coefficient stream itself: the 256
        // CUs reach it together and pull 0.47 GB per round.)
        if (full) {
            bf16_t* adst = reinterpret_cast<bf16_t*>(C) + (int64_t)mrow * ldc + 2 * (int64_t)ncol + epoff;
            auto& pre = gpre;
            auto swz = [&](const f32x4& t0, const f32x4& t1) {
                bf16x2 x0 = {(bf16_t)t0[0], (bf16_t)t0[1]}, x1 = {(bf16_t)t0[2], (bf16_t)t0[3]};
                bf16x2 y0 = {(bf16_t)t1[0], (bf16_t)t1[1]}, y1 = {(bf16_t)t1[2], (bf16_t)t1[3]};
                auto s0 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(uint32_t, x0), __builtin_bit_cast(uint32_t, y0), false, false);
                auto s1 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(uint32_t, x1), __builtin_bit_cast(uint32_t, y1), false, false);
                return u32x4{s0[0], s1[0], s0[1], s1[1]};
            };
            issue_gu(1, pre[1]);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                bf16_t* dst = adst + (int64_t)((b >> 2) * 64 + (b & 3) * 16) * ldc;
                f32x4 ag[4], au[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const u32x4 c = pre[b & 3][j];
                    const f32x4 gm4 = acc[b >> 2][b & 3][j];
                    ag[j] = f32x4{gm4[0] * bf16_lo(c[0]), gm4[1] * bf16_hi(c[0]), gm4[2] * bf16_lo(c[1]), gm4[3] * bf16_hi(c[1])};
                    au[j] = f32x4{gm4[0] * bf16_lo(c[2]), gm4[1] * bf16_hi(c[2]), gm4[2] * bf16_lo(c[3]), gm4[3] * bf16_hi(c[3])};
                }
                u32x4 out[4] = {swz(ag[0], ag[1]), swz(au[0], au[1]), swz(ag[2], ag[3]), swz(au[2], au[3])};
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 4; ++k) *reinterpret_cast<u32x4*>(dst + 32 * k) = out[k];
                if (b == 0) { issue_gu(2, pre[2]); issue_gu(3, pre[3]); }
                else if (b == 1) { issue_gu(4, pre[0]); issue_gu(5, pre[1]); }
                else if (b == 2) issue_gu(6, pre[2]);
                else if (b == 3) issue_gu(7, pre[3]);
                __builtin_amdgcn_sched_barrier(0);
            }
            done = true;
        }
    }
    if constexpr (EPI == 3 || EPI == 4) {
        // ---- residual add (+ row scale: EPI 4; + sum of squares of the rounded row: EPI 3), full tiles, pipelined like the block above but
        // DEEPER as the accumulators drain: blocks 0, 1 were requested ahead of the K loop / the staging pieces; behind block 0's stores go the
        // loads of blocks 2 AND 3 (into the registers block 0's accumulators and residual just released), behind block 1's those of 4 and 5,
        // then one per block: four blocks = 8 x 16 B per lane in flight where two (the first version) left the epilogue latency-bound
        // (4 KiB per wave in flight: +15 ... +37 us per launch for 67 MB).  The residual arrives in the STORE layout (8 consecutive
        // columns per lane after the v_permlane16_swap pairing of column tiles 2 jj, 2 jj + 1) and is un-paired by the same swap.
        if (full) {
            bf16_t* cdst = reinterpret_cast<bf16_t*>(C) + (int64_t)mrow * ldc + ncol + epoff;
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                float sq = 0.f;
                u32x4 out[2];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const u32x4 o = rpre[b & 3][jj];
                    auto s0 = __builtin_amdgcn_permlane16_swap(o[0], o[2], false, false);
                    auto s1 = __builtin_amdgcn_permlane16_swap(o[1], o[3], false, false);
                    const bf16x2 a0 = __builtin_bit_cast(bf16x2, (uint32_t)s0[0]), a1 = __builtin_bit_cast(bf16x2, (uint32_t)s1[0]);
                    const bf16x2 b0 = __builtin_bit_cast(bf16x2, (uint32_t)s0[1]), b1 = __builtin_bit_cast(bf16x2, (uint32_t)s1[1]);
                    const f32x4 r0 = {(float)a0[0], (float)a0[1], (float)a1[0], (float)a1[1]};
                    const f32x4 r1 = {(float)b0[0], (float)b0[1], (float)b1[0], (float)b1[1]};
                    f32x4 v0 = acc[b >> 2][b & 3][2 * jj], v1 = acc[b >> 2][b & 3][2 * jj + 1];
                    if constexpr (RS) { v0 *= rsv[b]; v1 *= rsv[b]; }
                    if constexpr (EPI == 3) {
                        // lxt.explicit placement: the Linear's own output z = x W^T is kept as well (its stabiliser z / (z + eps) and the add2
                        // rule's h / (h + eps) both need their operand; ref lxt/explicit/functional.py:355-364,430-459) -- one more 16-byte store per half block
                        if (ep.c2 != nullptr) {
                            bf16x2 p0 = {(bf16_t)v0[0], (bf16_t)v0[1]}, p1 = {(bf16_t)v0[2], (bf16_t)v0[3]};
                            bf16x2 q0 = {(bf16_t)v1[0], (bf16_t)v1[1]}, q1 = {(bf16_t)v1[2], (bf16_t)v1[3]};
                            auto w0 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(uint32_t, p0), __builtin_bit_cast(uint32_t, q0), false, false);
                            auto w1 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(uint32_t, p1), __builtin_bit_cast(uint32_t, q1), false, false);
                            *reinterpret_cast<u32x4*>(ep.c2 + (int64_t)(mrow + (b >> 2) * 64 + (b & 3) * 16) * ep.ldc2 + ncol + epoff + 32 * jj) =
                                u32x4{w0[0], w1[0], w0[1], w1[1]};
                        }
                    }
                    v0 += r0; v1 += r1;
                    bf16x2 x0 = {(bf16_t)v0[0], (bf16_t)v0[1]}, x1 = {(bf16_t)v0[2], (bf16_t)v0[3]};
                    bf16x2 y0 = {(bf16_t)v1[0], (bf16_t)v1[1]}, y1 = {(bf16_t)v1[2], (bf16_t)v1[3]};
                    if constexpr (EPI == 3) {
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const float q0 = (float)x0[e], q1 = (float)x1[e], q2 = (float)y0[e], q3 = (float)y1[e];
                            sq += q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3;
                        }
                    }
                    auto t0 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(uint32_t, x0), __builtin_bit_cast(uint32_t, y0), false, false);
                    auto t1 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(uint32_t, x1), __builtin_bit_cast(uint32_t, y1), false, false);
                    out[jj] = u32x4{t0[0], t1[0], t0[1], t1[1]};
                }
                __builtin_amdgcn_sched_barrier(0);
                bf16_t* dst = cdst + (int64_t)((b >> 2) * 64 + (b & 3) * 16) * ldc;
                *reinterpret_cast<u32x4*>(dst) = out[0];
                *reinterpret_cast<u32x4*>(dst + 32) = out[1];
                if constexpr (EPI == 3) {
                    // this wave's 64 columns of row gm: the four 16-lane rows hold 16 columns each
                    sq += __shfl_xor(sq, 16);
                    sq += __shfl_xor(sq, 32);
                    if (hi == 0) ep.ssq[(int64_t)(ncol >> 6) * ep.ldssq + mrow + (b >> 2) * 64 + (b & 3) * 16] = sq;
                }
                if (b == 0) { issue_res(2, rpre[2]); issue_res(3, rpre[3]); }
                else if (b == 1) { issue_res(4, rpre[0]); issue_res(5, rpre[1]); }
                else if (b == 2) issue_res(6, rpre[2]);
                else if (b == 3) issue_res(7, rpre[3]);
                __builtin_amdgcn_sched_barrier(0);
            }
            done = true;
        }
    }
    if constexpr (EPI == 5) {
        // ---- row scale + RoPE + bf16 pack, full tiles (the host admits only M, N multiples of 256).  Tables software-pipelined two row blocks
        // deep: blocks 0 and 1 were requested right after the K loop, block b + 2 goes out behind block b's stores (8 x 16 B per lane in flight;
        // the tables are L2-resident: 1 MB).
        if (full) {
            bf16_t* cdst = reinterpret_cast<bf16_t*>(C) + (int64_t)mrow * ldc + en0 + hb5 + 32 * w5 + epoff;
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                f32x4 v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[b >> 2][b & 3][j] * rsv[b];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    f32x4 cs = tpre[b & 1][jj], sn = tpre[b & 1][2 + jj];
                    if (!rot5) { cs = f32x4{1.f, 1.f, 1.f, 1.f}; sn = f32x4{0.f, 0.f, 0.f, 0.f}; }      // v tile: the identity rotation
                    const f32x4 x1 = v[jj], x2 = v[jj + 2];
                    v[jj] = x1 * cs - x2 * sn;
                    v[jj + 2] = x2 * cs + x1 * sn;
                }
                u32x4 out[2];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    bf16x2 x0 = {(bf16_t)v[2 * jj][0], (bf16_t)v[2 * jj][1]}, x1 = {(bf16_t)v[2 * jj][2], (bf16_t)v[2 * jj][3]};
                    bf16x2 y0 = {(bf16_t)v[2 * jj + 1][0], (bf16_t)v[2 * jj + 1][1]}, y1 = {(bf16_t)v[2 * jj + 1][2], (bf16_t)v[2 * jj + 1][3]};
                    auto s0 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(uint32_t, x0), __builtin_bit_cast(uint32_t, y0), false, false);
                    auto s1 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(uint32_t, x1), __builtin_bit_cast(uint32_t, y1), false, false);
                    out[jj] = u32x4{s0[0], s1[0], s0[1], s1[1]};
                }
                __builtin_amdgcn_sched_barrier(0);
                bf16_t* dst = cdst + (int64_t)((b >> 2) * 64 + (b & 3) * 16) * ldc;
                *reinterpret_cast<u32x4*>(dst) = out[0];                 // head-dim columns 32 w .. + 31 of the wave's head
                *reinterpret_cast<u32x4*>(dst + 64) = out[1];            // their rotate-half partners 64 + 32 w .. + 31
                if (b + 2 < 8) issue_tab(b + 2, tpre[b & 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
            done = true;
        }
    }
    if (!done)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gm = mrow + a * 64 + i * 16;
            f32x4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = acc[a][i][j];
                if constexpr (RS) v[j] *= rsv[a * 4 + i];
                if constexpr (EPI == 3 || EPI == 4) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int gn = ncol + j * 16 + 4 * hi + e;
                        if (gm < M && gn < N) {
                            if constexpr (EPI == 3) { if (ep.c2 != nullptr) ep.c2[(int64_t)gm * ep.ldc2 + gn] = (bf16_t)v[j][e]; }
                            v[j][e] += to_f32(ep.res[(int64_t)gm * ep.ldres + gn]);
                        }
                    }
                }
                if (bias) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int gn = ncol + j * 16 + 4 * hi + e;
                        if (gn < N) v[j][e] += to_f32(bias[gn]);
                    }
                }
            }
            if constexpr (EPI == 1) {
                // ---- gated-MLP forward rule in the gate/up GEMM's epilogue (round 6: the backward's COEFFICIENTS are stashed, not g / u).
                // The fused weight's rows are interleaved in blocks of 64 = [32 gate | 32 up], so this wave's column tiles j = 0, 1 are gate and
                // j = 2, 3 up of the SAME 32 intermediate indices, fp32, still in registers: m = act(g) (*) u, and the two factors the
                // down-projection dgrad's epilogue (EPI 2) multiplies Gm by -- cg = 1/2 u act(g) / (g + eps_g), cu = 1/2 act(g) u / (u + eps_lin)
                // (common.hpp: gated_coef).  act(g) is evaluated ONCE per explanation; g and u themselves are never written.  Stash layout
                // [M, 2 I] in ACCUMULATOR order: the 16 bytes {cg x 4 | cu x 4} of intermediate indices 4 t .. 4 t + 3 at columns 8 t .. 8 t + 7.
                f32x4 mv[2], cgv[2], cuv[2];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float m_, cg_, cu_;
                        gated_coef<LEAN, ACT>(v[jj][e], v[jj + 2][e], ep.eps_g, ep.eps_lin, m_, cg_, cu_);
                        mv[jj][e] = m_; cgv[jj][e] = cg_; cuv[jj][e] = cu_;
                    }
                const int mcol = ncol / 2;                               // first intermediate index of this wave's block
                if (full) {
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        bf16x2 c0 = {(bf16_t)cgv[jj][0], (bf16_t)cgv[jj][1]}, c1 = {(bf16_t)cgv[jj][2], (bf16_t)cgv[jj][3]};
                        bf16x2 c2_ = {(bf16_t)cuv[jj][0], (bf16_t)cuv[jj][1]}, c3 = {(bf16_t)cuv[jj][2], (bf16_t)cuv[jj][3]};
                        u32x4 o = {__builtin_bit_cast(uint32_t, c0), __builtin_bit_cast(uint32_t, c1), __builtin_bit_cast(uint32_t, c2_),
                                   __builtin_bit_cast(uint32_t, c3)};
                        *reinterpret_cast<u32x4*>(C + (int64_t)gm * ldc + ncol + 32 * jj + 8 * hi) = o;
                    }
                    bf16x2 x0 = {(bf16_t)mv[0][0], (bf16_t)mv[0][1]}, x1 = {(bf16_t)mv[0][2], (bf16_t)mv[0][3]};
                    bf16x2 y0 = {(bf16_t)mv[1][0], (bf16_t)mv[1][1]}, y1 = {(bf16_t)mv[1][2], (bf16_t)mv[1][3]};
                    auto s0 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(uint32_t, x0), __builtin_bit_cast(uint32_t, y0), false, false);
                    auto s1 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(uint32_t, x1), __builtin_bit_cast(uint32_t, y1), false, false);
                    u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
                    *reinterpret_cast<u32x4*>(ep.c2 + (int64_t)gm * ep.ldc2 + mcol + 16 * (hi & 1) + 8 * (hi >> 1)) = o;
                } else if (gm < M) {
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int ci = mcol + 16 * jj + 4 * hi + e;
                            if (2 * ci < N) {
                                ep.c2[(int64_t)gm * ep.ldc2 + ci] = (bf16_t)mv[jj][e];
                                C[(int64_t)gm * ldc + ncol + 32 * jj + 8 * hi + e] = (TO)cgv[jj][e];
                                C[(int64_t)gm * ldc + ncol + 32 * jj + 8 * hi + 4 + e] = (TO)cuv[jj][e];
                            }
                        }
                }
            }
            if constexpr (EPI == 3) {
                // ragged tiles: the wave's 64-column partial of row gm (columns past N contribute nothing; rows past M are not written)
                float sq = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float q = (float)(bf16_t)v[j][e];
                        if (ncol + j * 16 + 4 * hi + e < N) sq += q * q;
                    }
                sq += __shfl_xor(sq, 16);
                sq += __shfl_xor(sq, 32);
                if (hi == 0 && gm < M && ncol < N) ep.ssq[(int64_t)(ncol >> 6) * ep.ldssq + gm] = sq;
            }
            if constexpr (EPI == 1) {
                // (coefficients and m written above)
            } else if constexpr (EPI == 2) {
                // ragged tiles of the gated backward rule (the full tiles took the pipelined path above): element-wise, bounds-checked
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int64_t gcol = 2 * (int64_t)ncol + 64 * (j >> 1);            // this column tile's [32 gate | 32 up] block of Agu
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int ci = 16 * (j & 1) + 4 * hi + e;
                        if (gm < M && ncol + 16 * j + 4 * hi + e < N) {
                            const bf16_t* cp = ep.gu + (int64_t)gm * ep.ldgu + 2 * (int64_t)ncol + 32 * j + 8 * hi + e;
                            C[(int64_t)gm * ldc + gcol + ci] = (TO)(v[j][e] * (float)cp[0]);
                            C[(int64_t)gm * ldc + gcol + 32 + ci] = (TO)(v[j][e] * (float)cp[4]);
                        }
                    }
                }
            } else
            if constexpr (sizeof(TO) == 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int gn = ncol + j * 16 + 4 * hi;
                    TO* dst = C + (int64_t)gm * ldc + gn;
                    if (gm < M) {
                        if (vec4 && gn + 3 < N) {
                            *reinterpret_cast<f32x4*>(dst) = v[j];
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (gn + e < N) dst[e] = v[j][e];
                        }
                    }
                }
            } else if (full) {
                // pack to bf16 and pair the column tiles 2 jj, 2 jj + 1 with v_permlane16_swap (odd 16-lane rows of the first operand <->
                // even rows of the second): lane row q then holds 8 consecutive columns, 16 (2 jj + (q & 1)) + 8 (q >> 1) .. + 7
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    bf16x2 x0 = {(bf16_t)v[2 * jj][0], (bf16_t)v[2 * jj][1]}, x1 = {(bf16_t)v[2 * jj][2], (bf16_t)v[2 * jj][3]};
                    bf16x2 y0 = {(bf16_t)v[2 * jj + 1][0], (bf16_t)v[2 * jj + 1][1]}, y1 = {(bf16_t)v[2 * jj + 1][2], (bf16_t)v[2 * jj + 1][3]};
                    auto s0 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(uint32_t, x0), __builtin_bit_cast(uint32_t, y0), false, false);
                    auto s1 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(uint32_t, x1), __builtin_bit_cast(uint32_t, y1), false, false);
                    u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
                    *reinterpret_cast<u32x4*>(C + (int64_t)gm * ldc + ncol + 32 * jj + 16 * (hi & 1) + 8 * (hi >> 1)) = o;
                }
            } else if (gm < M) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int gn = ncol + j * 16 + 4 * hi + e;
                        if (gn < N) C[(int64_t)gm * ldc + gn] = (bf16_t)v[j][e];
                    }
            }
        }
    }
    // (a last "use" of the row scales at the very end of the tile: it keeps their registers out of the K loop's fragment set until the compiler's
    // wait model has seen them consumed on every path -- without it the EPI 4 instantiation carried a vmcnt wait at the top of the K loop, see
    // PP_VMWAIT)
    if constexpr (RS) {
#pragma unroll
        for (int b = 0; b < 8; ++b) asm volatile("" :: "v"(rsv[b]));
    }
    if constexpr (EPI == 5) {
#pragma unroll
        for (int k = 0; k < 4; ++k) asm volatile("" :: "v"(tpre[0][k]), "v"(tpre[1][k]));
    }
    if (g == 0) __builtin_amdgcn_s_barrier();                          // balances group 1's extra barrier
    if (!has_next) break;
    // the new tile's V0(0), V1(0) have landed = everything but the newest 8 staging pieces AND the stores issued behind them (loads and stores
    // retire in issue order through vmcnt on gfx9; the counter has 6 bits).  Full bf16 tiles: 16 stores per wave (+ 8 of m in the gated forward);
    // the fused gated backward's own gu loads were issued after the pieces and waited for by the compiler, only its last stores remain;
    // ragged tiles and fp32 output (scalar / conditional stores): drain.
    if (full && sizeof(TO) == 2) {
        if constexpr (EPI == 0 || EPI == 5) PP_VMWAIT(24);
        else if constexpr (EPI == 1) PP_VMWAIT(32);
        else PP_VMWAIT(8);
    } else PP_VMWAIT(0);
    }
    // (every LDS-DMA load was waited for inside the last K tile; the C stores may still be in flight when the wave ends)
#undef PP_KTILE
#undef PP_READ_A
#undef PP_READ_B
#undef PP_WAIT_A
#undef PP_WAIT_B
#undef PP_MMA
#undef PP_FENCE
}

template <typename TO, bool NN, int EPI, int ACT = 0, bool SK = false, bool LEAN = false, bool RS = false>
int launch_pp_t(const void* A, const void* B, void* C, const void* bias, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc,
                int splits, int kt_per_split, int64_t slab_stride, PPEpi ep, hipStream_t st) {
    const int tiles_m = (M + 255) / 256, tiles_n = (N + 255) / 256;
    int gx = tiles_m * tiles_n;
#if PP_PERSIST
    if (splits == 1) {
        const int ncu = lrp_num_cus();
        if (gx > ncu) gx = ncu;
    }
#endif
    dim3 grid(gx, splits), block(512);
    const size_t lds = 4 * (size_t)PP_OPND;
    auto kern = gemm_pp_kernel<TO, NN, EPI, ACT, SK, LEAN, RS>;
    LRP_SET_MAX_LDS(kern, lds);
    hipLaunchKernelGGL(kern, grid, block, lds, st, (const bf16_t*)A, (const bf16_t*)B, (TO*)C, (const bf16_t*)bias, M, N, K, lda,
                       ldb, ldc, tiles_m, tiles_n, kt_per_split, slab_stride, ep);
    return lrp_check_launch();
}

}  // namespace

// Host entry used by the dispatchers of gemm.hip.  bf16 operands; K a multiple of 64, >= 128 per split; every operand below 2^30 elements
// (32-bit buffer offsets).  nn = 0: B is [N, K] (ldb = row pitch of B); nn = 1: B is [K, N].  splits > 1: C must be fp32 slabs
// [splits][M][ldc] (slab_stride elements apart), each holding the partial sum of kt_per_split K tiles; no bias then.
int lrp_launch_gemm_pp(const void* A, const void* B, void* C, const void* bias, int M, int N, int K, int64_t lda, int64_t ldb,
                       int64_t ldc, int out_dtype, int nn, int splits, int kt_per_split, int64_t slab_stride, hipStream_t st) {
    const PPEpi ep{};
    if (M <= 240) {          // one row of tiles with at least one dead 16-row block: the skinny instantiations
        if (out_dtype == LRP_F32) {
            if (nn) return launch_pp_t<float, true, 0, 0, true>(A, B, C, bias, M, N, K, lda, ldb, ldc, splits, kt_per_split, slab_stride, ep, st);
            return launch_pp_t<float, false, 0, 0, true>(A, B, C, bias, M, N, K, lda, ldb, ldc, splits, kt_per_split, slab_stride, ep, st);
        }
        if (nn) return launch_pp_t<bf16_t, true, 0, 0, true>(A, B, C, bias, M, N, K, lda, ldb, ldc, splits, kt_per_split, slab_stride, ep, st);
        return launch_pp_t<bf16_t, false, 0, 0, true>(A, B, C, bias, M, N, K, lda, ldb, ldc, splits, kt_per_split, slab_stride, ep, st);
    }
    if (out_dtype == LRP_F32) {
        if (nn) return launch_pp_t<float, true, 0>(A, B, C, bias, M, N, K, lda, ldb, ldc, splits, kt_per_split, slab_stride, ep, st);
        return launch_pp_t<float, false, 0>(A, B, C, bias, M, N, K, lda, ldb, ldc, splits, kt_per_split, slab_stride, ep, st);
    }
    if (nn) return launch_pp_t<bf16_t, true, 0>(A, B, C, bias, M, N, K, lda, ldb, ldc, splits, kt_per_split, slab_stride, ep, st);
    return launch_pp_t<bf16_t, false, 0>(A, B, C, bias, M, N, K, lda, ldb, ldc, splits, kt_per_split, slab_stride, ep, st);
}

// gate/up forward with the gated rule in the epilogue: m[M, I] = act(g) (*) u and the backward's coefficient stash coef[M, 2 I] (accumulator order,
// include/lrp_hip.h) from x[M, K] . Wgu[2 I, K]^T (rows interleaved [32 gate | 32 up]); rs != NULL: the accumulators are scaled by rs[row] first
// (K1n: the folded RMSNorm's 1 / rms).  The activation is a compile-time parameter: with a run-time switch inside the 8-fold unrolled epilogue
// hipcc gives up unrolling and moves the 128 accumulators to scratch.  LEAN = the lxt.efficient placement (eps_g >= 1e-30, no Linear stabiliser).
namespace {
template <int ACT, bool LEAN>
int gated_fwd_t(const void* x, const void* Wgu, const float* rs, void* coef, int M, int I, int K, int64_t ldx, int64_t ldw, int64_t ldcoef,
                const PPEpi& ep, hipStream_t st) {
    if (rs) return launch_pp_t<bf16_t, false, 1, ACT, false, LEAN, true>(x, Wgu, coef, nullptr, M, 2 * I, K, ldx, ldw, ldcoef, 1, K / PP_KT, 0, ep, st);
    return launch_pp_t<bf16_t, false, 1, ACT, false, LEAN, false>(x, Wgu, coef, nullptr, M, 2 * I, K, ldx, ldw, ldcoef, 1, K / PP_KT, 0, ep, st);
}
}  // namespace
int lrp_launch_gemm_pp_gated_fwd(const void* x, const void* Wgu, const float* rs, void* coef, void* m, int M, int I, int K, int64_t ldx,
                                 int64_t ldw, int64_t ldcoef, int64_t ldm, float eps_g, float eps_lin, int act, hipStream_t st) {
    PPEpi ep{};
    ep.c2 = (bf16_t*)m; ep.ldc2 = ldm; ep.act = act; ep.rs = rs; ep.eps_g = eps_g; ep.eps_lin = eps_lin;
    const bool lean = eps_g >= 1e-30f && eps_lin == 0.f;
    if (act == LRP_ACT_SILU)
        return lean ? gated_fwd_t<LRP_ACT_SILU, true>(x, Wgu, rs, coef, M, I, K, ldx, ldw, ldcoef, ep, st)
                    : gated_fwd_t<LRP_ACT_SILU, false>(x, Wgu, rs, coef, M, I, K, ldx, ldw, ldcoef, ep, st);
    if (act == LRP_ACT_GELU_TANH)
        return lean ? gated_fwd_t<LRP_ACT_GELU_TANH, true>(x, Wgu, rs, coef, M, I, K, ldx, ldw, ldcoef, ep, st)
                    : gated_fwd_t<LRP_ACT_GELU_TANH, false>(x, Wgu, rs, coef, M, I, K, ldx, ldw, ldcoef, ep, st);
    return LRP_ESHAPE;
}

// down-projection dgrad with the gated rule in the epilogue: Gm = Adn[M, K] . Wdn[K, I] (NN, never stored) -> Agu[M, 2 I] = Gm (*) coef
int lrp_launch_gemm_pp_gated_bwd(const void* Adn, const void* Wdn, const void* coef, void* Agu, int M, int I, int K, int64_t lda,
                                 int64_t ldw, int64_t ldcoef, int64_t ldagu, hipStream_t st) {
    PPEpi ep{};
    ep.gu = (const bf16_t*)coef; ep.ldgu = ldcoef;
    return launch_pp_t<bf16_t, true, 2>(Adn, Wdn, Agu, nullptr, M, I, K, lda, ldw, ldagu, 1, K / PP_KT, 0, ep, st);
}

// ---- RMSNorm folded into the GEMMs around it (round 5; include/lrp_hip.h "K1n").  Same kernel, three more epilogue forms.
// out = res + x W^T (NT) and the partial sums of squares of out's rows, one per 64-column block: ssq [N / 64][ldssq]
int lrp_launch_gemm_pp_res_ssq(const void* x, const void* W, const void* res, void* out, float* ssq, int M, int N, int K, int64_t ldx,
                               int64_t ldw, int64_t ldres, int64_t ldout, int64_t ldssq, void* raw, int64_t ldraw, hipStream_t st) {
    PPEpi ep{};
    ep.res = (const bf16_t*)res; ep.ldres = ldres; ep.ssq = ssq; ep.ldssq = ldssq; ep.c2 = (bf16_t*)raw; ep.ldc2 = ldraw;
    return launch_pp_t<bf16_t, false, 3>(x, W, out, nullptr, M, N, K, ldx, ldw, ldout, 1, K / PP_KT, 0, ep, st);
}
// out = rs (.) (x W^T) (NT), rs [M] fp32
int lrp_launch_gemm_pp_nt_rs(const void* x, const void* W, const float* rs, void* out, int M, int N, int K, int64_t ldx, int64_t ldw,
                             int64_t ldout, hipStream_t st) {
    PPEpi ep{};
    ep.rs = rs;
    return launch_pp_t<bf16_t, false, 0, 0, false, false, true>(x, W, out, nullptr, M, N, K, ldx, ldw, ldout, 1, K / PP_KT, 0, ep, st);
}
// out = RoPE(rs (.) (x W^T)) on the q / k head columns [0, rope_cols), rs (.) (x W^T) on the rest (NT; heads of 128; M, N multiples of 256)
int lrp_launch_gemm_pp_nt_rs_rope(const void* x, const void* W, const float* rs, const float* cos, const float* sin, void* out, int M, int N, int K,
                                  int64_t ldx, int64_t ldw, int64_t ldout, int seq, int rope_cols, hipStream_t st) {
    PPEpi ep{};
    ep.rs = rs; ep.cos = cos; ep.sin = sin; ep.seq = seq; ep.rope_cols = rope_cols;
    return launch_pp_t<bf16_t, false, 5, 0, false, false, true>(x, W, out, nullptr, M, N, K, ldx, ldw, ldout, 1, K / PP_KT, 0, ep, st);
}
// out = rs (.) (s W) (NN: W [K, N] as stored)
int lrp_launch_gemm_pp_nn_rs(const void* s, const void* W, const float* rs, void* out, int M, int N, int K, int64_t lds_, int64_t ldw,
                             int64_t ldout, hipStream_t st) {
    PPEpi ep{};
    ep.rs = rs;
    return launch_pp_t<bf16_t, true, 0, 0, false, false, true>(s, W, out, nullptr, M, N, K, lds_, ldw, ldout, 1, K / PP_KT, 0, ep, st);
}
// out = rs (.) (s W) + res (NN: W [K, N] as stored)
int lrp_launch_gemm_pp_nn_rs_res(const void* s, const void* W, const float* rs, const void* res, void* out, int M, int N, int K, int64_t lds_,
                                 int64_t ldw, int64_t ldres, int64_t ldout, hipStream_t st) {
    PPEpi ep{};
    ep.rs = rs; ep.res = (const bf16_t*)res; ep.ldres = ldres;
    return launch_pp_t<bf16_t, true, 4, 0, false, false, true>(s, W, out, nullptr, M, N, K, lds_, ldw, ldout, 1, K / PP_KT, 0, ep, st);
}
