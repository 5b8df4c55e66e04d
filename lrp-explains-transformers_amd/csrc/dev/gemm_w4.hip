// dev/gemm_w4.hip -- EXPERIMENT (not linked into liblrp_hip.so; `make dev`): the big bf16 NT GEMMs in the vendor kernel's structure.
// Measured (profiles/r02_gemm_experiments.txt): correct on every shape tried, 1.27-1.29 PF on N=28672/K=4096 (16-wave product kernel
// 1.26-1.27), 1.32-1.34 on K=14336 (product 1.37-1.38): the MFMA stream itself runs at the single-wave issue limit (18.0-18.4 cycles
// per MFMA), what is left are ~13-50 cycles per LDS-DMA issue and ~270 cycles of waits / barriers per K tile that the vendor kernel
// does not pay (rocprofv3: identical L2 / fabric traffic, 63 % vs 88 % MFMA-slot occupancy, 1.96 vs 1.68 GHz).
//
//  C[M,N] = A[M,K] . B[N,K]^T (+bias), fp32 accumulate.
//
// 256 x 256 tile, FOUR waves = ONE wave per SIMD, 128 x 128 per wave on v_mfma_f32_16x16x32_bf16:
//   * the 64 accumulator tiles of a wave live in the 256 AGPRs, which leaves the 256 arch VGPRs for a FULL double buffer of
//     fragments (8 A + 8 B fragments per 32-element K sub-step, two sub-steps): the fragments of sub-step s+1 are read while
//     the 64 MFMAs of sub-step s issue, so no MFMA ever waits for the LDS, and one 16-byte LDS fragment feeds 8 MFMAs
//     (16-wave form of gemm.hip: 4) -- half the LDS traffic per FLOP;
//   * the K loop contains NO VALU instruction besides the MFMAs (they would compete for the one VALU issue port: with a single
//     wave per SIMD nothing hides them -- r01 measured +3.5 cycles per interleaved VALU): every LDS address is a per-lane base
//     register + an immediate, every global address a per-lane offset register + a buffer resource whose base the scalar
//     unit advances; fragment reads, direct-to-LDS loads, waits and barriers are placed BY HAND between the MFMAs, at most
//     one per gap (inline asm, all volatile: the order below is the order in the binary);
//   * LDS image per operand and 64-element K tile: 32 blocks of [8 rows][128 B] + 32 B pad (1056 B).  Block (h, q) holds the
//     EIGHT CONSECUTIVE rows h*128 + 8 q + i, i = 0..7, of the operand's 256 tile rows: ONE direct-to-LDS wave instruction
//     (lane l -> row 8 q + (l>>3), bytes 16 (l&7)) fills it with eight full 128-byte lines.  MFMA fragment i of a wave is the
//     row set {8 q + i, q = 0..15} (any bijection works: the accumulator layout follows from it), so fragment (i, sub-step s)
//     of lane l (row slot q = l&15, k-chunk l>>4) sits at  block(q) + i*128 + s*64 + (l>>4)*16 : the sixteen row slots of a
//     fragment are 1056 B apart = 8 banks (mod 64); with the hardware's ds_read_b128 lane groups (which mix k-chunks) that is the
//     conflict-free pad, 1040 B is not (tests/test_layout_maps_cpu.py).  With the same
//     bijection on the N side a lane's 64 accumulator tiles hold, for each of its 8 rows, 4 runs of 8 CONSECUTIVE columns:
//     32 sixteen-byte stores per lane;
//   * two LDS stages (2 x 67,584 B), tiles prefetched TWO ahead: the loads of tile t+2 go into tile t's stage as soon as every
//     wave has read tile t's last fragments (barrier in the middle of sub-step 0); tile t+1 is waited for (counted vmcnt) and
//     published (barrier) in the middle of sub-step 1, ~1.5 iterations after its loads were issued;
//   * persistent: one workgroup per CU walks the output tiles; the first two K tiles of the NEXT output tile are issued in the
//     middle of the current tile's last K iteration, so the pipeline never drains between tiles and the C stores of a tile
//     overlap the first loads of the next.
// The structure (one wave per SIMD, AGPR accumulators, register-double-buffered fragments, padded fragment-friendly LDS
// blocks) is what the vendor's best kernel for these shapes does (profiles/r02_gemm_experiments.txt); the schedule here is
// written for this tile walk and this ABI.
#include "../common.hpp"

namespace {

constexpr int W4_KT = 64;                  // K elements per tile (128 B per row)
#ifndef W4_BLK_BYTES
#define W4_BLK_BYTES 1056
#endif
constexpr int W4_BLK = W4_BLK_BYTES;       // one LDS block: 8 rows x 128 B + pad (16 B as the vendor kernel; 32 B is the conflict-free pad
                                           // under the ds_read_b128 lane groups of MI355X_MICROARCH.md: tests/test_layout_maps_cpu.py)
constexpr int W4_OPND = 32 * W4_BLK;       // one operand of one stage
constexpr int W4_STAGE = 2 * W4_OPND;      // 66,560 B

typedef __attribute__((ext_vector_type(4))) int i32x4;

template <typename TO>
__global__ __launch_bounds__(256, 1) void gemm_nt_w4_kernel(
    const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, TO* __restrict__ C, const bf16_t* __restrict__ bias,
    int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NSTORE = sizeof(TO) == 4 ? 64 : 32;                   // C store instructions per lane and tile
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ntile = tiles_m * tiles_n;
    const int nkt = K / W4_KT;                                          // host guarantees nkt >= 2

#ifndef W4_SPLIT
#define W4_SPLIT 2                          // best measured form (profiles/r02_gemm_experiments.txt)
#endif
#if W4_SPLIT
    // ---- staging role (variant): EVERY wave stages both operands -- rows 64 w .. 64 w + 63 of A (pieces 0..7) and of B (pieces 8..15)
    uint32_t voff[16];
#pragma unroll
#if W4_SPLIT == 2
    // (W4_SPLIT == 2: the vendor kernel's walk -- at any time the four waves cover 32 CONSECUTIVE rows (8 each) and step 32 rows
    // per piece: piece p of wave w = rows 32 p + 8 w .. + 7 = LDS block 4 p + w)
    for (int q = 0; q < 16; ++q)
        voff[q] = (uint32_t)(((int64_t)(32 * (q & 7) + 8 * wave + (lane >> 3)) * (q < 8 ? lda : ldb)) * 2 + (lane & 7) * 16);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const uint32_t st_dst = lds0 + wave * W4_BLK;
#define W4_DSTOFF(q) (((q) < 8 ? 0 : W4_OPND) + ((q)&7) * 4 * W4_BLK)
#else
    for (int q = 0; q < 16; ++q)
        voff[q] = (uint32_t)(((int64_t)(64 * wave + 8 * (q & 7) + (lane >> 3)) * (q < 8 ? lda : ldb)) * 2 + (lane & 7) * 16);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const uint32_t st_dst = lds0 + wave * 8 * W4_BLK;                    // + stage * W4_STAGE + (q < 8 ? 0 : W4_OPND) + (q & 7) * W4_BLK
#define W4_DSTOFF(q) (((q) < 8 ? 0 : W4_OPND) + ((q)&7) * W4_BLK)
#endif
#define W4_RS(rsrc, q) ((q) < 8 ? rsrc.a : rsrc.b)
#else
    // ---- staging role: waves 0, 1 stage the two 128-row halves of A, waves 2, 3 those of B: 16 blocks per wave and K tile
    const bool stB = wave >= 2;
    const int half = wave & 1;
    const int64_t ld = stB ? ldb : lda;
    uint32_t voff[16];                                                   // per-lane byte offset of this lane's piece of block q
    // (measured: the order of rows ACROSS consecutive pieces matters as much as within one -- with the N-side blocks permuted so that
    // a store instruction writes 64 contiguous bytes per row, consecutive pieces jump 32 rows and the kernel drops from 1.24 to 0.97
    // PFLOP/s: the LDS-DMA issue cost follows the address-translation locality of the rows, 8 KB .. 28 KB apart)
#pragma unroll
    for (int q = 0; q < 16; ++q) voff[q] = (uint32_t)(((int64_t)(8 * q + (lane >> 3)) * ld) * 2 + (lane & 7) * 16);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const uint32_t st_dst = lds0 + (stB ? W4_OPND : 0) + half * 16 * W4_BLK;   // + stage * W4_STAGE + q * W4_BLK
#define W4_DSTOFF(q) ((q) * W4_BLK)
#define W4_RS(rsrc, q) rsrc.a
#endif
    struct Srd2 { i32x4 a, b; };

    // ---- fragment addresses of this wave in stage 0; the stage is switched by XOR with (addr0 ^ addr1)
    const uint32_t vA0 = lds0 + (wm * 16 + (lane & 15)) * W4_BLK + (lane >> 4) * 16;
    const uint32_t vB0 = lds0 + W4_OPND + (wn * 16 + (lane & 15)) * W4_BLK + (lane >> 4) * 16;
    const uint32_t xA = vA0 ^ (vA0 + W4_STAGE), xB = vB0 ^ (vB0 + W4_STAGE);
    uint32_t vA = vA0, vB = vB0;

    f32x4 acc[8][8];
    bf16x8 af[2][8], bf[2][8];                                            // [sub-step][fragment]

#define W4_MM(s, i, j) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(bf[s][j]), "v"(af[s][i]))
#define W4_RDA(s, i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(af[s][i]) : "v"(vA), "n"((i) * 128 + (s) * 64))
#define W4_RDB(s, j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bf[s][j]) : "v"(vB), "n"((j) * 128 + (s) * 64))
// one direct-to-LDS piece: block q of this wave's operand half, into the stage at byte address `stg` (an SGPR), through `rsrc`.
// In the K loop the M0 update and the load sit in two different MFMA gaps (no s_nop for the M0 hazard, one instruction per gap)
#define W4_M0(stg, q) asm volatile("s_add_u32 m0, %0, %1" : : "s"(stg), "n"(W4_DSTOFF(q)) : "scc")
#define W4_LD(rsrc, q) asm volatile("buffer_load_dwordx4 %0, %1, 0 offen lds" : : "v"(voff[q]), "s"(W4_RS(rsrc, q)) : "memory")
#define W4_DMA(rsrc, stg, q)                                                                                        \
    asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds"                       \
                 : : "s"(stg), "n"(W4_DSTOFF(q)), "v"(voff[q]), "s"(W4_RS(rsrc, q)) : "memory", "scc")   /* s_add_u32 writes SCC */
#define W4_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define W4_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define W4_BARRIER() asm volatile("s_barrier" ::: "memory")
#define W4_SWAP() asm volatile("v_xor_b32 %0, %0, %2\n\tv_xor_b32 %1, %1, %3" : "+v"(vA), "+v"(vB) : "v"(xA), "v"(xB))
#define W4_ROW(s, i, X0, X1, X2, X3, X4, X5, X6, X7)                                                                \
    W4_MM(s, i, 0); X0; W4_MM(s, i, 1); X1; W4_MM(s, i, 2); X2; W4_MM(s, i, 3); X3;                                  \
    W4_MM(s, i, 4); X4; W4_MM(s, i, 5); X5; W4_MM(s, i, 6); X6; W4_MM(s, i, 7); X7
#define W4_NOP (void)0
#define W4_ROW_PLAIN(s, i) W4_ROW(s, i, W4_NOP, W4_NOP, W4_NOP, W4_NOP, W4_NOP, W4_NOP, W4_NOP, W4_NOP)
#define W4_DMA16(rsrc, stg)                                                                                         \
    W4_DMA(rsrc, stg, 0); W4_DMA(rsrc, stg, 1); W4_DMA(rsrc, stg, 2); W4_DMA(rsrc, stg, 3); W4_DMA(rsrc, stg, 4);    \
    W4_DMA(rsrc, stg, 5); W4_DMA(rsrc, stg, 6); W4_DMA(rsrc, stg, 7); W4_DMA(rsrc, stg, 8); W4_DMA(rsrc, stg, 9);    \
    W4_DMA(rsrc, stg, 10); W4_DMA(rsrc, stg, 11); W4_DMA(rsrc, stg, 12); W4_DMA(rsrc, stg, 13); W4_DMA(rsrc, stg, 14); \
    W4_DMA(rsrc, stg, 15)
// dev builds only (-DW4_TIMELINE, tools/gemm_w4_timeline.py): shader-clock totals of the loop's wait points
#ifdef W4_TIMELINE
#define W4_TS(k) { uint64_t now_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)); tl[k] += (uint32_t)now_ - tl_prev; tl_prev = (uint32_t)now_; }
    uint32_t tl[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tl_prev = 0, tl_tile = 0, tl_kernel = 0;
    { uint64_t now_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)); tl_kernel = tl_tile = (uint32_t)now_; }
#else
#define W4_TS(k)
#endif

    // this wave's staging source for an output tile: raw buffer resource over its 128 operand rows (rows past the operand's end
    // are out of range and read as zeros: their products are never stored)
    auto tile_origin = [&](int it, int& m0, int& n0) {
        int tm, tn;
        grouped_tile(xcd_remap(it, ntile), tiles_m, tiles_n, tm, tn);
        m0 = tm * 256;
        n0 = tn * 256;
    };
    auto one_srd = [&](const bf16_t* p, int64_t ld_, int row0, int rows_total, int span, int kt, bool live) {
        int rows_left = live ? rows_total - row0 : 0;                    // !live: every access out of range (zeros, no traffic)
        rows_left = rows_left < 0 ? 0 : (rows_left > span ? span : rows_left);
        const uint64_t base = reinterpret_cast<uint64_t>(p + (int64_t)row0 * ld_) + (uint64_t)kt * 128;
        i32x4 r;                                                         // (readfirstlane: keeps the tuple in SGPRs for the asm)
        r[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)base);
        r[1] = __builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32));   // stride 0 (raw buffer)
        r[2] = __builtin_amdgcn_readfirstlane((int)(uint32_t)((int64_t)rows_left * ld_ * 2));
        r[3] = 0x00020000;
        return r;
    };
    auto make_srd = [&](int m0, int n0, int kt, bool live = true) {
        Srd2 r;
#if W4_SPLIT
        r.a = one_srd(A, lda, m0, M, 256, kt, live);
        r.b = one_srd(B, ldb, n0, N, 256, kt, live);
#else
        r.a = r.b = one_srd(stB ? B : A, ld, (stB ? n0 : m0) + half * 128, stB ? N : M, 128, kt, live);
#endif
        return r;
    };

    int it = blockIdx.x;
    if (it >= ntile) return;
#ifdef W4_SKEW
    // all workgroups walk equally long tiles in lock step, so their C store bursts (32 MiB at once) coincide and are bandwidth-
    // bound; a one-off start skew of (workgroup mod 8) / 8 of a tile time spreads them (only worth it with many tiles per CU)
    if (ntile >= 8 * (int)gridDim.x) {
        const int naps = ((blockIdx.x >> 3) & 7) * nkt * 2900 / 8 / (64 * 127);
        for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(127);
    }
#endif
    int m0, n0;
    tile_origin(it, m0, n0);
    {   // the very first output tile: K tiles 0 and 1 in flight
        const Srd2 s0 = make_srd(m0, n0, 0), s1 = make_srd(m0, n0, 1);
        const uint32_t d0 = st_dst, d1 = st_dst + W4_STAGE;
        W4_DMA16(s0, d0);
        W4_DMA16(s1, d1);
    }
    bool first = true;
    for (; it < ntile; it += gridDim.x) {
        // K tile 0 has landed: its 16 pieces are the oldest outstanding operations of this wave; younger ones: the 16 pieces of K
        // tile 1 and (after the first output tile) the C stores of the previous tile (vmcnt counts stores too)
        if (first) { W4_WAIT_VM(16); } else if (NSTORE == 32) { W4_WAIT_VM(48); } else { W4_WAIT_VM(63); }
        first = false;
        W4_BARRIER();
        vA = vA0; vB = vB0;
        W4_RDB(0, 0); W4_RDB(0, 1); W4_RDB(0, 2); W4_RDB(0, 3); W4_RDB(0, 4); W4_RDB(0, 5); W4_RDB(0, 6); W4_RDB(0, 7);
        W4_RDA(0, 0); W4_RDA(0, 1); W4_RDA(0, 2); W4_RDA(0, 3); W4_RDA(0, 4); W4_RDA(0, 5); W4_RDA(0, 6); W4_RDA(0, 7);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        W4_WAIT_LGKM0();
        const int nit = it + (int)gridDim.x;                             // the next output tile of this workgroup
        const bool has_next = nit < ntile;
        int nm0 = 0, nn0 = 0;
        if (has_next) tile_origin(nit, nm0, nn0);
#ifdef W4_TIMELINE
        { uint64_t now_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)); tl_prev = (uint32_t)now_; }
        const uint32_t tl_start = tl_prev;
        tl[8] += tl_start - tl_tile;                                     // 8: tile prologue (wait for K tile 0, first reads, zeroing)
#endif

        // ---- one K tile t.  MODE 0: tile t+2 exists (its 16 pieces go into tile t's stage); 1: only tile t+1 exists;
        //      2: last K tile -- the first two K tiles of the NEXT output tile are issued instead (if there is one)
        uint32_t cur = 0;                                                // byte offset of tile t's stage
#define W4_BODY(MODE)                                                                                                \
    {                                                                                                                \
        const Srd2 srd = (MODE == 0) ? make_srd(m0, n0, t + 2) : make_srd(nm0, nn0, 0, has_next);                    \
        const Srd2 srd1 = make_srd(nm0, nn0, 1, has_next);                                                           \
        const uint32_t dst = (MODE == 0) ? st_dst + cur : st_dst, dst1 = st_dst + W4_STAGE;                          \
        /* sub-step 0: 64 MFMAs; the 16 fragments of sub-step 1 are read under the first 32 */                       \
        W4_ROW(0, 0, W4_RDB(1, 0), W4_NOP, W4_RDB(1, 1), W4_NOP, W4_RDB(1, 2), W4_NOP, W4_RDB(1, 3), W4_NOP);         \
        W4_ROW(0, 1, W4_RDB(1, 4), W4_NOP, W4_RDB(1, 5), W4_NOP, W4_RDB(1, 6), W4_NOP, W4_RDB(1, 7), W4_NOP);         \
        W4_ROW(0, 2, W4_RDA(1, 0), W4_NOP, W4_RDA(1, 1), W4_NOP, W4_RDA(1, 2), W4_NOP, W4_RDA(1, 3), W4_NOP);         \
        W4_ROW(0, 3, W4_RDA(1, 4), W4_NOP, W4_RDA(1, 5), W4_NOP, W4_RDA(1, 6), W4_NOP, W4_RDA(1, 7), W4_NOP);         \
        W4_ROW_PLAIN(0, 4);                                                                                          \
        /* every fragment of tile t is in registers: its stage is free once all four waves are here */               \
        if (MODE == 0) {                                                                                             \
            W4_ROW(0, 5, W4_TS(0), W4_WAIT_LGKM0(); W4_TS(1), W4_BARRIER(); W4_TS(2), W4_M0(dst, 0), W4_LD(srd, 0), W4_NOP, W4_M0(dst, 1), W4_LD(srd, 1)); \
            W4_ROW(0, 6, W4_NOP, W4_M0(dst, 2), W4_LD(srd, 2), W4_NOP, W4_M0(dst, 3), W4_LD(srd, 3), W4_NOP, W4_M0(dst, 4)); \
            W4_ROW(0, 7, W4_LD(srd, 4), W4_NOP, W4_M0(dst, 5), W4_LD(srd, 5), W4_NOP, W4_M0(dst, 6), W4_LD(srd, 6), W4_NOP); \
            W4_ROW(1, 0, W4_M0(dst, 7), W4_LD(srd, 7), W4_NOP, W4_M0(dst, 8), W4_LD(srd, 8), W4_NOP, W4_M0(dst, 9), W4_LD(srd, 9)); \
            W4_ROW(1, 1, W4_NOP, W4_M0(dst, 10), W4_LD(srd, 10), W4_NOP, W4_M0(dst, 11), W4_LD(srd, 11), W4_NOP, W4_M0(dst, 12)); \
            W4_ROW(1, 2, W4_LD(srd, 12), W4_NOP, W4_M0(dst, 13), W4_LD(srd, 13), W4_NOP, W4_M0(dst, 14), W4_LD(srd, 14), W4_NOP); \
            W4_ROW(1, 3, W4_M0(dst, 15), W4_LD(srd, 15), W4_NOP, W4_NOP, W4_NOP, W4_NOP, W4_NOP, W4_NOP); \
        } else if (MODE == 2) {                                                                                      \
            /* both stages are free: K tiles 0 and 1 of the next output tile, one piece per two MFMAs (no next tile: the \
               resource is empty and the pieces write zeros -- no branch around the MFMA stream) */                  \
            W4_ROW(0, 5, W4_NOP, W4_WAIT_LGKM0(), W4_BARRIER(), W4_DMA(srd, dst, 0), W4_NOP, W4_DMA(srd, dst, 1), W4_NOP, W4_DMA(srd, dst, 2)); \
            W4_ROW(0, 6, W4_NOP, W4_DMA(srd, dst, 3), W4_NOP, W4_DMA(srd, dst, 4), W4_NOP, W4_DMA(srd, dst, 5), W4_NOP, W4_DMA(srd, dst, 6)); \
            W4_ROW(0, 7, W4_NOP, W4_DMA(srd, dst, 7), W4_NOP, W4_DMA(srd, dst, 8), W4_NOP, W4_DMA(srd, dst, 9), W4_NOP, W4_DMA(srd, dst, 10)); \
            W4_ROW(1, 0, W4_NOP, W4_DMA(srd, dst, 11), W4_NOP, W4_DMA(srd, dst, 12), W4_NOP, W4_DMA(srd, dst, 13), W4_NOP, W4_DMA(srd, dst, 14)); \
            W4_ROW(1, 1, W4_NOP, W4_DMA(srd, dst, 15), W4_NOP, W4_DMA(srd1, dst1, 0), W4_NOP, W4_DMA(srd1, dst1, 1), W4_NOP, W4_DMA(srd1, dst1, 2)); \
            W4_ROW(1, 2, W4_NOP, W4_DMA(srd1, dst1, 3), W4_NOP, W4_DMA(srd1, dst1, 4), W4_NOP, W4_DMA(srd1, dst1, 5), W4_NOP, W4_DMA(srd1, dst1, 6)); \
            W4_ROW(1, 3, W4_NOP, W4_DMA(srd1, dst1, 7), W4_NOP, W4_DMA(srd1, dst1, 8), W4_NOP, W4_DMA(srd1, dst1, 9), W4_NOP, W4_DMA(srd1, dst1, 10)); \
        } else {                                                                                                     \
            W4_ROW(0, 5, W4_NOP, W4_WAIT_LGKM0(), W4_BARRIER(), W4_NOP, W4_NOP, W4_NOP, W4_NOP, W4_NOP);             \
            W4_ROW_PLAIN(0, 6); W4_ROW_PLAIN(0, 7); W4_ROW_PLAIN(1, 0); W4_ROW_PLAIN(1, 1); W4_ROW_PLAIN(1, 2); W4_ROW_PLAIN(1, 3); \
        }                                                                                                            \
        if (MODE != 2) {                                                                                             \
            /* tile t+1: this wave's 16 pieces have landed (MODE 0: the 16 younger ones are tile t+2's), then everybody's; \
               its first fragments are read under the rest of sub-step 1 */                                          \
            if (MODE == 0) { W4_ROW(1, 4, W4_TS(3); W4_WAIT_VM(16); W4_TS(4), W4_BARRIER(); W4_TS(5); W4_SWAP(), W4_RDB(0, 0), W4_RDB(0, 1), W4_RDB(0, 2), W4_RDB(0, 3), W4_RDB(0, 4), W4_RDB(0, 5)); } \
            else { W4_ROW(1, 4, W4_WAIT_VM(0), W4_BARRIER(); W4_SWAP(), W4_RDB(0, 0), W4_RDB(0, 1), W4_RDB(0, 2), W4_RDB(0, 3), W4_RDB(0, 4), W4_RDB(0, 5)); } \
            W4_ROW(1, 5, W4_RDB(0, 6), W4_NOP, W4_RDB(0, 7), W4_NOP, W4_RDA(0, 0), W4_NOP, W4_RDA(0, 1), W4_NOP);     \
            W4_ROW(1, 6, W4_RDA(0, 2), W4_NOP, W4_RDA(0, 3), W4_NOP, W4_RDA(0, 4), W4_NOP, W4_RDA(0, 5), W4_NOP);     \
            W4_ROW(1, 7, W4_RDA(0, 6), W4_NOP, W4_RDA(0, 7), W4_NOP, W4_NOP, W4_NOP, W4_NOP, W4_WAIT_LGKM0());         \
        } else {                                                                                                     \
            W4_ROW(1, 4, W4_NOP, W4_DMA(srd1, dst1, 11), W4_NOP, W4_DMA(srd1, dst1, 12), W4_NOP, W4_DMA(srd1, dst1, 13), W4_NOP, W4_DMA(srd1, dst1, 14)); \
            W4_ROW(1, 5, W4_NOP, W4_DMA(srd1, dst1, 15), W4_NOP, W4_NOP, W4_NOP, W4_NOP, W4_NOP, W4_NOP);            \
            W4_ROW_PLAIN(1, 6); W4_ROW_PLAIN(1, 7);                                                                  \
        }                                                                                                            \
        cur ^= W4_STAGE;                                                                                             \
    }

        int t = 0;
        for (; t + 2 < nkt; ++t) W4_BODY(0)
        W4_BODY(1)                                                       // t = nkt - 2
        W4_BODY(2)                                                       // t = nkt - 1
#undef W4_BODY
#ifdef W4_TIMELINE
        { uint64_t now_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)); tl[6] += (uint32_t)now_ - tl_start; tl[7] += (uint32_t)nkt; tl_tile = (uint32_t)now_; }
#endif

        // ---- epilogue: lane (row slot q = l&15, column group fq = l>>4) holds, for M-tile i, row 8 q + i of its wave's 128 and,
        // in register r of N-tile j, column 8 (4 fq + r) + j: the eight N-tiles give 8 consecutive columns -> 16-byte stores
        const int frow = lane & 15, fq = lane >> 4;
        const bool vec_ok = ((ldc & 7) == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int gm = m0 + wm * 128 + frow * 8 + i;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gn = n0 + wn * 128 + (4 * fq + r) * 8;
                if (gm >= M || gn >= N) continue;
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = acc[i][j][r];
                if (bias) {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (gn + j < N) v[j] += to_f32(bias[gn + j]);
                }
                TO* dstp = C + (int64_t)gm * ldc + gn;
                if (vec_ok && gn + 7 < N) {
                    if constexpr (sizeof(TO) == 4) {
                        *reinterpret_cast<f32x4*>(dstp) = f32x4{v[0], v[1], v[2], v[3]};
                        *reinterpret_cast<f32x4*>(dstp + 4) = f32x4{v[4], v[5], v[6], v[7]};
                    } else {
                        bf16x8 o;
#pragma unroll
                        for (int j = 0; j < 8; ++j) o[j] = (bf16_t)v[j];
                        *reinterpret_cast<bf16x8*>(dstp) = o;      // (non-temporal 16-byte stores: -15 %, measured)
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (gn + j < N) dstp[j] = from_f32<TO>(v[j]);
                }
            }
        }
#ifdef W4_TIMELINE
        { uint64_t now_; asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)); tl[9] += (uint32_t)now_ - tl_tile; tl_tile = (uint32_t)now_; tl[10] += 1; tl[11] = (uint32_t)now_ - tl_kernel; }   // 9: epilogue incl. store drain
        if (lane == 0 && !has_next) {                                     // over the wave's first output row (dev build only)
            uint32_t* w = reinterpret_cast<uint32_t*>(C + (int64_t)(m0 + wm * 128) * ldc + n0 + wn * 128);
#pragma unroll
            for (int i = 0; i < 12; ++i) w[i] = tl[i];
        }
#endif
        m0 = nm0;
        n0 = nn0;
    }
#undef W4_DMA16
#undef W4_ROW_PLAIN
#undef W4_ROW
#undef W4_NOP
#undef W4_MM
#undef W4_RDA
#undef W4_RDB
#undef W4_DMA
#undef W4_WAIT_LGKM0
#undef W4_WAIT_VM
#undef W4_BARRIER
#undef W4_SWAP
#undef W4_TS
#undef W4_DSTOFF
#undef W4_RS
}

template <typename TO>
int launch_w4(const void* A, const void* B, void* C, const void* bias, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc,
              hipStream_t st) {
    const int tiles_m = (M + 255) / 256, tiles_n = (N + 255) / 256;
    const size_t lds = 2 * (size_t)W4_STAGE;
    auto kern = gemm_nt_w4_kernel<TO>;
    static int ncu = 0;
    if (!ncu) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        int dev = 0, n = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        ncu = n > 0 ? n : 256;
    }
    const int ntile = tiles_m * tiles_n;
    // persistent: one workgroup per CU (130 KB of LDS and the whole register file each); a multiple of 8 keeps the XCD remap exact
    int grid = ntile < ncu ? ntile : ncu;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, (const bf16_t*)A, (const bf16_t*)B, (TO*)C, (const bf16_t*)bias, M, N, K,
                       lda, ldb, ldc, tiles_m, tiles_n);
    return lrp_check_launch();
}

}  // namespace

// bf16 operands, K a whole number (>= 2) of 64-element tiles, 128 rows of either operand within 2^31 bytes (the buffer resource's range)
bool lrp_gemm_w4_ok(int M, int N, int K, int64_t lda, int64_t ldb) {
    return K >= 128 && (K % 64) == 0 && 128 * lda * 2 < (int64_t)1 << 31 && 128 * ldb * 2 < (int64_t)1 << 31;
}
int lrp_gemm_w4(const void* A, const void* B, void* C, const void* bias, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc,
                int out_dtype, hipStream_t st) {
    if (out_dtype == LRP_F32) return launch_w4<float>(A, B, C, bias, M, N, K, lda, ldb, ldc, st);
    return launch_w4<bf16_t>(A, B, C, bias, M, N, K, lda, ldb, ldc, st);
}
