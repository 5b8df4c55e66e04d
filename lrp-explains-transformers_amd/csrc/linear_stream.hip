// linear_stream.hip -- K1 (Linear forward z = x W^T + b, ref lxt/explicit/functional.py:345-351, rules.py:188-205) in its HBM-BOUND regime,
// M <= 256 rows (bf16 arithmetic intensity ~2 M FLOP/B), as ONE launch: a narrow-N, FULL-K weight-streaming MFMA kernel.
//
// Why not the split-K ping-pong GEMM (lrp_gemm_skinny) here: it needs a second launch that sums fp32 slabs (5-9 us against a 15-us weight
// stream) and its 256-row tile multiplies 240 rows of zeros at M = 16.  Here a workgroup (4 waves) owns 64 rows of W and the whole K range:
//   * W is read ONCE: wave w owns W rows n0 + 16 w .. + 15 and stages them itself, K tile by K tile, into a WAVE-PRIVATE LDS ring of
//     LS_WD = 8 tiles with buffer_load_dwordx4 .. lds nt -- 2 pieces of 8 rows x 128 B per tile, i.e. FULL 128-byte lines per row visit
//     (version 1 of this kernel loaded the MFMA operand layout straight into registers: 16 rows x 64 B per wave instruction, half
//     lines -- 5.2 TB/s on the 1-GB LM head where the full-line staging of the split-K path reaches 6.05; profiles/r04_call2_*.txt).
//     16 KiB per wave = 64 KiB per workgroup in flight: the ~50 KB per CU that 25 GB/s per CU x ~2 us of loaded HBM latency asks for.
//     No barrier guards W: only the staging wave reads its pieces, its own vmcnt covers them;
//   * x (M rows, L2-resident, shared by the 4 waves) is staged per K tile through an LDS ring by buffer_load .. lds (8 rows x 128 B per
//     wave instruction, chunk ^ (row & 7) swizzle on the source side: the A-operand image of gemm_pp.hip, conflict-free for the
//     16-row ds_read_b128 fragments), XD tiles ahead, ring of XD + 2 tiles, ONE s_barrier per K tile;
//   * every wait is a hand-counted s_waitcnt: loads retire in order, so "x(t) has landed" = "all but the XD (P + 2) younger VMEM operations
//     have retired" -- which also covers W(t), issued 8 tiles earlier (2 W pieces + P x pieces per wave and K tile).  The last 8 tiles are peeled: nothing is fetched past K.
//   * 16-row blocks past M are not multiplied (nb live blocks of the MBMAX the instantiation has accumulators for).
// Grid: ceil(N / (4 rw)) workgroups, rw = W rows per wave (8 .. 16, stream_rows_per_wave: whole rounds of the 256 CUs); the host uses the kernel when
// ceil(N / 64) >= 192 and K is a multiple of 512.
// D = mfma(Wfrag, xfrag): lane l holds z[m = 16 i + (l & 15)][n = n0 + 16 w + 4 (l >> 4) + e].
#include "common.hpp"

namespace {

constexpr int LS_KT = 64;      // K elements per K tile
constexpr int LS_WD = 8;       // K tiles of W in flight per wave (register ring)
typedef __attribute__((address_space(3))) void* ls_lds_ptr_t;

template <int MBMAX> struct LSCfg {
    static constexpr int XD = 3;                                   // x prefetch distance in K tiles
    static constexpr int NBUF = XD + 2;                            // LDS ring: tile t + XD is written while tiles t - 1, t may still be read
    static constexpr int P = MBMAX / 2;                            // 1-KiB staging pieces (8 rows x 128 B) per wave per K tile
    static constexpr int XTILE = MBMAX * 16 * 128;                 // bytes of one x tile
    static constexpr int BS = MBMAX < 4 ? MBMAX : 4;               // row blocks per LDS read batch
    static constexpr int NBATCH = MBMAX / BS;
    static constexpr int VMC = XD * (P + 2);                       // VMEM operations younger than x(t) at the wait of tile t (steady state)
};
// the same count inside the peeled last block (static s = position in the block; no W loads, x stages only while t + XD < nkt)
template <int MBMAX> constexpr int ls_last_vmc(int s) {
    constexpr int XD = LSCfg<MBMAX>::XD, P = LSCfg<MBMAX>::P;
    int c = 0;
    for (int j = s - XD + 1; j <= s; ++j) c += (j <= 7 - XD) ? P : 0;      // x stages issued at (relative) iterations s - XD + 1 .. s
    for (int j = s - XD; j <= s - 1; ++j) c += (j < 0) ? 2 : 0;            // W loads issued at the end of iterations s - XD .. s - 1
    return c;
}

template <int V> struct LSI { static constexpr int value = V; };
template <int I, int N, typename F> LRP_DEVICE void ls_for(F&& f) {
    if constexpr (I < N) {
        f(LSI<I>{});
        ls_for<I + 1, N>(static_cast<F&&>(f));
    }
}

#define LS_DSRD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))

template <typename TO, int MBMAX>
__global__ __launch_bounds__(256, (MBMAX <= 4) ? 2 : 1) void linear_stream_fwd_kernel(
    const bf16_t* __restrict__ x, const bf16_t* __restrict__ W, TO* __restrict__ z, const bf16_t* __restrict__ bias, int M, int N, int K,
    int64_t ldx, int64_t ldw, int64_t ldz, int rw, int kt_per_split, int64_t slab_stride, unsigned* __restrict__ tickets,
    void* __restrict__ fin, int64_t ldf, int fin_f32) {
    using C = LSCfg<MBMAX>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // rw = W rows per wave (8 .. 16): the host picks it so that ceil(N / (4 rw)) workgroups cover the 256 CUs in whole rounds -- N = 14336 with
    // 16 rows per wave is 224 workgroups, and a CU streams at ~23 GB/s whatever the others do: 32 idle CUs are 12.5 % of the bandwidth.  Rows
    // rw .. 15 of the wave's 16-row MFMA block are never fetched (offset beyond num_records: zero fill) and never stored.
    const int n0 = blockIdx.x * (4 * rw);
    // round 5 -- K SPLITS for narrow weights (N = 4096: 64 workgroups of full K would leave three quarters of the chip idle, and the split-K path of
    // the ping-pong GEMM costs a second launch): blockIdx.y owns kt_per_split K tiles and writes an fp32 partial slab; the slabs of a column block
    // are summed by the LAST workgroup to arrive on it (tickets: the protocol of the dgrad kernel below)
    const int kt0 = (int)blockIdx.y * kt_per_split;
    const int nkt = (gridDim.y > 1) ? kt_per_split : K / LS_KT;
    int nb = (M + 15) >> 4;
    nb = nb > MBMAX ? MBMAX : nb;

    // ---- W: raw buffer over the whole weight (rows past N read as zero); this wave's rows n0 + 16 w .. + 15 as two 8-row pieces per K tile
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, (int)(((int64_t)(N - 1) * ldw + K) * 2), 0x00020000);
    const int prow = lane >> 3, pslot = lane & 7;
    int voW[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) voW[p] = (8 * p + prow < rw) ? (int)(prow * ldw * 2) + ((pslot ^ prow) << 4) : 0x40000000;
    const int soW = (int)((int64_t)(n0 + rw * wave) * ldw * 2);
    char* const wring = smem + C::NBUF * C::XTILE + wave * (LS_WD * 2048);       // [LS_WD tiles][16 rows][128 B], wave-private
    // ---- x: LDS-DMA pieces of 8 rows x 128 B; lane l -> row l >> 3, LDS slot l & 7, source chunk slot ^ (row & 7)
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)(((int64_t)(M - 1) * ldx + K) * 2), 0x00020000);
    const int voX = (int)(prow * ldx * 2) + ((pslot ^ prow) << 4);
    auto stage_x = [&](int kt, int buf) {
#pragma unroll
        for (int p = 0; p < C::P; ++p) {
            const int q = wave * C::P + p;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (ls_lds_ptr_t)(smem + buf * C::XTILE + q * 1024), 16, voX,
                                                     (int)((int64_t)(8 * q) * ldx * 2) + (kt + kt0) * 128, 0, 0);
        }
    };
    // fragment of row block i, k-step ks: row 16 i + (l & 15), chunk (4 ks + (l >> 4)) ^ (row & 7)
    const uint32_t lds0 = (uint32_t)(uintptr_t)(ls_lds_ptr_t)smem;
    uint32_t cX[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) cX[ks] = lds0 + (uint32_t)((lane & 15) * 128) + (uint32_t)((((4 * ks + (lane >> 4)) ^ (lane & 7))) << 4);

    // W fragment of ring slot s, k-step ks: same image as x (row l & 15 of the wave's 16 rows)
    uint32_t cW[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) cW[ks] = cX[ks] + (uint32_t)(C::NBUF * C::XTILE + wave * (LS_WD * 2048));
    f32x4 acc[MBMAX];
#pragma unroll
    for (int i = 0; i < MBMAX; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // stage W(kt) into ring slot s: two pieces (rows 0..7, 8..15 of the wave's block); nt: streamed once, by one CU
    auto load_w = [&](auto sc, int kt) {
        constexpr int s = decltype(sc)::value;
#pragma unroll
        for (int p = 0; p < 2; ++p)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (ls_lds_ptr_t)(wring + s * 2048 + p * 1024), 16, voW[p],
                                                     soW + (int)((int64_t)(8 * p) * ldw * 2) + (kt + kt0) * 128, 0, 2);
    };

    // ---- prologue: W(0 .. WD - XD - 1), then [x(v), W(WD - XD + v)] for v = 0 .. XD - 1 (the order the steady state continues)
    ls_for<0, LS_WD - C::XD>([&](auto sc) { load_w(sc, decltype(sc)::value); });
    ls_for<0, C::XD>([&](auto vc) {
        constexpr int v = decltype(vc)::value;
        stage_x(v, v);
        __builtin_amdgcn_sched_barrier(0);
        load_w(LSI<LS_WD - C::XD + v>{}, LS_WD - C::XD + v);
    });

    // one K tile: LAST = inside the peeled final block (static s decides what is still fetched and how much may be outstanding)
    auto tile = [&](auto sc, auto lastc, int t) {
        constexpr int s = decltype(sc)::value;
        constexpr bool LAST = decltype(lastc)::value != 0;
        if constexpr (!LAST || (s + C::XD < 8)) stage_x(t + C::XD, (t + C::XD) % C::NBUF);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (LAST) asm volatile("s_waitcnt vmcnt(%[cnt])" :: [cnt] "n"(ls_last_vmc<MBMAX>(s)) : "memory");
        else asm volatile("s_waitcnt vmcnt(%[cnt])" :: [cnt] "n"(C::VMC) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t xb = (uint32_t)((t % C::NBUF) * C::XTILE);
        const uint32_t a0 = cX[0] + xb, a1 = cX[1] + xb;
        // this wave's W fragments of the tile (ring slot s): two reads, first in the LDS queue
        u32x4 wq0, wq1;
        LS_DSRD(wq0, cW[0], s * 2048);
        LS_DSRD(wq1, cW[1], s * 2048);
        // units u = (ks, batch): BS fragment reads each, read one unit ahead
        u32x4 xf[2][C::BS];
        constexpr int U = 2 * C::NBATCH;
        auto issue = [&](auto uc) {
            constexpr int u = decltype(uc)::value, ks = u / C::NBATCH, b = u % C::NBATCH;
            (void)&xf; (void)&a0; (void)&a1;
            ls_for<0, C::BS>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                (void)&xf; (void)&a0; (void)&a1;
                if constexpr (ks == 0) LS_DSRD(xf[u & 1][i], a0, (b * C::BS + i) * 2048);
                else LS_DSRD(xf[u & 1][i], a1, (b * C::BS + i) * 2048);
            });
        };
        issue(LSI<0>{});
        if constexpr (C::BS == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(wq0), "+v"(wq1));
        else asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(wq0), "+v"(wq1));
        const bf16x8 w0 = __builtin_bit_cast(bf16x8, wq0), w1 = __builtin_bit_cast(bf16x8, wq1);
        ls_for<0, U>([&](auto uc) {
            constexpr int u = decltype(uc)::value, ks = u / C::NBATCH, b = u % C::NBATCH;
            (void)&xf;
            if constexpr (u + 1 < U) {
                issue(LSI<u + 1>{});
                if constexpr (C::BS == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(xf[u & 1][0]), "+v"(xf[u & 1][1]), "+v"(xf[u & 1][2]), "+v"(xf[u & 1][3]));
                else asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(xf[u & 1][0]), "+v"(xf[u & 1][1]));
            } else {
                if constexpr (C::BS == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xf[u & 1][0]), "+v"(xf[u & 1][1]), "+v"(xf[u & 1][2]), "+v"(xf[u & 1][3]));
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xf[u & 1][0]), "+v"(xf[u & 1][1]));
            }
            ls_for<0, C::BS>([&](auto ic) {
                constexpr int i = decltype(ic)::value, blk = b * C::BS + i;
                if (blk < nb)
                    acc[blk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ks == 0 ? w0 : w1, __builtin_bit_cast(bf16x8, xf[u & 1][i]), acc[blk], 0, 0, 0);
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (!LAST) load_w(sc, t + LS_WD);
        __builtin_amdgcn_sched_barrier(0);
    };

    int t0 = 0;
    for (; t0 + LS_WD < nkt; t0 += LS_WD)
        ls_for<0, LS_WD>([&](auto sc) { tile(sc, LSI<0>{}, t0 + decltype(sc)::value); });
    ls_for<0, LS_WD>([&](auto sc) { tile(sc, LSI<1>{}, t0 + decltype(sc)::value); });

    // ---- epilogue: lane holds z[16 i + (l & 15)][n0 + 16 w + 4 (l >> 4) + e]
    const int c16 = 4 * (lane >> 4);                                  // column of the wave's 16-row block: live while < rw
    const int ncol = n0 + rw * wave + c16;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (c16 + e < rw && ncol + e < N) bv[e] = to_f32(bias[ncol + e]);
    }
    const bool vec = (c16 + 3 < rw) && (ncol + 3 < N) && ((ncol & 3) == 0) && ((ldz & 3) == 0) && ((reinterpret_cast<uintptr_t>(z) & 15) == 0);
    if constexpr (sizeof(TO) == 4) {
        if (tickets != nullptr) {
            // ---- split K: publish the partial quad (the host guarantees whole, aligned quads: rw = 16, N % 64 == 0), take a ticket, and the last
            // arriver of the column block sums the slabs in slab order, adds the bias and writes z
            float* const slab = reinterpret_cast<float*>(z) + (int64_t)blockIdx.y * slab_stride;
#pragma unroll
            for (int i = 0; i < MBMAX; ++i) {
                const int m = 16 * i + (lane & 15);
                if (i < nb && m < M) {
                    uint64_t* d8 = reinterpret_cast<uint64_t*>(slab + (int64_t)m * ldz + ncol);
                    const f32x2 lo = {acc[i][0], acc[i][1]}, hi2 = {acc[i][2], acc[i][3]};
                    __hip_atomic_store(d8, __builtin_bit_cast(uint64_t, lo), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(d8 + 1, __builtin_bit_cast(uint64_t, hi2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                              // every wave's slab stores have left; the LDS rings are no longer read
            unsigned* tk = reinterpret_cast<unsigned*>(smem);
            if (threadIdx.x == 0) tk[0] = __hip_atomic_fetch_add(tickets + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            const unsigned nsp = gridDim.y;
            if (tk[0] != nsp - 1) return;
            // last arriver.  Ordering: the producers' side is the hardware path -- sc1 write-through stores, drained by vmcnt(0) and a
            // workgroup barrier BEFORE the ticket atomic is issued, so the slabs are at the device coherence point when the ticket moves (a RELEASE
            // on the ticket would add a whole-L2 write-back per workgroup: round 3 measured that at more than the second launch it replaces); the
            // consumer's side is made explicit: an agent-scope ACQUIRE fence (one L2 / vector-cache invalidate per column block) ahead of the
            // slab loads, which are agent-scope atomics the compiler may not hoist above it
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if (threadIdx.x == 0) __hip_atomic_store(tickets + blockIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm for the next launch
#pragma unroll
            for (int i = 0; i < MBMAX; ++i) {
                const int m = 16 * i + (lane & 15);
                if (i < nb && m < M) {
                    const float* src = reinterpret_cast<const float*>(z) + (int64_t)m * ldz + ncol;
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        if (k < (int)nsp) {
                            uint64_t* s8 = reinterpret_cast<uint64_t*>(const_cast<float*>(src + (int64_t)k * slab_stride));
                            const f32x2 lo = __builtin_bit_cast(f32x2, __hip_atomic_load(s8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                            const f32x2 hi2 = __builtin_bit_cast(f32x2, __hip_atomic_load(s8 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                            const f32x4 pk = {lo[0], lo[1], hi2[0], hi2[1]};
                            v = (k == 0) ? pk : v + pk;
                        }
                    v += bv;
                    if (fin_f32) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(fin) + (int64_t)m * ldf + ncol) = v;
                    else {
                        bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                        *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(fin) + (int64_t)m * ldf + ncol) = o;
                    }
                }
            }
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < MBMAX; ++i) {
        const int m = 16 * i + (lane & 15);
        if (i < nb && m < M) {
            const f32x4 v = acc[i] + bv;
            TO* dst = z + (int64_t)m * ldz + ncol;
            if (vec) {
                if constexpr (sizeof(TO) == 4) *reinterpret_cast<f32x4*>(dst) = v;
                else {
                    bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                    *reinterpret_cast<bf16x4*>(dst) = o;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (c16 + e < rw && ncol + e < N) dst[e] = from_f32<TO>(v[e]);
            }
        }
    }
}

// rows of W per wave: the count in 8 .. 16 that covers N in the fewest whole rounds of the 256 CUs (cost = rounds x rows per workgroup; ties
// go to the larger block).  N = 14336: 14 (256 workgroups instead of 224); N = 28672: 14 (512 = two rounds instead of 448 = 1.75).
inline int stream_rows_per_wave(int N) {
    int best = 16, best_cost = ((((N + 63) / 64) + 255) / 256) * 16;
    for (int rw = 15; rw >= 8; --rw) {
        const int wgs = (N + 4 * rw - 1) / (4 * rw), cost = ((wgs + 255) / 256) * rw;
        if (cost < best_cost) { best = rw; best_cost = cost; }
    }
    return best;
}

// =====================================================================================================================
// Redistribution half of the eps-rule in the same regime: c[M, Kout] = s[M, N] . W[N, Kout] from the STORED weight (contraction over W's
// rows; ref lxt/explicit/functional.py:355-364, rules.py:206-222), M <= 64 rows.
//   * a workgroup owns 64 OUTPUT columns (128-byte segments of every W row it visits: full lines) and one of `splits` contiguous ranges of
//     contraction rows; tiles of 128 rows, wave w takes rows 32 w .. + 31 of a tile = ONE 32-deep MFMA step for all 64 columns and all row
//     blocks of s.  Its operands -- 4 W pieces of 8 rows x 128 B, MBMAX pieces of s (16 rows x 64 B) -- go through WAVE-PRIVATE LDS rings
//     (D = 4 tiles, 16 KiB of W in flight per wave): no barrier inside the contraction loop, only the wave's own counted s_waitcnt.
//   * the MFMA's W operand (8 consecutive contraction rows of one output column) is gathered from the row-major image by two
//     ds_read_b64_tr_b16 (image [32 rows][128 B], 16-byte chunk c of row r at c ^ 2 (((r >> 1) & 1) + 2 ((r >> 3) & 1)): conflict-free for
//     the 32-lane transpose-read groups); s fragments are plain ds_read_b128 (image [16 rows][64 B], chunk c of row r at c ^ (-(r >> 2) & 3)).
//   * the four waves' partial sums meet in LDS once, at the end; splits > 1 (Kout = 4096: 64 column blocks x 4) writes fp32 slabs that
//     the split-K reduce kernel of gemm.hip sums in slab order (deterministic), splits = 1 (Kout >= 14336) writes the result directly.
// =====================================================================================================================
constexpr int LD_D = 4;        // tiles in flight per wave

#define LS_DSTR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))

// SC: the eps-rule's stabiliser formed on the fly -- the operand is g (incoming gradient or relevance) and z (the Linear's forward output)
// travels beside it: s = g z / (z + eps) (gradient form) or g / (z + eps) (relevance form, rel_in), rounded to bf16: no separate launch, no s
// round trip (ref lxt/explicit/functional.py:355-358).  The quotient uses v_rcp_f32 (1 ulp of fp32, then the bf16 rounding) where the stand-alone
// lrp_eps_scale divides exactly: the two can differ in the last bf16 bit of an element.  z with eps = 0 is refused (the factor is exactly 1: pass
// z = NULL; with a relevance operand the value at z = 0 is undefined).
template <typename TO, int MBMAX, bool SC>
__global__ __launch_bounds__(256, 1) void linear_stream_dgrad_kernel(
    const bf16_t* __restrict__ sm, const bf16_t* __restrict__ zm, const bf16_t* __restrict__ W, TO* __restrict__ c, int M, int N, int Kout,
    int64_t lds_, int64_t ldz, int64_t ldw, int64_t ldc, int tiles_per_split, int64_t slab_stride, float eps, int rel_in,
    unsigned* __restrict__ tickets, void* __restrict__ fin, int64_t ldf, int fin_f32) {
    constexpr int OPS = 4 + (SC ? 2 : 1) * MBMAX;        // VMEM operations per tile and wave
    constexpr int WSLOT = 4096, SSLOT = MBMAX * 1024, SLOT = WSLOT + (SC ? 2 : 1) * SSLOT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cb = blockIdx.x, sp = blockIdx.y;
    const int n_beg = sp * tiles_per_split * 128;
    const int nt = tiles_per_split;
    TO* const c0 = c;                                   // slab 0 (in-kernel reduction)
    c += (int64_t)sp * slab_stride;
    int nb = (M + 15) >> 4;
    nb = nb > MBMAX ? MBMAX : nb;
    char* const ring = smem + wave * (LD_D * SLOT);

    // ---- staging.  W piece p: rows 8 p + (l >> 3) of the wave's 32, LDS slot l & 7, source chunk slot ^ g(row)
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, (int)(((int64_t)(N - 1) * ldw + Kout) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc((void*)sm, 0, (int)(((int64_t)(M - 1) * lds_ + N) * 2), 0x00020000);
    int voW[2];
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) voW[pp] = (int)((lane >> 3) * ldw * 2) + (((lane & 7) ^ (2 * (((lane >> 4) & 1) + 2 * pp))) << 4);
    // s piece i: rows 16 i + (l >> 2), LDS slot l & 3, source chunk slot ^ f(row), f(r) = (-(r >> 2)) & 3
    const int voS = (int)((lane >> 2) * lds_ * 2) + (((lane & 3) ^ ((-(lane >> 4)) & 3)) << 4);
    const __amdgpu_buffer_rsrc_t rsZ = __builtin_amdgcn_make_buffer_rsrc((void*)(SC ? zm : sm), 0, (int)(((int64_t)(M - 1) * (SC ? ldz : lds_) + N) * 2), 0x00020000);
    const int voZ = (int)((lane >> 2) * ldz * 2) + (((lane & 3) ^ ((-(lane >> 4)) & 3)) << 4);
    auto issue = [&](int t, int slot) {
        const int r0 = n_beg + t * 128 + wave * 32;
        char* dst = ring + slot * SLOT;
#pragma unroll
        for (int p = 0; p < 4; ++p)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (ls_lds_ptr_t)(dst + p * 1024), 16, voW[p & 1],
                                                     (int)((int64_t)(r0 + 8 * p) * ldw * 2) + cb * 128, 0, 2);
#pragma unroll
        for (int i = 0; i < MBMAX; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsS, (ls_lds_ptr_t)(dst + WSLOT + i * 1024), 16, voS,
                                                     (int)((int64_t)(16 * i) * lds_ * 2) + r0 * 2, 0, 0);
        if constexpr (SC) {
#pragma unroll
            for (int i = 0; i < MBMAX; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsZ, (ls_lds_ptr_t)(dst + WSLOT + SSLOT + i * 1024), 16, voZ,
                                                         (int)((int64_t)(16 * i) * ldz * 2) + r0 * 2, 0, 0);
        }
    };
    // ---- read addresses (relative to a slot).  W operand of column tile j, half h: row 8 hi + 4 h + (i16 >> 2), chunk (2 j + ((i16 & 3) >> 1)) ^ g
    const int hi = lane >> 4, i16 = lane & 15;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(ls_lds_ptr_t)smem + (uint32_t)(wave * (LD_D * SLOT));
    const uint32_t aW = (uint32_t)((8 * hi + (i16 >> 2)) * 128) + (uint32_t)(((((i16 & 3) >> 1) ^ (2 * (((i16 >> 3) & 1) + 2 * (hi & 1))))) << 4) + 8u * (i16 & 1);
    const uint32_t aS = (uint32_t)(WSLOT + i16 * 64) + (uint32_t)(((hi ^ ((-(i16 >> 2)) & 3))) << 4);

    f32x4 acc[MBMAX][4];
#pragma unroll
    for (int i = 0; i < MBMAX; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int t = 0; t < LD_D && t < nt; ++t) issue(t, t);
    for (int t = 0; t < nt; ++t) {
        const int slot = t & (LD_D - 1);
        const int rem = nt - 1 - t;                      // tiles issued after this one (at most D - 1)
        if (rem >= LD_D - 1) asm volatile("s_waitcnt vmcnt(%[c])" :: [c] "n"((LD_D - 1) * OPS) : "memory");
        else if (rem == 2) asm volatile("s_waitcnt vmcnt(%[c])" :: [c] "n"(2 * OPS) : "memory");
        else if (rem == 1) asm volatile("s_waitcnt vmcnt(%[c])" :: [c] "n"(OPS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t base = lds0 + (uint32_t)(slot * SLOT);
        const uint32_t w0 = base + aW, s0 = base + aS;
        // the wave's W operand of its four column tiles (two transpose reads each), then the s fragments
        u32x2 tw[4][2];
        u32x4 sf[MBMAX], zf[MBMAX];
        ls_for<0, 4>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            (void)&tw; (void)&w0;
            uint32_t a_ = w0;
            if constexpr (j > 0) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(a_) : "n"(j << 5), "v"(w0));
            LS_DSTR(tw[j][0], a_, 0);
            LS_DSTR(tw[j][1], a_, 512);
        });
        ls_for<0, MBMAX>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            (void)&sf; (void)&s0;
            LS_DSRD(sf[i], s0, i * 1024);
        });
        if constexpr (SC) {
            ls_for<0, MBMAX>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                (void)&zf; (void)&s0;
                LS_DSRD(zf[i], s0, SSLOT + i * 1024);
            });
            if constexpr (MBMAX == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(zf[0]), "+v"(zf[1]));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(zf[0]), "+v"(zf[1]), "+v"(zf[2]), "+v"(zf[3]));
        }
        if constexpr (MBMAX == 2)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(tw[0][0]), "+v"(tw[0][1]), "+v"(tw[1][0]), "+v"(tw[1][1]), "+v"(tw[2][0]), "+v"(tw[2][1]),
                         "+v"(tw[3][0]), "+v"(tw[3][1]), "+v"(sf[0]), "+v"(sf[1]));
        else
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(tw[0][0]), "+v"(tw[0][1]), "+v"(tw[1][0]), "+v"(tw[1][1]), "+v"(tw[2][0]), "+v"(tw[2][1]),
                         "+v"(tw[3][0]), "+v"(tw[3][1]), "+v"(sf[0]), "+v"(sf[1]), "+v"(sf[2]), "+v"(sf[3]));
        __builtin_amdgcn_sched_barrier(0);
        // the slot is free as soon as its operands sit in registers: refill it before the MFMAs
        if (t + LD_D < nt) issue(t + LD_D, slot);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (SC) {
            // s = g z / (z + eps) or g / (z + eps), element-wise on the fragments (8 bf16 per lane and row block), rounded to bf16
#pragma unroll
            for (int i = 0; i < MBMAX; ++i)
                if (i < nb) {
                    const bf16x8 gv = __builtin_bit_cast(bf16x8, sf[i]), zv = __builtin_bit_cast(bf16x8, zf[i]);
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float zz = (float)zv[e];
                        const float r = __builtin_amdgcn_rcpf(zz + eps);
                        o[e] = (bf16_t)((float)gv[e] * (rel_in ? 1.f : zz) * r);
                    }
                    sf[i] = __builtin_bit_cast(u32x4, o);
                }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u32x4 wv = {tw[j][0][0], tw[j][0][1], tw[j][1][0], tw[j][1][1]};
            const bf16x8 wfrag = __builtin_bit_cast(bf16x8, wv);
#pragma unroll
            for (int i = 0; i < MBMAX; ++i)
                if (i < nb) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfrag, __builtin_bit_cast(bf16x8, sf[i]), acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- the four waves' partial sums (disjoint contraction rows) meet in LDS: wave w then owns column tile j = w
    __syncthreads();                                     // every wave is done with its ring (all LDS-DMA landed: each waited vmcnt(0) on its last tile)
    f32x4* red = reinterpret_cast<f32x4*>(smem);          // [wave][i][j][lane]
#pragma unroll
    for (int i = 0; i < MBMAX; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) red[((wave * MBMAX + i) * 4 + j) * 64 + lane] = acc[i][j];
    __syncthreads();
    const int col = cb * 64 + 16 * wave + 4 * hi;
#pragma unroll
    for (int i = 0; i < MBMAX; ++i) {
        const int m = 16 * i + i16;
        if (i < nb && m < M) {
            f32x4 v = red[((0 * MBMAX + i) * 4 + wave) * 64 + lane];
#pragma unroll
            for (int w2 = 1; w2 < 4; ++w2) v += red[((w2 * MBMAX + i) * 4 + wave) * 64 + lane];
            TO* dst = c + (int64_t)m * ldc + col;
            if constexpr (sizeof(TO) == 4) {
                if (tickets != nullptr) {                // slab of the in-kernel reduction: agent-scope stores (write-through: visible to the last
                    uint64_t* d8 = reinterpret_cast<uint64_t*>(dst);        // arriver's agent-scope loads whatever XCD it runs on)
                    const f32x2 lo = {v[0], v[1]}, hi2 = {v[2], v[3]};
                    __hip_atomic_store(d8, __builtin_bit_cast(uint64_t, lo), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(d8 + 1, __builtin_bit_cast(uint64_t, hi2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    continue;
                }
            }
            if (col + 3 < Kout && (ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(c) & 15) == 0) {
                if constexpr (sizeof(TO) == 4) *reinterpret_cast<f32x4*>(dst) = v;
                else {
                    bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                    *reinterpret_cast<bf16x4*>(dst) = o;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (col + e < Kout) dst[e] = from_f32<TO>(v[e]);
            }
        }
    }
    // ---- in-kernel reduction of the contraction splits (round 5; replaces the second launch: ~1.5 us of kernel boundary + 3.4-5.6 us of reduce kernel
    // for 1-5 MB of slabs).  Every split's workgroup publishes its fp32 partial tile write-through (sc0 sc1 stores, drained), takes a ticket of
    // its column block; the LAST arriver (ticket == splits - 1; it re-arms the word to 0 for the next launch) sums the slabs in
    // slab order -- its own from memory as well, so the result does not depend on who arrives last -- and writes the output.  Other workgroups'
    // slabs are read with sc0 sc1 loads: lines this workgroup's XCD never held in this launch (MI355X_MICROARCH.md: "sc0 sc1 stores and loads both
    // sides" is a valid inter-workgroup form on non-coherent per-XCD L2s).
    if constexpr (sizeof(TO) == 4) {
        if (tickets != nullptr) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                              // every wave's slab stores have left; `red` is no longer read
            unsigned* tk = reinterpret_cast<unsigned*>(smem);
            if (threadIdx.x == 0) tk[0] = __hip_atomic_fetch_add(tickets + cb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            const unsigned nsp = gridDim.y;
            if (tk[0] != nsp - 1) return;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");          // (see lrp_linear_stream_fwd_tk's last arriver above)
            if (threadIdx.x == 0) __hip_atomic_store(tickets + cb, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // all arrivals of this launch are in: re-arm
#pragma unroll
            for (int i = 0; i < MBMAX; ++i) {
                const int m = 16 * i + i16;
                if (i < nb && m < M) {
                    // (compiler-visible agent-scope loads, NOT inline asm: an asm load's destination registers are unprotected until one's own
                    // wait -- the register allocator re-used them for the next address while the load was in flight: a memory fault on the box)
                    const float* src = reinterpret_cast<const float*>(c0) + (int64_t)m * ldc + col;
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        if (k < (int)nsp) {
                            uint64_t* s8 = reinterpret_cast<uint64_t*>(const_cast<float*>(src + (int64_t)k * slab_stride));
                            const f32x2 lo = __builtin_bit_cast(f32x2, __hip_atomic_load(s8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                            const f32x2 hi2 = __builtin_bit_cast(f32x2, __hip_atomic_load(s8 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                            const f32x4 pk = {lo[0], lo[1], hi2[0], hi2[1]};
                            v = (k == 0) ? pk : v + pk;
                        }
                    if (fin_f32) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(fin) + (int64_t)m * ldf + col) = v;
                    else {
                        bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                        *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(fin) + (int64_t)m * ldf + col) = o;
                    }
                }
            }
        }
    }
}

// splits of the contraction range: the smallest count that fills the chip (>= 224 workgroups), divides the tile count and leaves >= 4 tiles
inline int dgrad_splits(int N, int Kout) {
    const int cbs = Kout / 64, tiles = N / 128;
    for (int s_ : {1, 2, 3, 4, 6, 8})
        if (cbs * s_ >= 224 && cbs * s_ <= 1024 && tiles % s_ == 0 && tiles / s_ >= LD_D) return s_;      // (validity; the policy is in _ok)
    return 0;
}

// K splits of the forward for narrow weights: the smallest count in {2, 3, 4, 6, 8} that puts 192 ... 1024 workgroups of 64 rows on the chip with
// whole 8-tile rings per split (1: the full-K form applies; 0: neither does)
inline int fwd_splits(int N, int K) {
    if (N < 1 || K < 512 || (K % 512)) return 0;
    const int wgs = (N + 63) / 64;
    if (wgs >= 192) return wgs <= 1024 ? 1 : 0;
    if (N % 64) return 0;                                 // whole quads per lane in the slabs: 16 rows per wave
    for (int s_ : {2, 3, 4, 6, 8})
        if (wgs * s_ >= 192 && wgs * s_ <= 1024 && (K % (512 * s_)) == 0) return s_;
    return 0;
}

template <typename TO, int MBMAX>
int launch_stream(const void* x, const void* W, const void* bias, void* z, int M, int N, int K, int64_t ldx, int64_t ldw, int64_t ldz,
                  hipStream_t st) {
    using C = LSCfg<MBMAX>;
    const size_t lds = (size_t)C::NBUF * C::XTILE + 4 * (size_t)LS_WD * 2048;        // x ring + four wave-private W rings
    auto kern = linear_stream_fwd_kernel<TO, MBMAX>;
    LRP_SET_MAX_LDS(kern, lds);
    const int rw = stream_rows_per_wave(N);
    hipLaunchKernelGGL(kern, dim3((N + 4 * rw - 1) / (4 * rw)), dim3(256), lds, st, (const bf16_t*)x, (const bf16_t*)W, (TO*)z, (const bf16_t*)bias, M,
                       N, K, ldx, ldw, ldz, rw, 0, (int64_t)0, (unsigned*)nullptr, (void*)nullptr, (int64_t)0, 0);
    return lrp_check_launch();
}

// split form: fp32 slabs [splits][M][N] in ws, summed in-kernel by the last arriver of each 64-row block into z
template <int MBMAX>
int launch_stream_split(const void* x, const void* W, const void* bias, void* z, int M, int N, int K, int64_t ldx, int64_t ldw, int64_t ldz,
                        int out_f32, int splits, void* ws, unsigned* tickets, hipStream_t st) {
    using C = LSCfg<MBMAX>;
    const size_t lds = (size_t)C::NBUF * C::XTILE + 4 * (size_t)LS_WD * 2048;
    auto kern = linear_stream_fwd_kernel<float, MBMAX>;
    LRP_SET_MAX_LDS(kern, lds);
    hipLaunchKernelGGL(kern, dim3(N / 64, splits), dim3(256), lds, st, (const bf16_t*)x, (const bf16_t*)W, (float*)ws, (const bf16_t*)bias, M, N, K, ldx,
                       ldw, (int64_t)N, 16, K / LS_KT / splits, (int64_t)M * N, tickets, z, ldz, out_f32);
    return lrp_check_launch();
}

template <typename TO>
int launch_stream_m(const void* x, const void* W, const void* bias, void* z, int M, int N, int K, int64_t ldx, int64_t ldw, int64_t ldz,
                    hipStream_t st) {
    const int nb = (M + 15) / 16;
    if (nb <= 2) return launch_stream<TO, 2>(x, W, bias, z, M, N, K, ldx, ldw, ldz, st);
    if (nb <= 4) return launch_stream<TO, 4>(x, W, bias, z, M, N, K, ldx, ldw, ldz, st);
    return launch_stream<TO, 8>(x, W, bias, z, M, N, K, ldx, ldw, ldz, st);
}

}  // namespace

// can lrp_linear_stream_fwd serve the problem, and is it the right kernel for it?  bf16, 1 <= M <= 128 rows (every workgroup re-reads x from
// L2: beyond that the x traffic outgrows the weight's), K a multiple of 512 (the 8-tile ring), operands below 2^31 bytes, and enough 64-row
// workgroups to put one on (nearly) every CU
extern "C" int lrp_linear_stream_ok(int M, int N, int K, int64_t ldx, int64_t ldw) {
    if (M < 1 || M > 128 || N < 1 || K < 512 || (K % 512)) return 0;
    if ((ldx % 8) || (ldw % 8) || ldx < K || ldw < K) return 0;
    if ((int64_t)N * ldw >= (1ll << 30) || (int64_t)M * ldx >= (1ll << 30)) return 0;
    // enough 64-row workgroups to put one on (nearly) every CU -- with K splits where the weight is narrow (round 5); for very many (the
    // 128256-row LM head: 2004) the split-K path with its 256 x 256 tiles streams ~4 % faster (6.1 vs 5.8 TB/s, profiles/r04_call3_*.txt) and its
    // second launch no longer matters
    return fwd_splits(N, K) >= 1 ? 1 : 0;
}

// K splits lrp_linear_stream_fwd_tk uses for the problem (1: none), its slab workspace in bytes and its ticket words (0 / 0 when not split)
extern "C" int lrp_linear_stream_fwd_splits(int M, int N, int K) { return (M >= 1 && M <= 128) ? fwd_splits(N, K) : 0; }
extern "C" int64_t lrp_linear_stream_fwd_ws(int M, int N, int K) {
    const int s_ = lrp_linear_stream_fwd_splits(M, N, K);
    return s_ > 1 ? (int64_t)s_ * M * N * 4 : 0;
}
extern "C" int lrp_linear_stream_fwd_tickets(int M, int N, int K) { return lrp_linear_stream_fwd_splits(M, N, K) > 1 ? N / 64 : 0; }

extern "C" int lrp_linear_stream_fwd_tk(const void* x, const void* W, const void* bias, void* z, int M, int N, int K, int64_t ldx, int64_t ldw,
                                        int64_t ldz, int dtype, int out_dtype, void* ws, void* tickets, void* stream) {
    if (!x || !W || !z || M < 0 || N < 0 || K < 0) return LRP_EINVAL;
    if (M == 0 || N == 0) return LRP_OK;
    if (dtype != LRP_BF16 || (out_dtype != LRP_BF16 && out_dtype != LRP_F32)) return LRP_ESHAPE;
    if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(W) & 15) || (ldx % 8) || (ldw % 8)) return LRP_EALIGN;
    if (M > 128 || K < 512 || (K % 512) || ldx < K || ldw < K || (int64_t)N * ldw >= (1ll << 30) || (int64_t)M * ldx >= (1ll << 30)) return LRP_ESHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int splits = (ws && tickets) ? fwd_splits(N, K) : 1;          // without the scratch: the full-K form, whatever the workgroup count
    if (splits > 1) {
        if ((reinterpret_cast<uintptr_t>(z) & 15) || (ldz % 4) || (reinterpret_cast<uintptr_t>(ws) & 15)) return LRP_EALIGN;
        const int nb = (M + 15) / 16, f32o = out_dtype == LRP_F32 ? 1 : 0;
        if (nb <= 2) return launch_stream_split<2>(x, W, bias, z, M, N, K, ldx, ldw, ldz, f32o, splits, ws, (unsigned*)tickets, st);
        if (nb <= 4) return launch_stream_split<4>(x, W, bias, z, M, N, K, ldx, ldw, ldz, f32o, splits, ws, (unsigned*)tickets, st);
        return launch_stream_split<8>(x, W, bias, z, M, N, K, ldx, ldw, ldz, f32o, splits, ws, (unsigned*)tickets, st);
    }
    if (out_dtype == LRP_F32) return launch_stream_m<float>(x, W, bias, z, M, N, K, ldx, ldw, ldz, st);
    return launch_stream_m<bf16_t>(x, W, bias, z, M, N, K, ldx, ldw, ldz, st);
}

extern "C" int lrp_linear_stream_fwd(const void* x, const void* W, const void* bias, void* z, int M, int N, int K, int64_t ldx, int64_t ldw,
                                     int64_t ldz, int dtype, int out_dtype, void* stream) {
    return lrp_linear_stream_fwd_tk(x, W, bias, z, M, N, K, ldx, ldw, ldz, dtype, out_dtype, nullptr, nullptr, stream);
}

// fp32 slab reduction (+ cast) of gemm.hip's split-K path, re-used for the dgrad's contraction splits
int lrp_launch_splitk_reduce(const float* ws, void* out, int M, int N, int64_t ldw, int64_t ldo, int splits, int64_t slab, int out_dtype,
                             hipStream_t st);

static bool dgrad_valid(int M, int N, int Kout, int64_t lds_, int64_t ldw) {
    if (M < 1 || M > 64 || N < 512 || (N % 128) || Kout < 64 || (Kout % 64)) return false;
    if ((lds_ % 8) || (ldw % 8) || lds_ < N || ldw < Kout) return false;
    if ((int64_t)N * ldw >= (1ll << 30) || (int64_t)M * lds_ >= (1ll << 30)) return false;
    return dgrad_splits(N, Kout) > 0;
}
// the kernel applies AND is the right one: measured (profiles/r04_call6_stream_dgrad_ab.txt) 1.5-2 us ahead of the split-K skinny path for
// M <= 32 on layer-sized weights with <= 320 workgroups; behind it at M = 64 (four row blocks per wave) and on the 128256-row LM head (6 splits)
extern "C" int lrp_linear_stream_dgrad_ok(int M, int N, int Kout, int64_t lds_, int64_t ldw) {
    if (!dgrad_valid(M, N, Kout, lds_, ldw)) return 0;
    return M <= 32 && (Kout / 64) * dgrad_splits(N, Kout) <= 320;
}

extern "C" int64_t lrp_linear_stream_dgrad_ws(int M, int N, int Kout) {
    if (M < 1 || N < 1 || Kout < 1) return 0;
    const int sp = dgrad_splits(N, Kout);
    return sp > 1 ? (int64_t)sp * M * Kout * 4 : 0;
}

extern "C" int lrp_linear_stream_dgrad_tk(const void* sm, const void* zm, const void* W, void* c, int M, int N, int Kout, int64_t lds_, int64_t ldz,
                                          int64_t ldw, int64_t ldc, float eps, int relevance_in, int dtype, int out_dtype, void* ws, void* tickets_,
                                          void* stream) {
    if (!sm || !W || !c || M < 0 || N < 0 || Kout < 0) return LRP_EINVAL;
    if (zm && eps == 0.f) return LRP_EINVAL;                           // g z rcp(z) is 0 * inf at z = 0: eps = 0 means "no stabiliser", i.e. z = NULL
    if (zm && ((reinterpret_cast<uintptr_t>(zm) & 15) || (ldz % 8) || ldz < N || (int64_t)M * ldz >= (1ll << 30))) return LRP_EALIGN;
    if (M == 0 || Kout == 0) return LRP_OK;
    if (dtype != LRP_BF16 || (out_dtype != LRP_BF16 && out_dtype != LRP_F32)) return LRP_ESHAPE;
    if ((reinterpret_cast<uintptr_t>(sm) & 15) || (reinterpret_cast<uintptr_t>(W) & 15) || (lds_ % 8) || (ldw % 8)) return LRP_EALIGN;
    if (!dgrad_valid(M, N, Kout, lds_, ldw)) return LRP_ESHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int sp = dgrad_splits(N, Kout), tps = N / 128 / sp;
    if (sp > 1 && (!ws || (reinterpret_cast<uintptr_t>(ws) & 15))) return LRP_EINVAL;
    const int mb = ((M + 15) / 16 <= 2) ? 2 : 4;
    if (zm && mb == 4) return LRP_ESHAPE;              // the fused stabiliser doubles the small operand's LDS ring: 32 rows at most
    const size_t lds = 4 * (size_t)LD_D * (4096 + (zm ? 2 : 1) * mb * 1024);
    dim3 grid(Kout / 64, sp), block(256);
    const int64_t slab = (int64_t)M * Kout;
#define LS_LAUNCH_DGRAD(TO, MB, OUT, LDO, TK)                                                                                        \
    {                                                                                                                               \
        if (zm) {                                                                                                                   \
            auto kern = linear_stream_dgrad_kernel<TO, MB, true>;                                                                   \
            LRP_SET_MAX_LDS(kern, lds);                                                                                             \
            hipLaunchKernelGGL(kern, grid, block, lds, st, (const bf16_t*)sm, (const bf16_t*)zm, (const bf16_t*)W, (TO*)(OUT), M, N, Kout, lds_, ldz, \
                               ldw, (int64_t)(LDO), tps, slab, eps, relevance_in, TK, c, ldc, (int)(out_dtype == LRP_F32));        \
        } else {                                                                                                                    \
            auto kern = linear_stream_dgrad_kernel<TO, MB, false>;                                                                  \
            LRP_SET_MAX_LDS(kern, lds);                                                                                             \
            hipLaunchKernelGGL(kern, grid, block, lds, st, (const bf16_t*)sm, (const bf16_t*)nullptr, (const bf16_t*)W, (TO*)(OUT), M, N, Kout, lds_, \
                               (int64_t)0, ldw, (int64_t)(LDO), tps, slab, 0.f, 0, TK, c, ldc, (int)(out_dtype == LRP_F32));        \
        }                                                                                                                           \
    }
    unsigned* const no_tk = nullptr;
    if (sp > 1) {
        // in-kernel reduction when the caller handed a ticket array (one zero-initialised 32-bit word per 64-column block, private to the stream and
        // NEVER written by anything else: the last arriver of a launch re-arms its word) and the output takes 16- / 8-byte row segments; else fp32 slabs + reduce launch
        unsigned* tk = reinterpret_cast<unsigned*>(tickets_);
        const int osz = out_dtype == LRP_F32 ? 4 : 2;
        if (tk && sp <= 8 && (ldc % 4) == 0 && (reinterpret_cast<uintptr_t>(c) % (4 * osz)) == 0) {
            if (mb == 2) LS_LAUNCH_DGRAD(float, 2, ws, Kout, tk) else LS_LAUNCH_DGRAD(float, 4, ws, Kout, tk)
            return lrp_check_launch();
        }
        if (mb == 2) LS_LAUNCH_DGRAD(float, 2, ws, Kout, no_tk) else LS_LAUNCH_DGRAD(float, 4, ws, Kout, no_tk)
        int rc = lrp_check_launch();
        if (rc != LRP_OK) return rc;
        return lrp_launch_splitk_reduce((const float*)ws, c, M, Kout, Kout, ldc, sp, slab, out_dtype, st);
    }
    if (out_dtype == LRP_F32) {
        if (mb == 2) LS_LAUNCH_DGRAD(float, 2, c, ldc, no_tk) else LS_LAUNCH_DGRAD(float, 4, c, ldc, no_tk)
    } else {
        if (mb == 2) LS_LAUNCH_DGRAD(bf16_t, 2, c, ldc, no_tk) else LS_LAUNCH_DGRAD(bf16_t, 4, c, ldc, no_tk)
    }
#undef LS_LAUNCH_DGRAD
    return lrp_check_launch();
}

extern "C" int lrp_linear_stream_dgrad(const void* sm, const void* zm, const void* W, void* c, int M, int N, int Kout, int64_t lds_, int64_t ldz,
                                       int64_t ldw, int64_t ldc, float eps, int relevance_in, int dtype, int out_dtype, void* ws, void* stream) {
    return lrp_linear_stream_dgrad_tk(sm, zm, W, c, M, N, Kout, lds_, ldz, ldw, ldc, eps, relevance_in, dtype, out_dtype, ws, nullptr, stream);
}

// 32-bit ticket words lrp_linear_stream_dgrad_tk needs for the problem (0: the problem is not split, or the kernel does not apply)
extern "C" int lrp_linear_stream_dgrad_tickets(int M, int N, int Kout) {
    if (M < 1 || N < 1 || Kout < 64) return 0;
    return dgrad_splits(N, Kout) > 1 ? Kout / 64 : 0;
}
