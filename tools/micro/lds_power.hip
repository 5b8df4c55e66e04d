// dev micro-benchmark (round 3): what do LDS fragment reads cost in POWER next to a saturated MFMA pipe?  Ping-pong skeleton of the GEMM
// (two groups of four waves alternate a 16-MFMA burst with an L phase) where the L phase issues R ds_read_b128 per wave (R = 0..24) of
// random data into the MFMA operand registers.  The clock the chip settles at (s_memtime / s_memrealtime) and the TFLOP/s give the price
// of LDS traffic at a constant 100 % matrix-pipe occupancy.   hipcc --offload-arch=gfx950 -O3 lds_power.hip -o lds_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define S2(x) #x
#define S(x) S2(x)
#define MF32(i, A, B) "v_mfma_f32_32x32x16_bf16 v[" S(i) ":" S(i+15) "], v[" S(A) ":" S(A+3) "], v[" S(B) ":" S(B+3) "], v[" S(i) ":" S(i+15) "]\n\t"
#define BURST32 MF32(0,128,144) MF32(16,132,144) MF32(32,128,148) MF32(48,132,148) MF32(64,128,152) MF32(80,132,152) MF32(96,128,156) MF32(112,132,156) \
                MF32(0,136,160) MF32(16,140,160) MF32(32,136,164) MF32(48,140,164) MF32(64,136,168) MF32(80,140,168) MF32(96,136,172) MF32(112,140,172)
#define CLOB "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "memory"
#define RD(r, off) "ds_read_b128 v[" S(r) ":" S(r+3) "], %0 offset:" S(off) "\n\t"
#define RD4(r, off) RD(r, off) RD(r+4, off+4096) RD(r+8, off+8192) RD(r+12, off+12288)

template <int R>
__global__ __launch_bounds__(512, 2) void k(unsigned long long* out, const uint4* rnd, int iters, int gap_every, int gap_sleeps, uint4* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // fill 64 KiB of LDS with random bf16
    for (int i = threadIdx.x; i < 4096; i += 512) ((uint4*)smem)[i] = rnd[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    // conflict-free fragment address (128-byte rows, chunk swizzle (row >> 1) & 7), as the GEMM
    unsigned addr = (unsigned)((lane & 31) * 128 + ((((lane >> 5)) ^ ((lane >> 1) & 7)) << 4));
    asm volatile(RD4(128, 0) RD4(144, 16384) RD4(160, 32768) "s_waitcnt lgkmcnt(0)" ::"v"(addr) : CLOB);
    const int g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
    if (g == 1) asm volatile("s_barrier" ::: "memory");
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (R >= 4) asm volatile(RD4(128, 0) ::"v"(addr) : CLOB);
        if constexpr (R >= 8) asm volatile(RD4(144, 16384) ::"v"(addr) : CLOB);
        if constexpr (R >= 12) asm volatile(RD4(160, 32768) ::"v"(addr) : CLOB);
        if constexpr (R >= 16) asm volatile(RD4(128, 49152) ::"v"(addr) : CLOB);
        if constexpr (R >= 20) asm volatile(RD4(144, 1024) ::"v"(addr) : CLOB);
        if constexpr (R >= 24) asm volatile(RD4(160, 17408) ::"v"(addr) : CLOB);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier\n\ts_setprio 1\n\t" BURST32 "s_setprio 0\n\ts_barrier" ::: CLOB);
        if (gap_every && (it % gap_every) == gap_every - 1) {            // a synchronized idle gap in every workgroup (tile boundary model)
            if (sink) {                                                  // ... with a 128-KiB store burst per workgroup
                for (int q = 0; q < 16; ++q) sink[((size_t)blockIdx.x * 16 + q) * 512 + threadIdx.x] = rnd[threadIdx.x];
            }
            for (int q = 0; q < gap_sleeps; ++q) asm volatile("s_sleep 127" ::: "memory");
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (g == 0) asm volatile("s_barrier" ::: "memory");
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
}

template <int R> void run(unsigned long long* d, const uint4* rnd, int gap_every = 0, int gap_sleeps = 0, uint4* sink = nullptr) {
    const int iters = 40000;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<R>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        k<R><<<256, 512, 65536>>>(d, rnd, iters, gap_every, gap_sleeps, sink);
        (void)hipEventRecord(e1);
        if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) { printf("launch failed\n"); return; }
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c[2] = {0, 0};
        (void)hipMemcpy(c, d, 16, hipMemcpyDeviceToHost);
        const double flops = 256.0 * 8 * iters * 16.0 * 32768.0;
        if (rep == 2)
            printf("gap: every %3d phase pairs, %2d x s_sleep 127 (%5d cycles)%s | ", gap_every, gap_sleeps, gap_sleeps * 127 * 64, sink ? " + 128 KiB stores" : ""),
            printf("R = %2d ds_read_b128 per wave per 16-MFMA phase (%4.2f KiB per MFMA): %6.0f TFLOP/s  clock %.3f GHz  %.0f cycles per phase pair (ideal 1024)\n",
                   R, R / 16.0, flops / (ms * 1e-3) / 1e12, (double)c[0] / (double)c[1] * 0.1, (double)c[0] / iters);
    }
}

int main(int argc, char** argv) {
    unsigned long long* d;
    (void)hipMalloc(&d, 64);
    const int zero = argc > 1 && atoi(argv[1]) == 0;
    std::vector<unsigned short> h(65536 / 2);
    srand(1);
    for (auto& x : h) {
        float f = (float)rand() / RAND_MAX * 2.f - 1.f;
        unsigned u; memcpy(&u, &f, 4);
        x = zero ? 0 : (unsigned short)(u >> 16);
    }
    uint4* rnd;
    (void)hipMalloc(&rnd, 65536);
    (void)hipMemcpy(rnd, h.data(), 65536, hipMemcpyHostToDevice);
    printf("LDS contents: %s\n", zero ? "zero" : "uniform random bf16 in [-1,1)");
    run<0>(d, rnd); run<12>(d, rnd); run<24>(d, rnd);
    uint4* sink;
    (void)hipMalloc(&sink, (size_t)256 * 16 * 512 * 16);
    for (int sl : {1, 2, 4}) run<12>(d, rnd, 128, sl);
    run<12>(d, rnd, 128, 0, sink);
    run<12>(d, rnd, 128, 1, sink);
    run<12>(d, rnd, 32, 1);
    run<12>(d, rnd, 512, 1);
    return 0;
}
