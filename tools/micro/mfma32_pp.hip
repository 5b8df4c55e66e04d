// dev micro-benchmark (round 3): issue rate of v_mfma_f32_32x32x16_bf16 streams as the ping-pong GEMM uses them.
//   hipcc --offload-arch=gfx950 -O3 mfma32_pp.hip -o mfma32_pp
// variants: accumulators in AGPRs / VGPRs; one fixed operand pair / the GEMM's 12 distinct fragments; plain stream (1 or 2 waves per
// SIMD) vs the ping-pong skeleton (two groups of four waves alternating 16-MFMA bursts between barriers).
#include <hip/hip_runtime.h>
#include <cstdio>
#define S2(x) #x
#define S(x) S2(x)
// acc tuple i (16 regs) in file F ("a" or "v"), A operand v[A:A+3], B operand v[B:B+3]
#define MF(F, i, A, B) "v_mfma_f32_32x32x16_bf16 " F "[" S(i) ":" S(i+15) "], v[" S(A) ":" S(A+3) "], v[" S(B) ":" S(B+3) "], " F "[" S(i) ":" S(i+15) "]\n\t"
// 16 MFMAs on 8 accumulators, ONE operand pair
#define BURST_SAME(F) MF(F,0,128,132) MF(F,16,128,132) MF(F,32,128,132) MF(F,48,128,132) MF(F,64,128,132) MF(F,80,128,132) MF(F,96,128,132) MF(F,112,128,132) \
                      MF(F,0,128,132) MF(F,16,128,132) MF(F,32,128,132) MF(F,48,128,132) MF(F,64,128,132) MF(F,80,128,132) MF(F,96,128,132) MF(F,112,128,132)
// 16 MFMAs on 8 accumulators, the GEMM's fragment pattern: B frags v[128:135] (j=0,1 of ks0), v[136:143] (ks1); A frags v[144:159] (ks0, i=0..3), v[160:175] (ks1)
#define BURST_GEMM(F) MF(F,0,128,144) MF(F,16,132,144) MF(F,32,128,148) MF(F,48,132,148) MF(F,64,128,152) MF(F,80,132,152) MF(F,96,128,156) MF(F,112,132,156) \
                      MF(F,0,136,160) MF(F,16,140,160) MF(F,32,136,164) MF(F,48,140,164) MF(F,64,136,168) MF(F,80,140,168) MF(F,96,136,172) MF(F,112,140,172)

template <int V>
__global__ __launch_bounds__(512) void k(unsigned long long* out, int iters) {
    const int g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);     // ping-pong group (8-wave launches)
    if (V >= 4 && g == 1) asm volatile("s_barrier" ::: "memory");
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (V == 0) asm volatile(BURST_SAME("a") ::: "memory");
        if constexpr (V == 1) asm volatile(BURST_SAME("v") ::: "memory");
        if constexpr (V == 2) asm volatile(BURST_GEMM("a") ::: "memory");
        if constexpr (V == 3) asm volatile(BURST_GEMM("v") ::: "memory");
        // ping-pong skeleton: [L: nothing] barrier [M: 16 MFMAs] barrier
        if constexpr (V == 4) asm volatile("s_barrier\n\ts_setprio 1\n\t" BURST_GEMM("v") "s_setprio 0\n\ts_barrier" ::: "memory");
        if constexpr (V == 5) asm volatile("s_barrier\n\ts_setprio 1\n\t" BURST_GEMM("a") "s_setprio 0\n\ts_barrier" ::: "memory");
        if constexpr (V == 6) asm volatile("s_barrier\n\t" BURST_GEMM("v") "s_barrier" ::: "memory");
        // 32 MFMAs per M interval (half the barriers)
        if constexpr (V == 7) asm volatile("s_barrier\n\ts_setprio 1\n\t" BURST_GEMM("v") BURST_GEMM("v") "s_setprio 0\n\ts_barrier" ::: "memory");
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (V >= 4 && g == 0) asm volatile("s_barrier" ::: "memory");
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int V> void run(const char* name, unsigned long long* d, int threads, int mf_per_iter) {
    const int iters = 20000;
    k<V><<<256, threads>>>(d, 10);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<V><<<256, threads>>>(d, iters);
    hipEventRecord(e1);
    if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) { printf("launch failed\n"); return; }
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c = 0;
    hipMemcpy(&c, d, 8, hipMemcpyDeviceToHost);
    const double flops = 256.0 * (threads / 64) * iters * (double)mf_per_iter * 32768.0;
    printf("%-78s %4d thr: %7.2f cycles per MFMA per wave, %8.1f per iteration; %6.0f TFLOP/s; clock %.2f GHz\n", name, threads,
           (double)c / (iters * (double)mf_per_iter), (double)c / iters, flops / (ms * 1e-3) / 1e12, (double)c / (ms * 1e-3) / 1e9);
}

int main() {
    unsigned long long* d;
    hipMalloc(&d, 64);
    for (int thr : {256, 512}) {
        run<0>("stream, acc AGPR, one operand pair", d, thr, 16);
        run<1>("stream, acc VGPR, one operand pair", d, thr, 16);
        run<2>("stream, acc AGPR, GEMM fragment pattern", d, thr, 16);
        run<3>("stream, acc VGPR, GEMM fragment pattern", d, thr, 16);
    }
    run<4>("ping-pong skeleton: barrier | setprio, 16 MFMA (acc VGPR) | barrier", d, 512, 16);
    run<5>("ping-pong skeleton: barrier | setprio, 16 MFMA (acc AGPR) | barrier", d, 512, 16);
    run<6>("ping-pong skeleton without setprio (acc VGPR)", d, 512, 16);
    run<7>("ping-pong skeleton, 32 MFMA per burst (acc VGPR)", d, 512, 32);
    return 0;
}
