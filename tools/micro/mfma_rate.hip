// dev micro-benchmark: issue rate of v_mfma_f32_16x16x32_bf16 from ONE wave per SIMD as a function of where its operands live
// (VGPR bank alignment of srcA / srcB, accumulators in VGPRs or AGPRs).  hipcc --offload-arch=gfx950 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define STR2(x) #x
#define STR(x) STR2(x)
// 16 independent MFMAs (16 accumulators), A at v[A0:A0+3], B at v[B0:B0+3] (and a second B at B1), accumulators ACC[4i:4i+3]
#define MF(acc, i, A0, B0) "v_mfma_f32_16x16x32_bf16 " acc "[" STR(i) ":" STR(i+3) "], v[" STR(A0) ":" STR(A0+3) "], v[" STR(B0) ":" STR(B0+3) "], " acc "[" STR(i) ":" STR(i+3) "]\n\t"

#define MF32(i) "v_mfma_f32_32x32x16_bf16 a[" STR(i) ":" STR(i+15) "], v[64:67], v[68:71], a[" STR(i) ":" STR(i+15) "]\n\t"
template <int VARIANT>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, int iters) {
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (VARIANT == 0)        // acc in AGPR, A v[64:67], B v[68:71]  (both tuples start at bank 0)
            asm volatile(MF("a",0,64,68) MF("a",4,64,68) MF("a",8,64,68) MF("a",12,64,68) MF("a",16,64,68) MF("a",20,64,68) MF("a",24,64,68) MF("a",28,64,68)
                         MF("a",32,64,68) MF("a",36,64,68) MF("a",40,64,68) MF("a",44,64,68) MF("a",48,64,68) MF("a",52,64,68) MF("a",56,64,68) MF("a",60,64,68)
                         ::: "memory");
        else if constexpr (VARIANT == 5)   // 8 independent 32x32x16 MFMAs (same FLOPs as the 16 16x16x32 above)
            asm volatile(MF32(0) MF32(16) MF32(32) MF32(48) MF32(64) MF32(80) MF32(96) MF32(112) ::: "memory");
        else if constexpr (VARIANT == 1)   // B shifted by two
            asm volatile(MF("a",0,64,70) MF("a",4,64,70) MF("a",8,64,70) MF("a",12,64,70) MF("a",16,64,70) MF("a",20,64,70) MF("a",24,64,70) MF("a",28,64,70)
                         MF("a",32,64,70) MF("a",36,64,70) MF("a",40,64,70) MF("a",44,64,70) MF("a",48,64,70) MF("a",52,64,70) MF("a",56,64,70) MF("a",60,64,70)
                         ::: "memory");
        else if constexpr (VARIANT == 2)   // B shifted by two
            asm volatile(MF("a",0,64,70) MF("a",4,64,70) MF("a",8,64,70) MF("a",12,64,70) MF("a",16,64,70) MF("a",20,64,70) MF("a",24,64,70) MF("a",28,64,70)
                         MF("a",32,64,70) MF("a",36,64,70) MF("a",40,64,70) MF("a",44,64,70) MF("a",48,64,70) MF("a",52,64,70) MF("a",56,64,70) MF("a",60,64,70)
                         ::: "memory");
        else if constexpr (VARIANT == 3)   // acc in VGPRs v[0:63], A v[64:67], B v[68:71]
            asm volatile(MF("v",0,64,68) MF("v",4,64,68) MF("v",8,64,68) MF("v",12,64,68) MF("v",16,64,68) MF("v",20,64,68) MF("v",24,64,68) MF("v",28,64,68)
                         MF("v",32,64,68) MF("v",36,64,68) MF("v",40,64,68) MF("v",44,64,68) MF("v",48,64,68) MF("v",52,64,68) MF("v",56,64,68) MF("v",60,64,68)
                         ::: "memory");
        else                                // acc in VGPRs, B shifted by two
            asm volatile(MF("v",0,64,70) MF("v",4,64,70) MF("v",8,64,70) MF("v",12,64,70) MF("v",16,64,70) MF("v",20,64,70) MF("v",24,64,70) MF("v",28,64,70)
                         MF("v",32,64,70) MF("v",36,64,70) MF("v",40,64,70) MF("v",44,64,70) MF("v",48,64,70) MF("v",52,64,70) MF("v",56,64,70) MF("v",60,64,70)
                         ::: "memory");
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int V> void run(const char* name, unsigned long long* d, int waves_per_simd) {
    const int iters = 100000;
    k<V><<<256, 64 * 4 * waves_per_simd>>>(d, 10);
    hipDeviceSynchronize();
    hipMemset(d, 0, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<V><<<256, 64 * 4 * waves_per_simd>>>(d, iters);
    hipEventRecord(e1);
    if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) { printf("launch failed\n"); return; }
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 256.0 * 4 * waves_per_simd * iters * 16.0 * 16384.0;
    printf("   wall %.3f ms -> %.0f TFLOP/s\n", ms, flops / (ms * 1e-3) / 1e12);
    unsigned long long c = 0;
    hipMemcpy(&c, d, 8, hipMemcpyDeviceToHost);
    printf("%-64s waves/SIMD %d: %.2f ticks per MFMA per wave (%.2f per SIMD); s_memtime rate %.3f GHz\n", name, waves_per_simd,
           (double)c / (iters * 16.0), (double)c / (iters * 16.0 * waves_per_simd), (double)c / (ms * 1e-3) / 1e9);
}

int main(int argc, char** argv) {
    unsigned long long* d;
    hipMalloc(&d, 64);
    if (argc > 2) {      // power mode: mfma_rate <variant 0|5> <seconds>: sustained loop for rocm-smi sampling
        const int v = atoi(argv[1]);
        const double secs = atof(argv[2]);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        double total_ms = 0; long launches = 0;
        while (total_ms < secs * 1e3) {
            hipEventRecord(e0);
            for (int i = 0; i < 20; ++i) { if (v == 5) k<5><<<256, 1024>>>(d, 20000); else k<0><<<256, 1024>>>(d, 20000); }
            hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1); total_ms += ms; launches += 20;
        }
        printf("variant %d: %.0f TFLOP/s sustained over %.1f s\n", v, 256.0 * 16 * 20000 * 16 * 16384.0 * launches / (total_ms * 1e-3) / 1e12, total_ms * 1e-3);
        return 0;
    }
    for (int w = 1; w <= 4; w *= 2) {
        run<0>("acc AGPR, A v[64:67] B v[68:71] (same bank phase)", d, w);
        run<5>("32x32x16: 8 MFMAs per block (flops reported as if 16)", d, w);
        if (0) run<1>("acc AGPR, A v[64:67] B v[70:73] (B phase +2, again)", d, w);
        run<2>("acc AGPR, A v[64:67] B v[70:73] (B phase +2)", d, w);
        if (0) run<3>("acc VGPR, A v[64:67] B v[68:71]", d, w);
        if (0) run<4>("acc VGPR, A v[64:67] B v[70:73]", d, w);
    }
    return 0;
}
