// dev micro-benchmark: per-CU L2 -> LDS (global_load_lds) and L2 -> VGPR (global_load_dwordx4) streaming
// bandwidth for an L2-resident working set, one workgroup per CU.  hipcc --offload-arch=gfx950 ldbw.hip -o ldbw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

template <int NW>
__global__ __launch_bounds__(64 * NW) void k_glds(const char* src, int iters, size_t region, unsigned* sink, int ring) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* base = src + (size_t)blockIdx.x * region;
    constexpr int PER = 65536 / 1024 / NW;              // 1-KiB groups per wave for a 64 KiB stage
    for (int it = 0; it < iters; ++it) {
        const char* p = base + (size_t)(it % ring) * 65536;   // ring x 64 KiB per block
        char* dst = smem + (it & 1) * 65536;
#pragma unroll
        for (int g = 0; g < PER; ++g) {
            const int grp = wave * PER + g;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(p + grp * 1024 + lane * 16), (lds_ptr_t)(dst + grp * 1024), 16, 0, 0);
        }
        if (it & 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = *(unsigned*)(smem + 64);
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void k_vgpr(const char* src, int iters, size_t region, unsigned* sink, int ring) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char* base = src + (size_t)blockIdx.x * region;
    constexpr int PER = 65536 / 1024 / NW;
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        const char* p = base + (size_t)(it % ring) * 65536;
#pragma unroll
        for (int g = 0; g < PER; ++g) {
            const int grp = wave * PER + g;
            const u32x4 v = *(const u32x4*)(p + grp * 1024 + lane * 16);
            acc ^= v;
        }
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) sink[blockIdx.x] = 1;
}

int main() {
    const int nb = 256, iters = 4096;
    const size_t region = 8 * 65536;
    char* src; unsigned* sink;
    hipMalloc(&src, nb * region); hipMalloc(&sink, nb * 4);
    hipMemset(src, 1, nb * region);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto launch) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double bytes = (double)nb * iters * 65536;
        printf("%-28s %8.3f ms  %7.2f TB/s chip  %6.1f GB/s per CU  (%5.1f B/clk/CU @2.4GHz)\n", name, ms, bytes / ms / 1e9,
               bytes / ms / 1e6 / nb, bytes / ms / 1e6 / nb / 2.4);
    };
    hipFuncSetAttribute((const void*)k_glds<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void*)k_glds<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    for (int ring : {1, 2, 8}) {
        printf("-- ring %d x 64 KiB per block (%d MiB per XCD)\n", ring, ring * 2);
        run("glds 8 waves", [&] { hipLaunchKernelGGL(k_glds<8>, dim3(nb), dim3(512), 131072, 0, src, iters, region, sink, ring); });
        run("glds 16 waves", [&] { hipLaunchKernelGGL(k_glds<16>, dim3(nb), dim3(1024), 131072, 0, src, iters, region, sink, ring); });
        run("vgpr dwordx4 8 waves", [&] { hipLaunchKernelGGL(k_vgpr<8>, dim3(nb), dim3(512), 0, 0, src, iters, region, sink, ring); });
        run("vgpr dwordx4 16 waves", [&] { hipLaunchKernelGGL(k_vgpr<16>, dim3(nb), dim3(1024), 0, 0, src, iters, region, sink, ring); });
    }
    return 0;
}
