// Dev micro-benchmark (gfx950): (1) element mapping of ds_read_b64_tr_b16, (2) LDS cycles per wave-instruction of the
// transposed fragment read out of a ROW-MAJOR [64 rows][256 B] bf16 tile under three 16-byte-chunk swizzles, next to the
// ds_read_b128 row fragment read of the same tile.  Build: hipcc --offload-arch=gfx950 -O3 trread.hip -o trread
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4;

__device__ __forceinline__ int swz(int r, int mode) {
    if (mode == 0) return 0;                                  // none
    if (mode == 1) return r & 15;                             // plain XOR with the row
    return ((r & 3) << 2) | ((r >> 2) & 3);                   // row bits rotated: 4 consecutive rows -> 4 distinct 64-B quarters
}

__global__ void mapping(short* out) {
    __shared__ __attribute__((aligned(16))) short lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (short)i;
    __syncthreads();
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(lds + threadIdx.x * 4));
    *(s16x4*)(out + threadIdx.x * 4) = v;
}

template <int MODE, bool TR>
__global__ __launch_bounds__(512) void bench(uint64_t* cyc, int iters, uint32_t* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [64 rows][256 B] x 2
    for (int i = threadIdx.x; i < 32768 / 4; i += blockDim.x) ((uint32_t*)smem)[i] = i * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5, i16 = lane & 15;
    uint32_t acc = 0;
    uint64_t t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {          // 16-key groups of the tile
#pragma unroll
            for (int db = 0; db < 4; ++db) {      // 32-wide d blocks
                if constexpr (TR) {
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        const int r = kb * 16 + half * 8 + 4 * hi + (i16 >> 2);
                        const int chunk = db * 4 + (l31 >> 4) * 2 + ((i16 & 3) >> 1);
                        const int off = r * 256 + ((chunk ^ swz(r, MODE)) << 4) + 8 * (i16 & 1);
                        s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(smem + off));
                        acc ^= (uint32_t)(uint16_t)v[0] + (uint32_t)(uint16_t)v[3];
                    }
                } else {
                    // row fragment: lane (row l31, hi) reads 16 B at d chunk (db*2 + hi) ... of 32-row block kb&1
                    const int r = (kb & 1) * 32 + l31;
                    const int chunk = db * 4 + (kb >> 1) * 2 + hi;
                    const int off = r * 256 + ((chunk ^ swz(r, MODE)) << 4);
                    u32x4 v = *(const u32x4*)(smem + off);
                    acc ^= v[0] + v[3];
                }
            }
        }
    }
    uint64_t t1 = clock64();
    if (lane == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE, bool TR>
static void run(const char* name, int nread) {
    uint64_t* cyc; uint32_t* sink;
    hipMalloc(&cyc, 8 * 8 * sizeof(uint64_t)); hipMalloc(&sink, 4);
    const int iters = 2000;
    hipLaunchKernelGGL((bench<MODE, TR>), dim3(1), dim3(512), 32768, 0, cyc, iters, sink);
    hipLaunchKernelGGL((bench<MODE, TR>), dim3(1), dim3(512), 32768, 0, cyc, iters, sink);
    hipDeviceSynchronize();
    uint64_t h[8]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double mx = 0; for (int i = 0; i < 8; ++i) mx = h[i] > mx ? (double)h[i] : mx;
    // clock64 = s_memtime ticks at 100 MHz on gfx9?  report raw ticks per wave-instruction; ratios between rows are what matter
    printf("%-34s ticks/wave-instr %.4f (8 waves on one CU; %d reads per iteration)\n", name, mx / ((double)iters * nread), nread);
}

int main() {
    short* out; hipMalloc(&out, 256 * sizeof(short));
    hipLaunchKernelGGL(mapping, dim3(1), dim3(64), 0, 0, out);
    short h[256]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    printf("ds_read_b64_tr_b16 with address = lds + lane*8 (values = element index):\n");
    for (int l = 0; l < 64; ++l) { printf("lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]); }
    run<0, true>("tr_b16  no swizzle", 32);
    run<1, true>("tr_b16  chunk ^= row&15", 32);
    run<2, true>("tr_b16  chunk ^= rot(row)", 32);
    run<0, false>("b128    no swizzle", 16);
    run<1, false>("b128    chunk ^= row&15", 16);
    run<2, false>("b128    chunk ^= rot(row)", 16);
    return 0;
}
