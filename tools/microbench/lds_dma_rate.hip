// Micro-benchmark (GPU box): bytes per clock and CU of the LDS-DMA fill the attention passes use -- 1-KB contiguous pieces
// (64 lanes x 16 B) issued by the 8 waves of a workgroup, one workgroup per CU -- against 4-byte-per-lane LDS-DMA and against register
// staging (global_load_dwordx4 + ds_write_b128).  Each workgroup streams its own 358-KB region (L2-resident after the first pass)
// STAGE bytes at a time: issue, drain, barrier -- the pattern of a single-stage pass.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/lds_dma_rate.hip -o gpurun_out/lds_dma_rate && gpurun_out/lds_dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int REGION = 358 * 1024;

template <int MODE, int WAVES>
__global__ __launch_bounds__(512) void fill_kernel(const char* __restrict__ src, int stage_bytes, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const char* base = src + (size_t)blockIdx.x * REGION;
    const int pieces = stage_bytes / 1024;
    float acc = 0.f;
    int off = 0;
    for (int it = 0; it < iters; ++it) {
        if (wave < WAVES) {
            for (int p = wave; p < pieces; p += WAVES) {
                const char* g = base + off + p * 1024;
                char* l = smem + p * 1024;
                if (MODE == 0) {
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + lane * 16),
                                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
                } else if (MODE == 1) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + q * 256 + lane * 4),
                                                         (__attribute__((address_space(3))) void*)(l + q * 256), 4, 0, 0);
                } else {
                    const float4 v = *reinterpret_cast<const float4*>(g + lane * 16);
                    *reinterpret_cast<float4*>(l + lane * 16) = v;
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        acc += reinterpret_cast<const float*>(smem)[(tid * 4 + it) & 1023];
        asm volatile("s_barrier" ::: "memory");
        off += stage_bytes;
        if (off + stage_bytes > REGION) off = 0;
    }
    if (acc == 12345.678f) sink[0] = acc;
}

template <int MODE, int WAVES>
static int run(const char* name, const char* src, float* sink, int grid, int stage_bytes, int iters) {
    auto k = fill_kernel<MODE, WAVES>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t lds = (size_t)(stage_bytes > 100 * 1024 ? stage_bytes : 100 * 1024);   // >= 100 KB: one workgroup per CU, as in the passes
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, 0, src, stage_bytes, iters, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, 0, src, stage_bytes, iters, sink);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)grid * iters * stage_bytes;
    printf("%-34s grid %3d waves %d stage %3d KB: %7.1f us  %6.2f TB/s  %5.1f GB/s per CU  (%4.1f B/clk/CU at 2.1 GHz)\n", name, grid, WAVES,
           stage_bytes / 1024, ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e6 / grid, bytes / ms / 1e6 / grid / 2.1);
    return 0;
}

int main() {
    const int maxgrid = 256;
    char* src; float* sink;
    CK(hipMalloc(&src, (size_t)maxgrid * REGION));
    CK(hipMemset(src, 1, (size_t)maxgrid * REGION));
    CK(hipMalloc(&sink, 64));
    for (int grid : {160, 256}) {
        for (int stage : {28 * 1024, 56 * 1024, 84 * 1024}) {
            if (run<0, 8>("LDS-DMA 16 B/lane", src, sink, grid, stage, 200)) return 1;
            if (run<0, 4>("LDS-DMA 16 B/lane", src, sink, grid, stage, 200)) return 1;
            if (run<1, 8>("LDS-DMA 4 B/lane (4 per KB)", src, sink, grid, stage, 200)) return 1;
            if (run<2, 8>("registers: dwordx4 + ds_write_b128", src, sink, grid, stage, 200)) return 1;
        }
    }
    return 0;
}
