// ubench_ldsadd_exec.hip -- does a ds_add_f64 cost less when only some lanes are active?  (costvol_bwd: east hand-over / run-length
// schemes would issue the same number of LDS atomics with fewer active lanes each)
//   hipcc --offload-arch=gfx950 -O2 -o gpurun_ab/ubench_ldsadd_exec tools/ubench_ldsadd_exec.hip && gpurun_ab/ubench_ldsadd_exec
// 8 waves per CU (the backward kernel's occupancy), whole chip, every wave issues N ds_add_f64 into its own 4 KB box with
// lanes [0, nact) active (stride: every k-th lane); time from HIP events.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ __launch_bounds__(512) void k(int nact, int stride, int n, float* sinkp)
{
    __shared__ __attribute__((aligned(16))) double lds[8 * 512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 8 * 512; i += blockDim.x) lds[i] = 0.0;
    __syncthreads();
    const uint32_t a = (uint32_t)(uintptr_t)(lds + wave * 512) + (uint32_t)lane * 8u;
    const double v = 1.0;
    const bool on = (lane % stride) == 0 && (lane / stride) < nact;
    for (int i = 0; i < n; i += 8) {
        if (on) {
#pragma unroll
            for (int u = 0; u < 8; ++u) asm volatile("ds_add_f64 %0, %1 offset:%2" :: "v"(a), "v"(v), "n"(0) : "memory");
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (lds[threadIdx.x] == 12345.0) *sinkp = 1.0f;
}

int main()
{
    float* sink; hipMalloc(&sink, 4);
    const int n = 1 << 15;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int cases[][2] = {{64, 1}, {32, 1}, {32, 2}, {16, 1}, {16, 4}, {8, 1}, {8, 8}, {4, 16}, {2, 32}, {1, 1}};
    for (auto& c : cases) {
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, c[0], c[1], n, sink);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, c[0], c[1], n, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("ds_add_f64  %2d active lanes, every %2d-th: %7.3f ms  %6.3f ns per instruction per CU\n", c[0], c[1], ms, ms * 1e6 / ((double)n * 8));
    }
    return 0;
}
