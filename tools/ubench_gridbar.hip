// ubench_gridbar.hip -- what does a grid-wide barrier inside one kernel cost on MI355X, against a dependent launch?
//   hipcc --offload-arch=gfx950 -O2 -o gpurun_ab/ubench_gridbar tools/ubench_gridbar.hip && gpurun_ab/ubench_gridbar
// Persistent grid of G workgroups x 256 threads; each round every workgroup writes a slice of a buffer another
// workgroup (on another XCD) reads in the next round, then crosses a barrier = counter in global memory, agent-scope
// release before the increment, agent-scope acquire after the spin.  Compared with the same rounds as separate launches.
#include <hip/hip_runtime.h>
#include <stdio.h>

__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned target)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE);                       // agent scope by default for global atomics in HIP
        while (__atomic_load_n(ctr, __ATOMIC_ACQUIRE) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

__device__ __forceinline__ void grid_barrier_fence(unsigned* ctr, unsigned target)
{
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(ctr, 1u);
        while (atomicAdd(ctr, 0u) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    __threadfence();
}

template <int MODE>
__global__ __launch_bounds__(256) void persistent(float* a, float* b, unsigned* ctr, int rounds, int n_per_wg, int* bad)
{
    const int G = gridDim.x;
    float* src = a; float* dst = b;
    for (int r = 0; r < rounds; ++r) {
        // read the slice the workgroup "G/2+3 away" wrote last round, write mine
        const int from = (blockIdx.x + G / 2 + 3) % G;
        for (int i = threadIdx.x; i < n_per_wg; i += 256) {
            const float v = src[(size_t)from * n_per_wg + i];
            if (r > 0 && v != (float)(r - 1 + from)) atomicAdd(bad, 1);
            dst[(size_t)blockIdx.x * n_per_wg + i] = (float)(r + blockIdx.x);
        }
        if (MODE == 0) grid_barrier(ctr, (unsigned)(r + 1) * G);
        else grid_barrier_fence(ctr, (unsigned)(r + 1) * G);
        float* t = src; src = dst; dst = t;
    }
}

__global__ __launch_bounds__(256) void one_round(const float* src, float* dst, int r, int n_per_wg, int* bad)
{
    const int G = gridDim.x;
    const int from = (blockIdx.x + G / 2 + 3) % G;
    for (int i = threadIdx.x; i < n_per_wg; i += 256) {
        const float v = src[(size_t)from * n_per_wg + i];
        if (r > 0 && v != (float)(r - 1 + from)) atomicAdd(bad, 1);
        dst[(size_t)blockIdx.x * n_per_wg + i] = (float)(r + blockIdx.x);
    }
}

int main()
{
    setvbuf(stdout, NULL, _IONBF, 0);
    const int rounds = 200;
    for (int G : {64, 256, 512, 1024}) for (int n_per_wg : {256, 4096}) {
        float *a, *b; unsigned* ctr; int* bad;
        hipMalloc(&a, (size_t)G * n_per_wg * 4); hipMalloc(&b, (size_t)G * n_per_wg * 4); hipMalloc(&ctr, 4); hipMalloc(&bad, 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float ms[3];
        for (int mode = 0; mode < 3; ++mode) {
            hipMemset(ctr, 0, 4); hipMemset(bad, 0, 4); hipMemset(a, 0, (size_t)G * n_per_wg * 4);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(persistent<0>, dim3(G), dim3(256), 0, 0, a, b, ctr, rounds, n_per_wg, bad);
            else if (mode == 1) hipLaunchKernelGGL(persistent<1>, dim3(G), dim3(256), 0, 0, a, b, ctr, rounds, n_per_wg, bad);
            else for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(one_round, dim3(G), dim3(256), 0, 0, (r & 1) ? b : a, (r & 1) ? a : b, r, n_per_wg, bad);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms[mode], e0, e1);
            int hb = 0; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
            if (hb) printf("  mode %d: %d stale reads!\n", mode, hb);
        }
        printf("G=%4d wg, %5d floats/wg/round: barrier(acq/rel atomics) %6.2f us/round   barrier(threadfence) %6.2f us/round   launches %6.2f us/round\n",
               G, n_per_wg, ms[0] * 1e3 / rounds, ms[1] * 1e3 / rounds, ms[2] * 1e3 / rounds);
        hipFree(a); hipFree(b); hipFree(ctr); hipFree(bad);
    }
    return 0;
}
