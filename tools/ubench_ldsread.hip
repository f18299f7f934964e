// ubench_ldsread.hip -- LDS read throughput for the tap-fetch patterns of the staged cost-volume kernel on gfx950:
// does a ds_read_b64 at a 4-byte (not 8-byte) aligned address keep its 2-cycle rate?
//   hipcc --offload-arch=gfx950 -O2 -o gpurun_ab/ubench_ldsread tools/ubench_ldsread.hip && gpurun_ab/ubench_ldsread
// Every wave issues N reads into its own LDS tile; 8 waves per CU (2 per SIMD, the staged kernel's occupancy), whole chip;
// time from HIP events -> LDS clocks per wave-instruction per CU at the measured sclk-free unit "ns per instruction per CU".
// Address patterns (dword index per lane): "tap" = lane + 3*(lane>>5)*44 (two pixel rows, unit stride: neighbouring lanes read
// overlapping windows), "tap odd" = the same + 1, "even" = 2*lane (8-byte aligned, no overlap).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

enum { R2_B32, R_B64, R_B32, R2_B64, R_B128 };

template <int KIND>
__global__ __launch_bounds__(512) void k(int pattern, int n, float* sinkp)
{
    __shared__ __attribute__((aligned(16))) float lds[8 * 2048];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 8 * 2048; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    uint32_t cell = lane + 3 * (lane >> 5) * 44;
    if (pattern == 1) cell += 1;
    if (pattern == 2) cell = 2 * lane;
    if (pattern == 3) cell = 4 * lane;
    const uint32_t a = (uint32_t)(uintptr_t)(lds + wave * 2048) + cell * 4;
    float acc = 0.0f;
    for (int i = 0; i < n; i += 8) {
        float2 v[8]; float4 q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == R2_B32) asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v[u]) : "v"(a), "n"(0), "n"(44) : "memory");
            if (KIND == R_B64) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[u]) : "v"(a), "n"(0) : "memory");
            if (KIND == R_B32) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v[u].x) : "v"(a), "n"(0) : "memory");
            if (KIND == R2_B64) asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(q[u]) : "v"(a), "n"(0), "n"(22) : "memory");
            if (KIND == R_B128) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[u]) : "v"(a), "n"(0) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += (KIND == R2_B64 || KIND == R_B128) ? q[u].x + q[u].w : (KIND == R_B32 ? v[u].x : v[u].x + v[u].y);
    }
    if (acc == 12345.0f) *sinkp = acc;
}

template <int KIND>
static void run(const char* name, int bytes)
{
    float* sink; hipMalloc(&sink, 4);
    const int n = 1 << 15;
    const char* pats[] = {"tap (unit stride, 4-B aligned)", "tap odd", "even (8-B aligned)", "16-B aligned"};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pat = 0; pat < 4; ++pat) {
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, pat, n, sink);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, pat, n, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipError_t err = hipGetLastError();
        const double ns_per = ms * 1e6 / ((double)n * 8);        // per wave-instruction per CU (8 waves per CU, one workgroup per CU)
        printf("%-14s %-32s %8.3f ms  %6.3f ns per instr per CU = %5.1f B/ns per CU  %s\n", name, pats[pat], ms, ns_per, 64.0 * bytes / ns_per,
               err == hipSuccess ? "" : hipGetErrorString(err));
    }
}

int main()
{
    run<R_B32>("ds_read_b32", 4);
    run<R2_B32>("ds_read2_b32", 8);
    run<R_B64>("ds_read_b64", 8);
    run<R2_B64>("ds_read2_b64", 16);
    run<R_B128>("ds_read_b128", 16);
    return 0;
}
