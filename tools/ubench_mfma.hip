// ubench_mfma.hip -- issue cost of v_mfma_f32_32x32x2_f32 (the regularisers' exact-float32 matrix instruction):
// one wave per SIMD, REP x 16 instructions, as a dependent chain on ONE accumulator (what a wave with one cout tile
// runs) or round-robin over 2 / 4 independent accumulators.  Also 16x16x4.
//   hipcc --offload-arch=gfx950 -O2 -o gpurun_ab/ubench_mfma tools/ubench_mfma.hip && gpurun_ab/ubench_mfma
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define REP 256

template <int NACC, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k32(float* out, unsigned long long* clk, float a, float b)
{
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = (float)threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < REP; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i % NACC], 0, 0, 0);
    }
    float s = 0;
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, unsigned long long* clk, float a, float b)
{
    f32x4 acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 4; ++r) acc[n][r] = (float)threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < REP; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i % NACC], 0, 0, 0);
    }
    float s = 0;
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 4; ++r) s += acc[n][r];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <typename F> static void run(const char* name, F launch, int threads)
{
    float* out; unsigned long long* clk;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&clk, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(out, clk);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) launch(out, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    printf("%-52s %7.1f s_memtime ticks / instr   %7.2f ns / instr (wall, 10 launches)\n", name, (double)c / (REP * 16.0), ms * 1e6 / 10 / (REP * 16.0));
    hipFree(out); hipFree(clk);
}

int main()
{
    setvbuf(stdout, NULL, _IONBF, 0);
#define L32(NACC, WAVES) [](float* o, unsigned long long* c) { hipLaunchKernelGGL((k32<NACC, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, o, c, 1.0f, 2.0f); }
#define L16(NACC) [](float* o, unsigned long long* c) { hipLaunchKernelGGL((k16<NACC>), dim3(256), dim3(256), 0, 0, o, c, 1.0f, 2.0f); }
    run("32x32x2 f32, 1 accumulator (dependent), 4 waves/CU", L32(1, 4), 256);
    run("32x32x2 f32, 2 accumulators, 4 waves/CU", L32(2, 4), 256);
    run("32x32x2 f32, 4 accumulators, 4 waves/CU", L32(4, 4), 256);
    run("32x32x2 f32, 1 accumulator, 8 waves/CU (2 per SIMD)", L32(1, 8), 512);
    run("16x16x4 f32, 1 accumulator (dependent), 4 waves/CU", L16(1), 256);
    run("16x16x4 f32, 4 accumulators, 4 waves/CU", L16(4), 256);
    return 0;
}
