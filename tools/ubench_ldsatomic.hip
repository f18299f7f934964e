// ubench_ldsatomic.hip -- issue rate of LDS float atomics on gfx950 (does a wave-private ds_add_f32 scatter pay?)
//   hipcc --offload-arch=gfx950 -O2 -o gpurun_ab/ubench_ldsatomic tools/ubench_ldsatomic.hip && gpurun_ab/ubench_ldsatomic
// Each wave issues N instructions of one kind into its own 4 KB of LDS; cycles per instruction from s_memtime, for 1, 4 and
// 12 waves per CU.  Address patterns: linear (lane*4), two rows of 32 at a pitch of 96 dwords (the backward's box),
// pairs of lanes on one cell, all lanes on one cell.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

enum { K_ADD_F32, K_ADD_RTN_F32, K_ADD_U32, K_WRITE, K_READ, K_WRXCHG, K_PK_ADD_F16, K_ADD_F64, K_ADD_U64, K_MAX_F32, K_WRXCHG64 };

template <int KIND>
__device__ __forceinline__ void op(uint32_t a, float v, float& sink)
{
    if (KIND == K_ADD_F32) asm volatile("ds_add_f32 %0, %1" :: "v"(a), "v"(v) : "memory");
    if (KIND == K_ADD_U32) asm volatile("ds_add_u32 %0, %1" :: "v"(a), "v"(v) : "memory");
    if (KIND == K_WRITE) asm volatile("ds_write_b32 %0, %1" :: "v"(a), "v"(v) : "memory");
    if (KIND == K_ADD_RTN_F32) asm volatile("ds_add_rtn_f32 %0, %1, %2" : "=v"(sink) : "v"(a), "v"(v) : "memory");
    if (KIND == K_READ) asm volatile("ds_read_b32 %0, %1" : "=v"(sink) : "v"(a) : "memory");
    if (KIND == K_WRXCHG) asm volatile("ds_wrxchg_rtn_b32 %0, %1, %2" : "=v"(sink) : "v"(a), "v"(v) : "memory");
    if (KIND == K_PK_ADD_F16) asm volatile("ds_pk_add_f16 %0, %1" :: "v"(a), "v"(v) : "memory");
    if (KIND == K_ADD_F64) { double d = v; asm volatile("ds_add_f64 %0, %1" :: "v"(a), "v"(d) : "memory"); }
    if (KIND == K_ADD_U64) { unsigned long long d = 3; asm volatile("ds_add_u64 %0, %1" :: "v"(a), "v"(d) : "memory"); }
    if (KIND == K_MAX_F32) asm volatile("ds_max_f32 %0, %1" :: "v"(a), "v"(v) : "memory");
    if (KIND == K_WRXCHG64) { unsigned long long d = 0, r; asm volatile("ds_wrxchg_rtn_b64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(d) : "memory"); sink += (float)r; }
}

template <int KIND>
__global__ void k(int pattern, int n, unsigned long long* out, float* sinkp)
{
    __shared__ __attribute__((aligned(8))) float lds[16 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16 * 1024; i += blockDim.x) lds[i] = 0.0f;
    __syncthreads();
    uint32_t cell = lane;
    if (pattern == 1) cell = (lane & 31) + (lane >> 5) * 96;
    if (pattern == 2) cell = lane >> 1;
    if (pattern == 3) cell = 0;
    if (pattern == 4) cell = (lane & 31) + (lane >> 5) * 64;
    if (pattern == 5) cell = lane * 2;
    if (pattern == 6) cell = (lane * 37) & 511;          // pseudo-random scatter
    const uint32_t a = (uint32_t)(uintptr_t)(lds + wave * 1024) + cell * ((KIND == K_ADD_F64 || KIND == K_ADD_U64 || KIND == K_WRXCHG64) ? 8 : 4);
    float sink = 0.0f, v = 1.0f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; i += 8) {
        op<KIND>(a, v, sink); op<KIND>(a, v, sink); op<KIND>(a, v, sink); op<KIND>(a, v, sink);
        op<KIND>(a, v, sink); op<KIND>(a, v, sink); op<KIND>(a, v, sink); op<KIND>(a, v, sink);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
    if (sink == 12345.0f) *sinkp = sink + lds[threadIdx.x];
}

template <int KIND>
static void run(const char* name)
{
    unsigned long long* out; float* sink;
    hipMalloc(&out, 1024 * 16 * 8); hipMalloc(&sink, 4);
    const int n = 4096;
    const char* pats[] = {"linear", "2x32 pitch 96", "lane pairs", "one cell", "2x32 pitch 64", "stride 2", "scatter"};
    for (int pat = 0; pat < 7; ++pat) {
        printf("%-18s %-14s", name, pats[pat]);
        for (int waves : {1, 4, 12}) {
            hipMemset(out, 0, 1024 * 16 * 8);
            hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(64 * waves), 0, 0, pat, n, out, sink);
            hipDeviceSynchronize();
            unsigned long long h[16];
            hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
            double worst = 0;
            for (int w = 0; w < waves; ++w) worst = h[w] > worst ? (double)h[w] : worst;
            // s_memtime counts at 100 MHz on gfx9: report raw ticks per instruction and per-CU instruction rate
            printf("  %2d waves: %7.3f ticks/instr/wave (%7.3f per CU instr)", waves, worst / n, worst / n / waves);
        }
        printf("\n");
    }
}

int main()
{
    setvbuf(stdout, NULL, _IONBF, 0);
    printf("ticks = __builtin_readcyclecounter() units (s_memtime, 100 MHz: 1 tick = 10 ns ~ 20-24 shader clocks)\n");
    run<K_ADD_F64>("ds_add_f64");
    run<K_ADD_U64>("ds_add_u64");
    run<K_MAX_F32>("ds_max_f32");
    run<K_WRXCHG64>("ds_wrxchg_rtn_b64");
    run<K_WRITE>("ds_write_b32");
    run<K_ADD_U32>("ds_add_u32");
    run<K_ADD_F32>("ds_add_f32");
    return 0;
}
