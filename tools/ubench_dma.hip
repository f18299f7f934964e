// ubench_dma.hip -- vector-memory issue cost per CU for the access shapes of the staged cost-volume kernel.
//   hipcc --offload-arch=gfx950 -O2 -o gpurun_ab/ubench_dma tools/ubench_dma.hip && gpurun_ab/ubench_dma
// 256 workgroups x 768 threads (= 12 waves per CU, 3 per SIMD like the kernel).  Every wave runs REP rounds of 16 loads
// from an L2-resident 2 MiB window followed by s_waitcnt vmcnt(0).  Reported: ns per wave-instruction per CU and
// bytes/clock/CU at an assumed 2.1 GHz.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef int i32x4 __attribute__((ext_vector_type(4)));
#define REP 400

__device__ __forceinline__ i32x4 rsrc(const void* p, unsigned bytes)
{
    const unsigned long long a = (unsigned long long)p;
    i32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((unsigned)a);
    r.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    r.z = bytes; r.w = 0x00020000;
    return r;
}

// MODE 0: dword LDS-DMA, lane -> (column lane>>1, channel lane&1) [kernel's map]
// MODE 1: dword LDS-DMA, lane -> column (contiguous 256 B)
// MODE 2: dword load to VGPR, contiguous
// MODE 3: dwordx4 LDS-DMA, contiguous 1 KiB
// MODE 4: like 0 but only 10 of 64 lanes in range (the second half-row of a 37-column box)
// MODE 5: dwordx4 load to VGPR
template <int MODE>
__global__ __launch_bounds__(768) void k(float* src, float* sink, unsigned plane_bytes)
{
    __shared__ float lds[12][2048];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const i32x4 r = rsrc(src, 2u << 20);
    const unsigned ldsb = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)&lds[wave][0]);
    unsigned vo;
    if (MODE == 0) vo = (lane >> 1) * 4 + (lane & 1) * plane_bytes;
    else if (MODE == 4) vo = (lane < 10) ? (lane >> 1) * 4 + (lane & 1) * plane_bytes : 0x80000000u;
    else if (MODE == 3 || MODE == 5) vo = lane * 16;
    else vo = lane * 4;
    // modes 16/17: the loads of mode 1 plus 8 stores per round (ns are per LOAD instruction: 16 loads + 8 stores)
    float acc = 0;
    double v2 = lane; i32x4 v4 = {lane, lane, lane, lane};
    const unsigned vo2 = (lane & 31) * 4 + (lane >> 5) * 3072;   // two 128-byte segments, like the kernel's 32x2 patch
    if (MODE == 7) vo = lane * 8;
    if (MODE == 8 || MODE == 11) vo = lane * 16;
    unsigned so = (blockIdx.x * 12 + wave) * 3072 % (1u << 19);
    for (int it = 0; it < REP; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const unsigned s = (so + j * 3072) & ((1u << 19) - 1);
            if (MODE == 2) {
                float v;
                asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(v) : "v"(vo), "s"(r), "s"(s) : "memory");
                asm volatile("" :: "v"(v));
            } else if (MODE == 5) {
                i32x4 v;
                asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v) : "v"(vo), "s"(r), "s"(s) : "memory");
                asm volatile("" :: "v"(v));
            } else if (MODE == 6) {
                asm volatile("buffer_store_dword %0, %1, %2, %3 offen nt" :: "v"(acc), "v"(vo), "s"(r), "s"(s) : "memory");
            } else if (MODE == 7) {
                asm volatile("buffer_store_dwordx2 %0, %1, %2, %3 offen nt" :: "v"(v2), "v"(vo), "s"(r), "s"(s) : "memory");
            } else if (MODE == 8) {
                asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen nt" :: "v"(v4), "v"(vo), "s"(r), "s"(s) : "memory");
            } else if (MODE == 9) {
                asm volatile("buffer_store_dword %0, %1, %2, %3 offen nt" :: "v"(acc), "v"(vo2), "s"(r), "s"(s) : "memory");
            } else if (MODE == 10) {
                asm volatile("buffer_store_dword %0, %1, %2, %3 offen" :: "v"(acc), "v"(vo), "s"(r), "s"(s) : "memory");
            } else if (MODE == 11) {
                asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen" :: "v"(v4), "v"(vo), "s"(r), "s"(s) : "memory");
            } else if (MODE == 12) {
                asm volatile("buffer_store_dword %0, %1, %2, %3 offen sc0 sc1" :: "v"(acc), "v"(vo), "s"(r), "s"(s) : "memory");
            } else if (MODE == 13) {
                asm volatile("buffer_store_dword %0, %1, %2, %3 offen sc1" :: "v"(acc), "v"(vo), "s"(r), "s"(s) : "memory");
            } else if (MODE == 14) {
                asm volatile("buffer_store_dword %0, %1, %2, %3 offen sc0" :: "v"(acc), "v"(vo), "s"(r), "s"(s) : "memory");
            } else if (MODE == 15) {
                asm volatile("buffer_store_dword %0, %1, %2, %3 offen nt sc1" :: "v"(acc), "v"(vo), "s"(r), "s"(s) : "memory");
            } else if (MODE == 3) {
                asm volatile("s_add_u32 m0, %0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds"
                             :: "s"(ldsb), "v"(vo), "s"(r), "n"(0), "s"(s) : "memory", "m0", "scc");
            } else {
                asm volatile("s_add_u32 m0, %0, %3\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %4 offen lds"
                             :: "s"(ldsb), "v"(vo), "s"(r), "n"(0), "s"(s) : "memory", "m0", "scc");
            }
        }
        if (MODE == 16 || MODE == 17) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned s = ((so + j * 3072) & ((1u << 19) - 1)) + (1u << 20);
                asm volatile("buffer_store_dword %0, %1, %2, %3 offen nt" :: "v"(acc), "v"(vo2), "s"(r), "s"(s) : "memory");
            }
        }
        if (MODE != 17) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        so = (so + 49152) & ((1u << 19) - 1);
    }
    acc += lds[wave][lane];
    sink[blockIdx.x * 768 + threadIdx.x] = acc;
}

template <int MODE>
static void run(const char* name, float* src, float* sink, int bytes_per_instr)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(768), 0, 0, src, sink, 1u << 20);
    hipEventRecord(a);
    const int n = 5;
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(768), 0, 0, src, sink, 1u << 20);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double ns_per_instr_cu = ms * 1e6 / n / (REP * 16.0 * 12);
    printf("%-64s %7.2f ns/instr/CU  %6.1f B/clk/CU @2.1GHz\n", name, ns_per_instr_cu, bytes_per_instr / (ns_per_instr_cu * 2.1));
}

int main()
{
    setvbuf(stdout, NULL, _IONBF, 0);
    float *src, *sink;
    hipMalloc(&src, 4u << 20);
    hipMemset(src, 0, 4u << 20);
    hipMalloc(&sink, 256 * 768 * 4);
    run<0>("dword LDS-DMA, lane -> (col, channel) interleaved (kernel)", src, sink, 256);
    run<1>("dword LDS-DMA, contiguous 256 B", src, sink, 256);
    run<4>("dword LDS-DMA, 10 of 64 lanes in range", src, sink, 40);
    run<2>("dword load to VGPR, contiguous 256 B", src, sink, 256);
    run<3>("dwordx4 LDS-DMA, contiguous 1 KiB", src, sink, 1024);
    run<5>("dwordx4 load to VGPR, contiguous 1 KiB", src, sink, 1024);
    run<6>("dword store nt, contiguous 256 B (L2-resident window)", src, sink, 256);
    run<9>("dword store nt, 2 x 128 B segments", src, sink, 256);
    run<7>("dwordx2 store nt, contiguous 512 B", src, sink, 512);
    run<8>("dwordx4 store nt, contiguous 1 KiB", src, sink, 1024);
    run<16>("16 dword LDS-DMA (contiguous) + 8 dword stores per round, vmcnt(0)", src, sink, 256);
    run<17>("16 dword LDS-DMA (contiguous) + 8 dword stores per round, vmcnt(8)", src, sink, 256);
    run<10>("dword store default policy, contiguous 256 B", src, sink, 256);
    run<11>("dwordx4 store default policy, contiguous 1 KiB", src, sink, 1024);
    run<12>("dword store sc0 sc1", src, sink, 256);
    run<13>("dword store sc1", src, sink, 256);
    run<14>("dword store sc0", src, sink, 256);
    run<15>("dword store nt sc1", src, sink, 256);
    return 0;
}
