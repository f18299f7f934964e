// ubench_store.hip -- what HBM write rate does the cost-volume kernel's store PATTERN reach, with no compute at all?
//   hipcc --offload-arch=gfx950 -O2 -o gpurun_ab/ubench_store tools/ubench_store.hip && gpurun_ab/ubench_store
// Volume (C=32, D=64, H=384, W=768) float32 = 2.416 GB, written once per launch in the order the staged kernel writes
// it: a wave owns TX x TY pixels and DP planes, loops over channel pairs, 2*DP stores of one dword per lane per pair.
// Variants: tile shape, store width (1/2/4 pixels per lane), cache policy, workgroup order.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ void st1(float v, i32x4 r, int vo, int so, int aux) __asm("llvm.amdgcn.raw.buffer.store.f32");
__device__ void st2(f32x2 v, i32x4 r, int vo, int so, int aux) __asm("llvm.amdgcn.raw.buffer.store.v2f32");
__device__ void st4(f32x4 v, i32x4 r, int vo, int so, int aux) __asm("llvm.amdgcn.raw.buffer.store.v4f32");

constexpr int C = 32, D = 64, H = 384, W = 768, DP = 4;

__device__ __forceinline__ unsigned xcd_remap(unsigned i, unsigned n)
{
    const unsigned nx = 8, q = n / nx, r = n % nx, x = i % nx, j = i / nx;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
}

// PPL = pixels per lane (store width), wave covers (TXL*PPL) x TY pixels with TXL*TY = 64; workgroup = WV waves stacked in
// y (WGX = 1) or side by side in x (WGX = 1 means stacked).  ORDER 0: x tile fastest, then plane chunk, then y; 1: plane
// chunk fastest; 2: no xcd remap.
template <int PPL, int TXL, int TY, int WGX, int AUX, int ORDER>
__global__ __launch_bounds__(256) void store_kernel(float* out, float val)
{
    constexpr int TX = TXL * PPL;
    constexpr int WX = WGX ? 4 : 1, WY = WGX ? 1 : 4;
    constexpr int xt = W / (TX * WX), yt = H / (TY * WY), dct = D / DP;
    unsigned L = ORDER == 2 ? blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
    int xtile, dchunk, ytile;
    if (ORDER == 1) { dchunk = L % dct; L /= dct; xtile = L % xt; L /= xt; ytile = L; }
    else            { xtile = L % xt; L /= xt; dchunk = L % dct; L /= dct; ytile = L; }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = (xtile * WX + (WGX ? wave : 0)) * TX + (lane % TXL) * PPL;
    const int y = (ytile * WY + (WGX ? 0 : wave)) * TY + lane / TXL;
    const size_t HW = (size_t)H * W, ostride = (size_t)D * HW;
    unsigned ovo[DP];
#pragma unroll
    for (int pl = 0; pl < DP; ++pl) ovo[pl] = (unsigned)(((dchunk * DP + pl) * HW + (size_t)y * W + x) * 4);
    for (int c = 0; c < C; c += 2) {
        const unsigned long long a = (unsigned long long)(out + c * ostride);
        i32x4 r;
        r.x = __builtin_amdgcn_readfirstlane((unsigned)a);
        r.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        r.z = (int)(2 * ostride * 4);
        r.w = 0x00020000;
#pragma unroll
        for (int pl = 0; pl < DP; ++pl) {
            const float v = val + c + pl;
            if (PPL == 1) { st1(v, r, (int)ovo[pl], 0, AUX); st1(v + 1, r, (int)ovo[pl], (int)(ostride * 4), AUX); }
            if (PPL == 2) { f32x2 q = {v, v}; st2(q, r, (int)ovo[pl], 0, AUX); st2(q + 1.0f, r, (int)ovo[pl], (int)(ostride * 4), AUX); }
            if (PPL == 4) { f32x4 q = {v, v, v, v}; st4(q, r, (int)ovo[pl], 0, AUX); st4(q + 1.0f, r, (int)ovo[pl], (int)(ostride * 4), AUX); }
        }
    }
}

template <int PPL, int TXL, int TY, int WGX, int AUX, int ORDER>
static void run(const char* name, float* out)
{
    constexpr int TX = TXL * PPL;
    constexpr int WX = WGX ? 4 : 1, WY = WGX ? 1 : 4;
    const int nb = (W / (TX * WX)) * (H / (TY * WY)) * (D / DP);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((store_kernel<PPL, TXL, TY, WGX, AUX, ORDER>), dim3(nb), dim3(256), 0, 0, out, 1.0f);
    hipEventRecord(a);
    const int n = 50;
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL((store_kernel<PPL, TXL, TY, WGX, AUX, ORDER>), dim3(nb), dim3(256), 0, 0, out, 1.0f);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    ms /= n;
    printf("%-58s %8.3f ms  %7.0f GB/s\n", name, ms, (double)C * D * H * W * 4 / ms / 1e6);
}

// sustained mode: ubench_store <variant 0..4> <seconds> -- loops one variant so that power can be sampled beside it (tools/power_probe.sh)
template <int PPL, int TXL, int TY, int WGX, int AUX, int ORDER>
static void sustain(const char* name, float* out, double seconds)
{
    constexpr int TX = TXL * PPL;
    constexpr int WX = WGX ? 4 : 1, WY = WGX ? 1 : 4;
    const int nb = (W / (TX * WX)) * (H / (TY * WY)) * (D / DP);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int n = (int)(seconds / 0.42e-3);
    hipEventRecord(a);
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL((store_kernel<PPL, TXL, TY, WGX, AUX, ORDER>), dim3(nb), dim3(256), 0, 0, out, 1.0f);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    printf("%-58s %8.3f ms per launch over %d launches\n", name, ms / n, n);
}

int main(int argc, char** argv)
{
    setvbuf(stdout, NULL, _IONBF, 0);
    float* out;
    hipMalloc(&out, (size_t)C * D * H * W * 4);
    hipMemset(out, 0, (size_t)C * D * H * W * 4);
    if (argc > 2) {
        const double sec = atof(argv[2]);
        switch (atoi(argv[1])) {
        case 0: sustain<1, 32, 2, 0, 2, 0>("32x2 wave, 4 waves in y, dword, nt (kernel)", out, sec); break;
        case 1: sustain<1, 64, 1, 1, 2, 0>("64x1 wave, 4 waves in x, dword, nt", out, sec); break;
        case 2: sustain<4, 16, 4, 0, 2, 0>("64x4 wave (4 px/lane), 4 waves in y, dwordx4, nt", out, sec); break;
        case 3: sustain<4, 64, 1, 0, 2, 0>("256x1 wave (4 px/lane), 4 waves in y, dwordx4, nt", out, sec); break;
        default: sustain<1, 32, 2, 0, 0, 0>("32x2 wave, 4 waves in y, dword, default policy", out, sec); break;
        }
        return 0;
    }
    //   PPL TXL TY WGX AUX ORDER
    run<1, 32, 2, 0, 2, 0>("32x2 wave, 4 waves in y, dword, nt   (current kernel)", out);
    run<1, 32, 2, 0, 0, 0>("32x2 wave, 4 waves in y, dword, default policy", out);
    run<1, 32, 2, 0, 2, 1>("32x2 wave, 4 waves in y, dword, nt, plane chunk fastest", out);
    run<1, 32, 2, 0, 2, 2>("32x2 wave, 4 waves in y, dword, nt, no xcd remap", out);
    run<1, 32, 2, 1, 2, 0>("32x2 wave, 4 waves in x, dword, nt", out);
    run<1, 64, 1, 0, 2, 0>("64x1 wave, 4 waves in y, dword, nt", out);
    run<1, 64, 1, 1, 2, 0>("64x1 wave, 4 waves in x, dword, nt", out);
    run<2, 32, 2, 0, 2, 0>("64x2 wave (2 px/lane), 4 waves in y, dwordx2, nt", out);
    run<2, 64, 1, 0, 2, 0>("128x1 wave (2 px/lane), 4 waves in y, dwordx2, nt", out);
    // (128x1 waves side by side in x would need W % 512 == 0; at W = 768 that variant skips a third of the volume)
    run<4, 32, 2, 0, 2, 0>("128x2 wave (4 px/lane), 4 waves in y, dwordx4, nt", out);
    run<4, 64, 1, 0, 2, 0>("256x1 wave (4 px/lane), 4 waves in y, dwordx4, nt", out);
    run<4, 16, 4, 0, 2, 0>("64x4 wave (4 px/lane), 4 waves in y, dwordx4, nt", out);
    run<4, 16, 4, 0, 0, 0>("64x4 wave (4 px/lane), 4 waves in y, dwordx4, default", out);
    run<2, 16, 4, 0, 2, 0>("32x4 wave (2 px/lane), 4 waves in y, dwordx2, nt", out);
    run<1, 16, 4, 0, 2, 0>("16x4 wave, 4 waves in y, dword, nt", out);
    return 0;
}
