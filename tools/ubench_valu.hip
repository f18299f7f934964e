// ubench_valu.hip -- per-instruction VALU issue cost on gfx950, the cost model behind DESIGN.md section 4.
//   hipcc --offload-arch=gfx950 -O2 -o gpurun_ab/ubench_valu tools/ubench_valu.hip && gpurun_ab/ubench_valu
// Every kernel runs REP x 32 copies of one instruction on 8 independent register sets, timed per wave with
// s_memtime (shader cycles); waves/SIMD = 1, 2, 4 to separate issue cost from latency.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define REP 512

#define KERNEL(name, body)                                                                   \
    __global__ void k_##name(unsigned long long* out, double* sink)                          \
    {                                                                                        \
        double d0 = threadIdx.x + 1.0, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, d4 = d0 + 4,  \
               d5 = d0 + 5, d6 = d0 + 6, d7 = d0 + 7, dm = 1.0000001, da = 1e-9;            \
        float f0 = threadIdx.x + 1.0f, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4,  \
              f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7, fm = 1.0000001f, fa = 1e-9f;           \
        int s0 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) + 3;                       \
        double sd = __builtin_bit_cast(double, ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(0x3ff00000 + (int)(threadIdx.x >> 6)) << 32)); \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                \
        for (int i = 0; i < REP; ++i) {                                                      \
            asm volatile(body body body body                                                 \
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), \
                           "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7), "+s"(s0) \
                         : "v"(dm), "v"(da), "v"(fm), "v"(fa), "s"(sd) : "scc", "vcc", "s20", "s21", "s22", "s23");                     \
        }                                                                                    \
        asm volatile("s_waitcnt lgkmcnt(0)");                                                \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0; \
        sink[blockIdx.x * blockDim.x + threadIdx.x] = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + s0; \
    }

// operands: %0-%7 double regs, %8-%15 float regs, %16 sgpr int, %17 dm, %18 da, %19 fm, %20 fa, %21 sgpr double
#define X8(a, b, c, d, e, f, g, h) a "\n" b "\n" c "\n" d "\n" e "\n" f "\n" g "\n" h "\n"

KERNEL(fma_f32, X8("v_fma_f32 %8, %8, %19, %20", "v_fma_f32 %9, %9, %19, %20", "v_fma_f32 %10, %10, %19, %20", "v_fma_f32 %11, %11, %19, %20",
                   "v_fma_f32 %12, %12, %19, %20", "v_fma_f32 %13, %13, %19, %20", "v_fma_f32 %14, %14, %19, %20", "v_fma_f32 %15, %15, %19, %20"))
KERNEL(mul_f32, X8("v_mul_f32 %8, %8, %19", "v_mul_f32 %9, %9, %19", "v_mul_f32 %10, %10, %19", "v_mul_f32 %11, %11, %19",
                   "v_mul_f32 %12, %12, %19", "v_mul_f32 %13, %13, %19", "v_mul_f32 %14, %14, %19", "v_mul_f32 %15, %15, %19"))
KERNEL(pk_fma_f32, X8("v_pk_fma_f32 %0, %0, %17, %18", "v_pk_fma_f32 %1, %1, %17, %18", "v_pk_fma_f32 %2, %2, %17, %18", "v_pk_fma_f32 %3, %3, %17, %18",
                      "v_pk_fma_f32 %4, %4, %17, %18", "v_pk_fma_f32 %5, %5, %17, %18", "v_pk_fma_f32 %6, %6, %17, %18", "v_pk_fma_f32 %7, %7, %17, %18"))
KERNEL(pk_mul_f32, X8("v_pk_mul_f32 %0, %0, %17", "v_pk_mul_f32 %1, %1, %17", "v_pk_mul_f32 %2, %2, %17", "v_pk_mul_f32 %3, %3, %17",
                      "v_pk_mul_f32 %4, %4, %17", "v_pk_mul_f32 %5, %5, %17", "v_pk_mul_f32 %6, %6, %17", "v_pk_mul_f32 %7, %7, %17"))
KERNEL(pk_add_f32, X8("v_pk_add_f32 %0, %0, %17", "v_pk_add_f32 %1, %1, %17", "v_pk_add_f32 %2, %2, %17", "v_pk_add_f32 %3, %3, %17",
                      "v_pk_add_f32 %4, %4, %17", "v_pk_add_f32 %5, %5, %17", "v_pk_add_f32 %6, %6, %17", "v_pk_add_f32 %7, %7, %17"))
KERNEL(fma_f64, X8("v_fma_f64 %0, %0, %17, %18", "v_fma_f64 %1, %1, %17, %18", "v_fma_f64 %2, %2, %17, %18", "v_fma_f64 %3, %3, %17, %18",
                   "v_fma_f64 %4, %4, %17, %18", "v_fma_f64 %5, %5, %17, %18", "v_fma_f64 %6, %6, %17, %18", "v_fma_f64 %7, %7, %17, %18"))
KERNEL(fmac_f64_sgpr, X8("v_fmac_f64 %0, %21, %17", "v_fmac_f64 %1, %21, %17", "v_fmac_f64 %2, %21, %17", "v_fmac_f64 %3, %21, %17",
                         "v_fmac_f64 %4, %21, %17", "v_fmac_f64 %5, %21, %17", "v_fmac_f64 %6, %21, %17", "v_fmac_f64 %7, %21, %17"))
KERNEL(fma_f64_sgpr_addend, X8("v_fma_f64 %0, %0, %17, %21", "v_fma_f64 %1, %1, %17, %21", "v_fma_f64 %2, %2, %17, %21", "v_fma_f64 %3, %3, %17, %21",
                               "v_fma_f64 %4, %4, %17, %21", "v_fma_f64 %5, %5, %17, %21", "v_fma_f64 %6, %6, %17, %21", "v_fma_f64 %7, %7, %17, %21"))
KERNEL(mul_f64, X8("v_mul_f64 %0, %0, %17", "v_mul_f64 %1, %1, %17", "v_mul_f64 %2, %2, %17", "v_mul_f64 %3, %3, %17",
                   "v_mul_f64 %4, %4, %17", "v_mul_f64 %5, %5, %17", "v_mul_f64 %6, %6, %17", "v_mul_f64 %7, %7, %17"))
KERNEL(add_f64, X8("v_add_f64 %0, %0, %18", "v_add_f64 %1, %1, %18", "v_add_f64 %2, %2, %18", "v_add_f64 %3, %3, %18",
                   "v_add_f64 %4, %4, %18", "v_add_f64 %5, %5, %18", "v_add_f64 %6, %6, %18", "v_add_f64 %7, %7, %18"))
KERNEL(rcp_f64, X8("v_rcp_f64 %0, %0", "v_rcp_f64 %1, %1", "v_rcp_f64 %2, %2", "v_rcp_f64 %3, %3",
                   "v_rcp_f64 %4, %4", "v_rcp_f64 %5, %5", "v_rcp_f64 %6, %6", "v_rcp_f64 %7, %7"))
KERNEL(rcp_f32, X8("v_rcp_f32 %8, %8", "v_rcp_f32 %9, %9", "v_rcp_f32 %10, %10", "v_rcp_f32 %11, %11",
                   "v_rcp_f32 %12, %12", "v_rcp_f32 %13, %13", "v_rcp_f32 %14, %14", "v_rcp_f32 %15, %15"))
KERNEL(cvt_f32_f64, X8("v_cvt_f32_f64 %8, %0", "v_cvt_f32_f64 %9, %1", "v_cvt_f32_f64 %10, %2", "v_cvt_f32_f64 %11, %3",
                       "v_cvt_f32_f64 %12, %4", "v_cvt_f32_f64 %13, %5", "v_cvt_f32_f64 %14, %6", "v_cvt_f32_f64 %15, %7"))
KERNEL(cvt_f64_f32, X8("v_cvt_f64_f32 %0, %8", "v_cvt_f64_f32 %1, %9", "v_cvt_f64_f32 %2, %10", "v_cvt_f64_f32 %3, %11",
                       "v_cvt_f64_f32 %4, %12", "v_cvt_f64_f32 %5, %13", "v_cvt_f64_f32 %6, %14", "v_cvt_f64_f32 %7, %15"))
KERNEL(mov_b32, X8("v_mov_b32 %8, %19", "v_mov_b32 %9, %19", "v_mov_b32 %10, %19", "v_mov_b32 %11, %19",
                   "v_mov_b32 %12, %19", "v_mov_b32 %13, %19", "v_mov_b32 %14, %19", "v_mov_b32 %15, %19"))
KERNEL(mov_b64, X8("v_mov_b64 %0, %17", "v_mov_b64 %1, %17", "v_mov_b64 %2, %17", "v_mov_b64 %3, %17",
                   "v_mov_b64 %4, %17", "v_mov_b64 %5, %17", "v_mov_b64 %6, %17", "v_mov_b64 %7, %17"))
KERNEL(mov_b64_sgpr, X8("v_mov_b64 %0, %21", "v_mov_b64 %1, %21", "v_mov_b64 %2, %21", "v_mov_b64 %3, %21",
                        "v_mov_b64 %4, %21", "v_mov_b64 %5, %21", "v_mov_b64 %6, %21", "v_mov_b64 %7, %21"))
KERNEL(readlane, X8("v_readlane_b32 %16, %8, 3", "v_readlane_b32 %16, %9, 3", "v_readlane_b32 %16, %10, 3", "v_readlane_b32 %16, %11, 3",
                    "v_readlane_b32 %16, %12, 3", "v_readlane_b32 %16, %13, 3", "v_readlane_b32 %16, %14, 3", "v_readlane_b32 %16, %15, 3"))
KERNEL(writelane, X8("v_writelane_b32 %8, %16, 3", "v_writelane_b32 %9, %16, 3", "v_writelane_b32 %10, %16, 3", "v_writelane_b32 %11, %16, 3",
                     "v_writelane_b32 %12, %16, 3", "v_writelane_b32 %13, %16, 3", "v_writelane_b32 %14, %16, 3", "v_writelane_b32 %15, %16, 3"))
KERNEL(min_i32, X8("v_min_i32 %8, %8, %19", "v_min_i32 %9, %9, %19", "v_min_i32 %10, %10, %19", "v_min_i32 %11, %11, %19",
                   "v_min_i32 %12, %12, %19", "v_min_i32 %13, %13, %19", "v_min_i32 %14, %14, %19", "v_min_i32 %15, %15, %19"))
KERNEL(min_i32_dpp, X8("v_min_i32_dpp %8, %8, %8 row_shr:1 row_mask:0xf bank_mask:0xf", "v_min_i32_dpp %9, %9, %9 row_shr:1 row_mask:0xf bank_mask:0xf",
                       "v_min_i32_dpp %10, %10, %10 row_shr:1 row_mask:0xf bank_mask:0xf", "v_min_i32_dpp %11, %11, %11 row_shr:1 row_mask:0xf bank_mask:0xf",
                       "v_min_i32_dpp %12, %12, %12 row_shr:1 row_mask:0xf bank_mask:0xf", "v_min_i32_dpp %13, %13, %13 row_shr:1 row_mask:0xf bank_mask:0xf",
                       "v_min_i32_dpp %14, %14, %14 row_shr:1 row_mask:0xf bank_mask:0xf", "v_min_i32_dpp %15, %15, %15 row_shr:1 row_mask:0xf bank_mask:0xf"))
KERNEL(floor_f32, X8("v_floor_f32 %8, %8", "v_floor_f32 %9, %9", "v_floor_f32 %10, %10", "v_floor_f32 %11, %11",
                     "v_floor_f32 %12, %12", "v_floor_f32 %13, %13", "v_floor_f32 %14, %14", "v_floor_f32 %15, %15"))
KERNEL(cvt_i32_f32, X8("v_cvt_i32_f32 %8, %8", "v_cvt_i32_f32 %9, %9", "v_cvt_i32_f32 %10, %10", "v_cvt_i32_f32 %11, %11",
                       "v_cvt_i32_f32 %12, %12", "v_cvt_i32_f32 %13, %13", "v_cvt_i32_f32 %14, %14", "v_cvt_i32_f32 %15, %15"))
KERNEL(cndmask, X8("v_cndmask_b32 %8, %8, %19, vcc", "v_cndmask_b32 %9, %9, %19, vcc", "v_cndmask_b32 %10, %10, %19, vcc", "v_cndmask_b32 %11, %11, %19, vcc",
                   "v_cndmask_b32 %12, %12, %19, vcc", "v_cndmask_b32 %13, %13, %19, vcc", "v_cndmask_b32 %14, %14, %19, vcc", "v_cndmask_b32 %15, %15, %19, vcc"))
KERNEL(salu_add, X8("s_add_u32 %16, %16, 1", "s_add_u32 %16, %16, 1", "s_add_u32 %16, %16, 1", "s_add_u32 %16, %16, 1",
                    "s_add_u32 %16, %16, 1", "s_add_u32 %16, %16, 1", "s_add_u32 %16, %16, 1", "s_add_u32 %16, %16, 1"))
// mixed: does a 2-cycle f32 op pair with a 4-cycle f64 op (dual issue from one wave? no -- but shows the sum)
KERNEL(mix_f64_f32, X8("v_fma_f64 %0, %0, %17, %18", "v_fma_f32 %8, %8, %19, %20", "v_fma_f64 %1, %1, %17, %18", "v_fma_f32 %9, %9, %19, %20",
                       "v_fma_f64 %2, %2, %17, %18", "v_fma_f32 %10, %10, %19, %20", "v_fma_f64 %3, %3, %17, %18", "v_fma_f32 %11, %11, %19, %20"))

KERNEL(add_f32, X8("v_add_f32 %8, %8, %19", "v_add_f32 %9, %9, %19", "v_add_f32 %10, %10, %19", "v_add_f32 %11, %11, %19", "v_add_f32 %12, %12, %19", "v_add_f32 %13, %13, %19", "v_add_f32 %14, %14, %19", "v_add_f32 %15, %15, %19"))
KERNEL(sub_f32, X8("v_sub_f32 %8, %8, %19", "v_sub_f32 %9, %9, %19", "v_sub_f32 %10, %10, %19", "v_sub_f32 %11, %11, %19", "v_sub_f32 %12, %12, %19", "v_sub_f32 %13, %13, %19", "v_sub_f32 %14, %14, %19", "v_sub_f32 %15, %15, %19"))
KERNEL(max_f32, X8("v_max_f32 %8, %8, %19", "v_max_f32 %9, %9, %19", "v_max_f32 %10, %10, %19", "v_max_f32 %11, %11, %19", "v_max_f32 %12, %12, %19", "v_max_f32 %13, %13, %19", "v_max_f32 %14, %14, %19", "v_max_f32 %15, %15, %19"))
KERNEL(add_u32, X8("v_add_u32 %8, %8, %19", "v_add_u32 %9, %9, %19", "v_add_u32 %10, %10, %19", "v_add_u32 %11, %11, %19", "v_add_u32 %12, %12, %19", "v_add_u32 %13, %13, %19", "v_add_u32 %14, %14, %19", "v_add_u32 %15, %15, %19"))
KERNEL(and_b32, X8("v_and_b32 %8, %8, %19", "v_and_b32 %9, %9, %19", "v_and_b32 %10, %10, %19", "v_and_b32 %11, %11, %19", "v_and_b32 %12, %12, %19", "v_and_b32 %13, %13, %19", "v_and_b32 %14, %14, %19", "v_and_b32 %15, %15, %19"))
KERNEL(lshl_add_u32, X8("v_lshl_add_u32 %8, %8, 2, %19", "v_lshl_add_u32 %9, %9, 2, %19", "v_lshl_add_u32 %10, %10, 2, %19", "v_lshl_add_u32 %11, %11, 2, %19", "v_lshl_add_u32 %12, %12, 2, %19", "v_lshl_add_u32 %13, %13, 2, %19", "v_lshl_add_u32 %14, %14, 2, %19", "v_lshl_add_u32 %15, %15, 2, %19"))
KERNEL(mad_u32_u24, X8("v_mad_u32_u24 %8, %8, %19, %20", "v_mad_u32_u24 %9, %9, %19, %20", "v_mad_u32_u24 %10, %10, %19, %20", "v_mad_u32_u24 %11, %11, %19, %20", "v_mad_u32_u24 %12, %12, %19, %20", "v_mad_u32_u24 %13, %13, %19, %20", "v_mad_u32_u24 %14, %14, %19, %20", "v_mad_u32_u24 %15, %15, %19, %20"))
KERNEL(cvt_f32_i32, X8("v_cvt_f32_i32 %8, %8", "v_cvt_f32_i32 %9, %9", "v_cvt_f32_i32 %10, %10", "v_cvt_f32_i32 %11, %11", "v_cvt_f32_i32 %12, %12", "v_cvt_f32_i32 %13, %13", "v_cvt_f32_i32 %14, %14", "v_cvt_f32_i32 %15, %15"))
KERNEL(cndmask_e64_sgpr, X8("v_cndmask_b32_e64 %8, %8, %19, s[20:21]", "v_cndmask_b32_e64 %9, %9, %19, s[20:21]", "v_cndmask_b32_e64 %10, %10, %19, s[20:21]", "v_cndmask_b32_e64 %11, %11, %19, s[20:21]", "v_cndmask_b32_e64 %12, %12, %19, s[20:21]", "v_cndmask_b32_e64 %13, %13, %19, s[20:21]", "v_cndmask_b32_e64 %14, %14, %19, s[20:21]", "v_cndmask_b32_e64 %15, %15, %19, s[20:21]"))
KERNEL(cndmask_e64_const, X8("v_cndmask_b32_e64 %8, 0, 1.0, s[20:21]", "v_cndmask_b32_e64 %9, 0, 1.0, s[20:21]", "v_cndmask_b32_e64 %10, 0, 1.0, s[20:21]", "v_cndmask_b32_e64 %11, 0, 1.0, s[20:21]", "v_cndmask_b32_e64 %12, 0, 1.0, s[20:21]", "v_cndmask_b32_e64 %13, 0, 1.0, s[20:21]", "v_cndmask_b32_e64 %14, 0, 1.0, s[20:21]", "v_cndmask_b32_e64 %15, 0, 1.0, s[20:21]"))
KERNEL(cmp_f32_vcc, X8("v_cmp_le_f32 vcc, %8, %19", "v_cmp_le_f32 vcc, %9, %19", "v_cmp_le_f32 vcc, %10, %19", "v_cmp_le_f32 vcc, %11, %19", "v_cmp_le_f32 vcc, %12, %19", "v_cmp_le_f32 vcc, %13, %19", "v_cmp_le_f32 vcc, %14, %19", "v_cmp_le_f32 vcc, %15, %19"))
KERNEL(cmp_u32_sgpr, X8("v_cmp_le_u32_e64 s[22:23], %8, %19", "v_cmp_le_u32_e64 s[22:23], %9, %19", "v_cmp_le_u32_e64 s[22:23], %10, %19", "v_cmp_le_u32_e64 s[22:23], %11, %19", "v_cmp_le_u32_e64 s[22:23], %12, %19", "v_cmp_le_u32_e64 s[22:23], %13, %19", "v_cmp_le_u32_e64 s[22:23], %14, %19", "v_cmp_le_u32_e64 s[22:23], %15, %19"))
KERNEL(cmp_then_cnd, X8("v_cmp_le_f32 vcc, %8, %19\nv_cndmask_b32 %8, %8, %20, vcc", "v_cmp_le_f32 vcc, %9, %19\nv_cndmask_b32 %9, %9, %20, vcc", "v_cmp_le_f32 vcc, %10, %19\nv_cndmask_b32 %10, %10, %20, vcc", "v_cmp_le_f32 vcc, %11, %19\nv_cndmask_b32 %11, %11, %20, vcc", "v_cmp_le_f32 vcc, %12, %19\nv_cndmask_b32 %12, %12, %20, vcc", "v_cmp_le_f32 vcc, %13, %19\nv_cndmask_b32 %13, %13, %20, vcc", "v_cmp_le_f32 vcc, %14, %19\nv_cndmask_b32 %14, %14, %20, vcc", "v_cmp_le_f32 vcc, %15, %19\nv_cndmask_b32 %15, %15, %20, vcc"))

typedef void (*kern_t)(unsigned long long*, double*);
struct Entry { const char* name; kern_t k; };
#define E(n) {#n, k_##n}

int main()
{
    Entry es[] = {E(fma_f32), E(mul_f32), E(pk_fma_f32), E(pk_mul_f32), E(pk_add_f32), E(fma_f64), E(fmac_f64_sgpr), E(fma_f64_sgpr_addend),
                  E(mul_f64), E(add_f64), E(rcp_f64), E(rcp_f32), E(cvt_f32_f64), E(cvt_f64_f32), E(mov_b32), E(mov_b64), E(mov_b64_sgpr),
                  E(readlane), E(writelane), E(min_i32), E(min_i32_dpp), E(floor_f32), E(cvt_i32_f32), E(cndmask), E(salu_add), E(mix_f64_f32), E(add_f32), E(sub_f32), E(max_f32), E(add_u32), E(and_b32), E(lshl_add_u32), E(mad_u32_u24), E(cvt_f32_i32), E(cndmask_e64_sgpr), E(cndmask_e64_const), E(cmp_f32_vcc), E(cmp_u32_sgpr), E(cmp_then_cnd)};
    setvbuf(stdout, NULL, _IONBF, 0);
    const int nblk = 256;
    unsigned long long* out; double* sink;
    hipMalloc(&out, 2 * nblk * 16 * sizeof(*out));
    hipMalloc(&sink, 2 * nblk * 1024 * sizeof(*sink));
    unsigned long long* h = (unsigned long long*)malloc(nblk * 16 * sizeof(*h));
    printf("%-22s %10s %10s %10s   (shader cycles per wave-instruction, mean over waves; waves/SIMD = 1, 2, 4)\n", "instruction", "1w", "2w", "4w");
    hipEvent_t ea, eb;
    hipEventCreate(&ea); hipEventCreate(&eb);
    for (auto& e : es) {
        printf("%-22s", e.name);
        for (int wps : {1, 2, 4}) {
            const int threads = 256 * wps;     // 4 SIMDs x wps waves, one workgroup per CU
            for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(e.k, dim3(nblk), dim3(threads), 0, 0, out, sink);
            hipDeviceSynchronize();
            hipMemcpy(h, out, nblk * (threads / 64) * sizeof(*h), hipMemcpyDeviceToHost);
            double s = 0; int n = nblk * (threads / 64);
            for (int i = 0; i < n; ++i) s += (double)h[i];
            printf(" %10.2f", s / n / (REP * 32.0) / wps);
        }
        // wall clock: 8 waves/SIMD (two 1024-thread workgroups per CU), ns per wave-instruction per SIMD
        hipEventRecord(ea);
        const int reps = 20;
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(e.k, dim3(2 * nblk), dim3(1024), 0, 0, out, sink);
        hipEventRecord(eb);
        hipEventSynchronize(eb);
        float ms = 0;
        hipEventElapsedTime(&ms, ea, eb);
        printf("   wall @8w: %6.3f ns/instr/SIMD\n", ms * 1e6 / reps / (REP * 32.0 * 8));
    }
    return 0;
}
