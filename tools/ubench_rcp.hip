// ubench_rcp.hip -- how good is v_rcp_f64's seed on gfx950, and how many Newton steps does div_pair (smvs_device.h) need?
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -o gpurun_ab/ubench_rcp tools/ubench_rcp.hip && gpurun_ab/ubench_rcp
// Prints the largest relative error (in units of 2^-53) of the seed and of the seed after one / two Newton steps, over
// 2^24 denominators of the size the RPC denominator products take (0.25 .. 4).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

__global__ void k(const double* d, double* e0, double* e1, double* e2, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = d[i];
    const double r0 = __builtin_amdgcn_rcp(x);
    const double r1 = fma(fma(-x, r0, 1.0), r0, r0);
    const double r2 = fma(fma(-x, r1, 1.0), r1, r1);
    const double t = 1.0 / x;                       // IEEE division (correctly rounded)
    e0[i] = fabs(r0 - t) / t; e1[i] = fabs(r1 - t) / t; e2[i] = fabs(r2 - t) / t;
}

int main()
{
    const int n = 1 << 24;
    double* h = (double*)malloc(n * sizeof(double));
    srand(1);
    for (int i = 0; i < n; ++i) h[i] = 0.25 * exp2(4.0 * (rand() / (double)RAND_MAX)) * (1.0 + 1e-9 * (rand() % 1000));
    double *d, *e[3];
    hipMalloc(&d, n * 8);
    for (int j = 0; j < 3; ++j) hipMalloc(&e[j], n * 8);
    hipMemcpy(d, h, n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, e[0], e[1], e[2], n);
    const char* names[3] = {"v_rcp_f64 seed", "+ 1 Newton step", "+ 2 Newton steps"};
    for (int j = 0; j < 3; ++j) {
        hipMemcpy(h, e[j], n * 8, hipMemcpyDeviceToHost);
        double m = 0;
        for (int i = 0; i < n; ++i) m = h[i] > m ? h[i] : m;
        printf("%-18s max relative error %.3e = %.3g x 2^-53\n", names[j], m, m / 1.1102230246251565e-16);
    }
    return 0;
}
