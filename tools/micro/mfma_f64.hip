// Micro-benchmark: sustained rate of v_mfma_f64_16x16x4_f64 (the instruction behind the coarse-matrix inversion): four independent
// accumulator chains per wave, 1..4 waves per SIMD, operands in registers.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef double v4f64 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(double* out, int iters, double a0, double b0)
{
    v4f64 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    double a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int c = 0; c < 4; c++) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
    }
    double s = 0;
    for (int c = 0; c < 4; c++) for (int r = 0; r < 4; r++) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main()
{
    double* d; CK(hipMalloc(&d, 8 * 256 * 4096));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int wg_per_cu : {1, 2, 4}) {
        const int grid = 256 * wg_per_cu, iters = 4096;
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, 16, 1.0, 2.0);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, iters, 1.0, 2.0); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double flops = (double)grid * 4 /*waves*/ * iters * 16.0 * 2048.0;
        printf("%d workgroup(s) of 4 waves per CU: %.1f TFLOP/s fp64 MFMA (%.2f ms)\n", wg_per_cu, flops / (ms * 1e-3) / 1e12, ms);
    }
    return 0;
}
