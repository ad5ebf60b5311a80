// Can a workgroup on gfx950 use more than 64 KB of LDS (the CU has 160 KB)?  Dynamic LDS of 132 KB after hipFuncSetAttribute.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double* out, int n)
{
    extern __shared__ double lds[];
    for (int i = threadIdx.x; i < n; i += blockDim.x) lds[i] = (double)i;
    __syncthreads();
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += lds[n - 1 - i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("sharedMemPerBlock %zu, maxSharedMemoryPerMultiProcessor %zu, sharedMemPerBlockOptin %zu\n", p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor, p.sharedMemPerBlockOptin);
    double* d; hipMalloc(&d, 256 * 8 * 4);
    for (size_t bytes : {size_t(64) << 10, size_t(96) << 10, size_t(132) << 10, size_t(160) << 10}) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        hipLaunchKernelGGL(k, dim3(4), dim3(256), bytes, 0, d, (int)(bytes / 8));
        hipError_t e2 = hipGetLastError(), e3 = hipDeviceSynchronize();
        double h = 0; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        printf("%zu KB: setattr %s, launch %s, sync %s, out %.0f\n", bytes >> 10, hipGetErrorString(e), hipGetErrorString(e2), hipGetErrorString(e3), h);
    }
    return 0;
}
