// stream_read.hip -- what does a plain streaming read reach on this MI355X, from HBM (buffer >> 256 MB Infinity Cache) and
// from a buffer that fits the Infinity Cache?  Context for the SpMV roofline fractions.
// build: hipcc --offload-arch=gfx950 -O3 -o stream_read stream_read.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// every lane reads 16 B per load, UNROLL loads in flight, grid-stride over the buffer; one partial sum per block is written
template <int UNROLL>
__global__ __launch_bounds__(256) void k_read(const double2* __restrict__ p, size_t n2, double* out)
{
    double s = 0.0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n2; i += UNROLL * stride) {
        double2 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) v[u] = p[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) s += v[u].x + v[u].y;
    }
    for (; i < n2; i += stride) s += p[i].x + p[i].y;
    __shared__ double red[256];
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0; for (int k = 0; k < 256; k++) t += red[k]; out[blockIdx.x] = t; }
}

int main()
{
    double* out; CHK(hipMalloc(&out, 1 << 20));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (size_t mb : {64, 128, 352, 1024, 4096}) {
        const size_t bytes = mb << 20, n2 = bytes / 16;
        double2* p; CHK(hipMalloc(&p, bytes)); CHK(hipMemset(p, 0, bytes));
        for (int blocks : {2048, 8192, 32768}) {
            float best = 1e30f;
            for (int rep = 0; rep < 12; rep++) {
                CHK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(k_read<8>, dim3(blocks), dim3(256), 0, 0, p, n2, out);
                CHK(hipEventRecord(e1, 0));
                CHK(hipDeviceSynchronize());
                float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
                if (rep >= 2 && ms < best) best = ms;
            }
            printf("%5zu MB, %5d blocks: %7.1f us  %6.2f TB/s\n", mb, blocks, 1e3 * best, bytes / (best * 1e-3) / 1e12);
        }
        CHK(hipFree(p));
    }
    return 0;
}
