// Micro-benchmark: is data kept in the L2 / Infinity Cache across kernel boundaries?  One lane per block chases a
// pointer chain of STEPS dependent loads inside a private chunk; kernels are launched back to back in a graph.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
constexpr int STEPS = 64;
__global__ void chase(const int* chain, int stride_ints, int* out)
{
    if (threadIdx.x != 0) return;
    const int* base = chain + (size_t)blockIdx.x * stride_ints;
    int p = 0;
    for (int i = 0; i < STEPS; i++) p = base[p];
    out[blockIdx.x] = p;
}
__global__ void trash(double* buf, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) buf[i] += 1.0; }
int main()
{
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    const int nb = 256, stride = 4096;  // 16 KB per block, 4 MB total
    std::vector<int> h((size_t)nb * stride);
    for (int b = 0; b < nb; b++) for (int i = 0; i < stride; i++) h[(size_t)b * stride + i] = (i * 67 + 32) % stride;  // jumps of 67*4 B > a cache line
    int *d, *out; hipMalloc(&d, h.size() * 4); hipMalloc(&out, nb * 4); hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    double* big; const size_t nbig = (size_t)96 << 20;  // 768 MB: evicts L2 and the 256 MB Infinity Cache
    hipMalloc(&big, nbig * 8); hipMemset(big, 0, nbig * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time_chase = [&](const char* what) {
        hipEventRecord(e0, st); hipLaunchKernelGGL(chase, dim3(nb), dim3(64), 0, st, d, stride, out); hipEventRecord(e1, st); hipStreamSynchronize(st);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-46s %7.2f us  => %5.0f ns per dependent load\n", what, ms * 1e3, ms * 1e6 / STEPS);
    };
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(trash, dim3((unsigned)((nbig + 255) / 256)), dim3(256), 0, st, big, nbig); hipStreamSynchronize(st);
        time_chase("after streaming 768 MB (cold: HBM)");
        time_chase("immediately again (previous kernel touched it)");
        time_chase("and again");
        hipLaunchKernelGGL(trash, dim3((unsigned)(((size_t)8 << 20) / 256)), dim3(256), 0, st, big, (size_t)8 << 20); hipStreamSynchronize(st);
        time_chase("after a 64 MB kernel in between (L2 evicted)");
    }
    return 0;
}
