// Micro-benchmark: cost of dependent kernel launches on one stream (eager and hipGraph), same kernel vs alternating kernels.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#pragma clang diagnostic ignored "-Wunused-value"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void ka(double* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] + 1.0; }
__global__ void kb(double* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0000001; }
__global__ void kc(double* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] - 0.5; }
static float run(hipStream_t st, double* d, int n, int reps, int pattern, bool graph)
{
    auto enqueue = [&]() {
        for (int i = 0; i < reps; i++) {
            int which = pattern == 0 ? 0 : (pattern == 1 ? i % 2 : i % 3);
            int nb = (n + 255) / 256;
            if (which == 0) hipLaunchKernelGGL(ka, dim3(nb), dim3(256), 0, st, d, n);
            else if (which == 1) hipLaunchKernelGGL(kb, dim3(nb), dim3(256), 0, st, d, n);
            else hipLaunchKernelGGL(kc, dim3(nb), dim3(256), 0, st, d, n);
        }
    };
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    if (graph) {
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal); enqueue(); hipStreamEndCapture(st, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, st); hipStreamSynchronize(st);
        hipEventRecord(e0, st); for (int r = 0; r < 10; r++) hipGraphLaunch(ge, st); hipEventRecord(e1, st); hipStreamSynchronize(st);
        hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    } else {
        enqueue(); hipStreamSynchronize(st);
        hipEventRecord(e0, st); enqueue(); hipEventRecord(e1, st); hipStreamSynchronize(st);
        hipEventElapsedTime(&ms, e0, e1);
    }
    return ms * 1000.f / reps;
}
int main()
{
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int nmax = 1 << 22; double* d; CK(hipMalloc(&d, nmax * sizeof(double))); CK(hipMemset(d, 0, nmax * sizeof(double)));
    for (int n : {256, 16384, 262144, 1 << 22})
        for (int graph = 0; graph < 2; graph++)
            for (int pattern = 0; pattern < 3; pattern++)
                printf("n=%8d %s pattern=%s : %.2f us/kernel\n", n, graph ? "graph" : "eager", pattern == 0 ? "same" : (pattern == 1 ? "alt2" : "alt3"), run(st, d, n, 200, pattern, graph));
    return 0;
}
