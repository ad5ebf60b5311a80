// Micro-benchmark: what a launch of a SMALL level can cost at best.  A chain of dependent kernels in one hipGraph, each reading what the
// previous one wrote, with D dependent memory round trips inside the kernel:
//   D = 0: y[i] = const                                  (launch floor)
//   D = 1: y[i] = x[i] + 1                               (one round trip: x was written by the previous launch, on another CU)
//   D = 2: y[i] = x[idx[i]] + 1                          (index load, then the gather: what an SpMV / colour launch does per panel column)
//   D = 3: y[i] = x[idx2[idx[i]]] + 1                    (+ a slice-table read in front: what a compact-panel launch does)
// n = 16 384 rows (C3 level 3) and 65 536 (level 2), 64 / 256 workgroups of 256 threads, ping-pong between two vectors.
// Result = us per launch = launch floor + D x (round trip to data the previous kernel wrote: L2 write-back + miss in another XCD's L2).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int D>
__global__ void kd(const double* __restrict__ x, double* __restrict__ y, const int* __restrict__ idx, const int* __restrict__ idx2, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (D == 0) y[i] = 1.0;
    else if (D == 1) y[i] = x[i] + 1.0;
    else if (D == 2) y[i] = x[idx[i]] + 1.0;
    else y[i] = x[idx2[idx[i]]] + 1.0;
}
template <int D>
static float run(hipStream_t st, double* a, double* b, const int* idx, const int* idx2, int n, int reps)
{
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(kd<D>, dim3((n + 255) / 256), dim3(256), 0, st, (r & 1) ? b : a, (r & 1) ? a : b, idx, idx2, n);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    float best = 1e9f;
    for (int t = 0; t < 5; t++) {
        float ms = 0;
        hipEventRecord(e0, st); for (int r = 0; r < 5; r++) hipGraphLaunch(ge, st); hipEventRecord(e1, st); hipStreamSynchronize(st);
        hipEventElapsedTime(&ms, e0, e1);
        if (ms / 5 < best) best = ms / 5;
    }
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return best * 1000.f / reps;
}
int main()
{
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int nmax = 1 << 18;
    double *a, *b; int *idx, *idx2;
    CK(hipMalloc(&a, nmax * sizeof(double))); CK(hipMalloc(&b, nmax * sizeof(double)));
    CK(hipMalloc(&idx, nmax * sizeof(int))); CK(hipMalloc(&idx2, nmax * sizeof(int)));
    CK(hipMemset(a, 0, nmax * sizeof(double))); CK(hipMemset(b, 0, nmax * sizeof(double)));
    for (int n : {4096, 16384, 65536, 262144}) {
        std::vector<int> h(n), h2(n);
        for (int i = 0; i < n; i++) { h[i] = (int)(((long)i * 7919 + 13) % n); h2[i] = (int)(((long)i * 104729 + 7) % n); }   // scattered like mesh neighbours after colouring
        CK(hipMemcpy(idx, h.data(), n * sizeof(int), hipMemcpyHostToDevice)); CK(hipMemcpy(idx2, h2.data(), n * sizeof(int), hipMemcpyHostToDevice));
        const float t0 = run<0>(st, a, b, idx, idx2, n, 200), t1 = run<1>(st, a, b, idx, idx2, n, 200), t2 = run<2>(st, a, b, idx, idx2, n, 200), t3 = run<3>(st, a, b, idx, idx2, n, 200);
        printf("n=%7d (%4d workgroups): launch floor %.2f us | 1 dependent round trip %.2f | 2 (index -> gather) %.2f | 3 (table -> index -> gather) %.2f us per launch\n", n, (n + 255) / 256, t0, t1, t2, t3);
    }
    return 0;
}
