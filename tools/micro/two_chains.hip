// Micro-benchmark: can two dependency chains inside ONE hipGraph overlap each other's dispatch ramps?
// A level-0 colour launch at C3 streams 28 MB in ~4.3 us of bandwidth time but takes ~6.3 us: the rest is ramp / drain, paid once per launch
// because every colour launch waits for the whole previous one.  If the mesh were cut in two halves (+ a thin interface band), the two halves'
// chains could run side by side and only the band would tie them together.  This measures what the graph machinery makes of that:
//   A: S steps, one kernel of `bytes` per step, linear chain                                     (what the cycle does today)
//   B: S steps, two kernels of bytes/2 per step on two branches, each depending on BOTH kernels of the previous step   (full cross dependency)
//   C: the same two branches, each kernel depending only on its own predecessor                  (no cross dependency: upper bound of the gain)
//   D: two branches + a tiny band kernel per step; interior kernels depend on (own interior, band) of the previous step, the band on all three
// Built by hand with hipGraphAddKernelNode (explicit dependencies).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_stream(const double* __restrict__ x, double* __restrict__ y, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    double s = 0.0;
    for (; i < n; i += stride) s += x[i];
    if (s == 123.456) y[0] = s;
}
struct Node { hipGraphNode_t n; };
static hipGraphNode_t add(hipGraph_t g, const std::vector<hipGraphNode_t>& deps, const double* x, double* y, size_t n, int blocks)
{
    hipKernelNodeParams p = {};
    void* args[3] = {(void*)&x, (void*)&y, (void*)&n};
    p.func = (void*)k_stream; p.gridDim = dim3(blocks); p.blockDim = dim3(256); p.sharedMemBytes = 0; p.kernelParams = args; p.extra = nullptr;
    hipGraphNode_t node;
    if (hipGraphAddKernelNode(&node, g, deps.empty() ? nullptr : deps.data(), deps.size(), &p) != hipSuccess) { printf("add node failed\n"); exit(1); }
    return node;
}
static float time_graph(hipGraph_t g, hipStream_t st, int steps)
{
    hipGraphExec_t ge;
    if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) { printf("instantiate failed\n"); exit(1); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    float best = 1e9f;
    for (int t = 0; t < 7; t++) {
        float ms = 0;
        hipEventRecord(e0, st); for (int r = 0; r < 5; r++) hipGraphLaunch(ge, st); hipEventRecord(e1, st); hipStreamSynchronize(st);
        hipEventElapsedTime(&ms, e0, e1);
        if (ms / 5 < best) best = ms / 5;
    }
    hipGraphExecDestroy(ge);
    return best * 1000.f / steps;
}
int main()
{
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const size_t nmax = (size_t)64 << 20;      // doubles: 512 MB
    double *x, *y;
    CK(hipMalloc(&x, nmax * sizeof(double))); CK(hipMalloc(&y, 1024));
    CK(hipMemset(x, 0, nmax * sizeof(double)));
    const int S = 64;
    for (size_t mb : {7, 28, 112}) {
        const size_t n = mb * 1024 * 1024 / 8;
        const int blocks = (int)((n / 256 + 7) / 8);          // 8 elements per thread
        float tA, tB, tC, tD;
        { hipGraph_t g; CK(hipGraphCreate(&g, 0)); std::vector<hipGraphNode_t> prev;
          for (int s = 0; s < S; s++) { hipGraphNode_t a = add(g, prev, x, y, n, blocks); prev = {a}; }
          tA = time_graph(g, st, S); hipGraphDestroy(g); }
        { hipGraph_t g; CK(hipGraphCreate(&g, 0)); std::vector<hipGraphNode_t> prev;
          for (int s = 0; s < S; s++) { hipGraphNode_t a = add(g, prev, x, y, n / 2, blocks / 2), b = add(g, prev, x + n / 2, y, n / 2, blocks / 2); prev = {a, b}; }
          tB = time_graph(g, st, S); hipGraphDestroy(g); }
        { hipGraph_t g; CK(hipGraphCreate(&g, 0)); std::vector<hipGraphNode_t> pa, pb;
          for (int s = 0; s < S; s++) { hipGraphNode_t a = add(g, pa, x, y, n / 2, blocks / 2), b = add(g, pb, x + n / 2, y, n / 2, blocks / 2); pa = {a}; pb = {b}; }
          tC = time_graph(g, st, S); hipGraphDestroy(g); }
        { hipGraph_t g; CK(hipGraphCreate(&g, 0)); std::vector<hipGraphNode_t> pa, pb, pc;
          const size_t nb = n / 32;      // the band: ~3 % of the rows
          for (int s = 0; s < S; s++) {
              std::vector<hipGraphNode_t> da = pa, db = pb, dc = pa;
              da.insert(da.end(), pc.begin(), pc.end()); db.insert(db.end(), pc.begin(), pc.end());
              dc.insert(dc.end(), pb.begin(), pb.end()); dc.insert(dc.end(), pc.begin(), pc.end());
              hipGraphNode_t a = add(g, da, x, y, (n - nb) / 2, blocks / 2), b = add(g, db, x + n / 2, y, (n - nb) / 2, blocks / 2), c = add(g, dc, x + n - nb, y, nb, (blocks + 31) / 32);
              pa = {a}; pb = {b}; pc = {c};
          }
          tD = time_graph(g, st, S); hipGraphDestroy(g); }
        printf("%4zu MB per step: A one kernel %.2f us | B two halves, full cross dependency %.2f | C two independent chains %.2f | D two chains + band %.2f us per step\n", mb, tA, tB, tC, tD);
    }
    return 0;
}
