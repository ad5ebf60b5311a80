// Micro-benchmark: a MINIMAL colour launch of a Gauss-Seidel sweep on a small level (n rows, 4 colours, 7 entries per row at a fixed
// panel pitch of 12, SELL-64 panels) as a chain in a hipGraph -- how far is libsmg's k_sell<SELL_GS> (2.85 - 3.4 us per launch on C3
// levels 2 / 3) from what the simplest possible kernel needs for the same loads?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int PITCH = 12;
template <int W>
__global__ __launch_bounds__(256) void k_gs_min(const int* __restrict__ col, const double* __restrict__ val, const double* __restrict__ b, double* u, int s_begin, int s_end)
{
    const int lane = threadIdx.x & 63, s = s_begin + blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= s_end) return;
    const size_t off = (size_t)s * PITCH * 64 + lane;
    int c[W]; double v[W];
#pragma unroll
    for (int j = 0; j < W; j++) { c[j] = col[off + (size_t)j * 64]; v[j] = val[off + (size_t)j * 64]; }
    const int row = s * 64 + lane;
    const double bv = b[row];
    double x[W];
#pragma unroll
    for (int j = 0; j < W; j++) x[j] = (c[j] >= 0 && c[j] != row) ? u[c[j]] : 0.0;
    double acc = 0.0, diag = 1.0;
#pragma unroll
    for (int j = 0; j < W; j++) { if (c[j] == row) diag = v[j]; else if (c[j] >= 0) acc += v[j] * x[j]; }
    u[row] = (bv - acc) / diag;
}
int main()
{
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int n : {16384, 65536, 262144}) {
        const int nc = 4, per = n / nc, ns = n / 64;
        std::vector<int> col((size_t)ns * PITCH * 64, -1); std::vector<double> val(col.size(), 0.0);
        std::mt19937 rng(1);
        for (int r = 0; r < n; r++) {
            const int s = r / 64, l = r % 64, c0 = r / per;
            std::vector<int> cs{r};
            for (int j = 0; j < 6; j++) { int cc = (c0 + 1 + j % 3) % nc; int pos = (r % per) + (int)(rng() % 257) - 128; pos = std::max(0, std::min(per - 1, pos)); cs.push_back(cc * per + pos); }
            std::sort(cs.begin(), cs.end());
            for (int j = 0; j < 7; j++) { col[((size_t)s * PITCH + j) * 64 + l] = cs[j]; val[((size_t)s * PITCH + j) * 64 + l] = cs[j] == r ? 8.0 : -1.0; }
        }
        int* dcol; double *dval, *db, *du;
        CK(hipMalloc(&dcol, col.size() * 4)); CK(hipMalloc(&dval, val.size() * 8)); CK(hipMalloc(&db, n * 8)); CK(hipMalloc(&du, n * 8));
        CK(hipMemcpy(dcol, col.data(), col.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dval, val.data(), val.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemset(db, 0, n * 8)); CK(hipMemset(du, 0, n * 8));
        for (int W : {7, 12}) {
            hipGraph_t g; hipGraphExec_t ge;
            const int sweeps = 50, spc = per / 64;
            hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
            for (int sw = 0; sw < sweeps; sw++)
                for (int c = 0; c < nc; c++) {
                    if (W == 7) hipLaunchKernelGGL(k_gs_min<7>, dim3((spc + 3) / 4), dim3(256), 0, st, dcol, dval, db, du, c * spc, (c + 1) * spc);
                    else hipLaunchKernelGGL(k_gs_min<12>, dim3((spc + 3) / 4), dim3(256), 0, st, dcol, dval, db, du, c * spc, (c + 1) * spc);
                }
            hipStreamEndCapture(st, &g);
            hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipGraphLaunch(ge, st); hipStreamSynchronize(st);
            float best = 1e9f;
            for (int t = 0; t < 5; t++) { float ms; hipEventRecord(e0, st); hipGraphLaunch(ge, st); hipEventRecord(e1, st); hipStreamSynchronize(st); hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
            printf("n=%7d rows, %4d workgroups per colour launch, %2d panel columns loaded: %.2f us per colour launch\n", n, (spc + 3) / 4, W, best * 1000.f / (sweeps * nc));
            hipGraphExecDestroy(ge); hipGraphDestroy(g);
        }
        hipFree(dcol); hipFree(dval); hipFree(db); hipFree(du);
    }
    return 0;
}
