// Micro-benchmark: a chain of small dependent "colour sweep" launches (each reads what the previous ones wrote through an
// index gather) with the workgroups spread over all 8 XCDs, against the same work confined to the workgroups of ONE XCD
// (grid 8x larger, blockIdx % 8 != 0 exits at once): does keeping a small level inside one L2 shorten the launch-to-launch chain?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#pragma clang diagnostic ignored "-Wunused-value"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int W = 8;
template <int NX, int BS = 256>   // XCDs used: workgroup g runs on XCD g % 8; BS threads per workgroup
__global__ __launch_bounds__(BS) void k_sweep(const int* __restrict__ col, const double* __restrict__ val, double* u, const double* __restrict__ b, int r0, int r1, int n)
{
    int bid = blockIdx.x;
    if (NX < 8) { if ((bid & 7) >= NX) return; bid = (bid >> 3) * NX + (bid & 7); }
    const int i = r0 + bid * BS + threadIdx.x;
    if (i >= r1) return;
    double acc = 0.0, d = 1.0;
#pragma unroll
    for (int t = 0; t < W; t++) {
        const int c = col[(size_t)t * n + i];   // column-major panel, like SELL
        const double v = val[(size_t)t * n + i];
        if (c == i) d = v; else acc += v * u[c];
    }
    u[i] = (b[i] - acc) / d;
}
int main()
{
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int n : {4096, 16384, 65536, 262144}) {
        const int ncol = 4, per = n / ncol;
        std::vector<int> col((size_t)W * n); std::vector<double> val((size_t)W * n, -0.1), b(n, 1.0);
        srand(1);
        for (int i = 0; i < n; i++) {
            const int ci = i / per;
            col[i] = i; val[i] = 2.0;   // diagonal first
            for (int t = 1; t < W; t++) {
                int oc = (ci + 1 + rand() % (ncol - 1)) % ncol;              // a neighbour of another colour, nearby position
                int pos = (i % per) + (rand() % 129) - 64; if (pos < 0) pos = 0; if (pos >= per) pos = per - 1;
                col[(size_t)t * n + i] = oc * per + pos;
            }
        }
        int* dcol; double *dval, *du, *db;
        CK(hipMalloc(&dcol, col.size() * 4)); CK(hipMalloc(&dval, val.size() * 8)); CK(hipMalloc(&du, n * 8)); CK(hipMalloc(&db, n * 8));
        CK(hipMemcpy(dcol, col.data(), col.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dval, val.data(), val.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice)); CK(hipMemset(du, 0, n * 8));
        for (int nx : {8, 4, 2, 1}) {
            const int sweeps = 16, nb = (per + 255) / 256;
            const int grid = nx == 8 ? nb : ((nb + nx - 1) / nx) * 8;
            CK(hipMemset(du, 0, n * 8));
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            for (int s = 0; s < sweeps; s++)
                for (int c = 0; c < ncol; c++) {
                    const int r0 = c * per, r1 = (c + 1) * per;
                    if (nx == 8) hipLaunchKernelGGL(k_sweep<8>, dim3(grid), dim3(256), 0, st, dcol, dval, du, db, r0, r1, n);
                    else if (nx == 4) hipLaunchKernelGGL(k_sweep<4>, dim3(grid), dim3(256), 0, st, dcol, dval, du, db, r0, r1, n);
                    else if (nx == 2) hipLaunchKernelGGL(k_sweep<2>, dim3(grid), dim3(256), 0, st, dcol, dval, du, db, r0, r1, n);
                    else hipLaunchKernelGGL(k_sweep<1>, dim3(grid), dim3(256), 0, st, dcol, dval, du, db, r0, r1, n);
                }
            CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0, st); for (int r = 0; r < 20; r++) hipGraphLaunch(ge, st); hipEventRecord(e1, st); CK(hipStreamSynchronize(st));
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double chk = 0; std::vector<double> hu(n); CK(hipMemcpy(hu.data(), du, n * 8, hipMemcpyDeviceToHost)); for (double x : hu) chk += x;
            printf("n=%7d rows (%4d workgroups per launch) on %d XCD(s): %.2f us per launch   (checksum %.9f)\n", n, nb, nx, ms * 1000.0 / (20 * sweeps * ncol), chk);
            hipGraphExecDestroy(ge); hipGraphDestroy(g);
        }
        {   // one XCD, 1024-thread workgroups: 4x fewer workgroups to dispatch
            const int sweeps = 16, nb = (per + 1023) / 1024, grid = nb * 8;
            CK(hipMemset(du, 0, n * 8));
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            for (int s = 0; s < sweeps; s++)
                for (int c = 0; c < ncol; c++) hipLaunchKernelGGL((k_sweep<1, 1024>), dim3(grid), dim3(1024), 0, st, dcol, dval, du, db, c * per, (c + 1) * per, n);
            CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0, st); for (int r = 0; r < 20; r++) hipGraphLaunch(ge, st); hipEventRecord(e1, st); CK(hipStreamSynchronize(st));
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double chk = 0; std::vector<double> hu(n); CK(hipMemcpy(hu.data(), du, n * 8, hipMemcpyDeviceToHost)); for (double x : hu) chk += x;
            printf("n=%7d rows (%4d workgroups of 1024 per launch) on 1 XCD: %.2f us per launch   (checksum %.9f)\n", n, nb, ms * 1000.0 / (20 * sweeps * ncol), chk);
            hipGraphExecDestroy(ge); hipGraphDestroy(g);
        }
        hipFree(dcol); hipFree(dval); hipFree(du); hipFree(db);
    }
    return 0;
}
