// examples/05_mean_curvature_flow_sharded.cpp -- the reference's 05_example_mean_curvature_flow/main.cpp:57-79 with its right-hand
// side columns SHARDED over the GPUs of one node (BASELINE config 4): one process per GPU, the hierarchy
// replicated, rank g owns columns [g k / N, (g+1) k / N) of RHS / U, and the only communication is the all-reduce of the residual sum
// of squares, because the reference's stopping test is one Frobenius norm over all columns (src/min_quad_with_fixed_mg.cpp:110).
// The loop itself runs inside libsmg (smg_solve_sharded); this file contributes the closure: ncclAllReduce (RCCL over xGMI) on a
// communicator it created, enqueued on the solve's own stream -- no host round trip per iteration.
//
// The reference's example has k = 3 coordinate columns; to have something to shard this one solves for k = 3 + extra columns per step
// (the extra ones: M times fixed smooth functions of the rest positions, as BASELINE config 4's 64 columns do).
//
//   launch (N processes, one per GPU):   RANK=g WORLD_SIZE=N SMG_NCCL_ID_FILE=/tmp/smg_id ./05_mean_curvature_flow_sharded mesh.smgm [steps] [k]
//   single process:                      ./05_mean_curvature_flow_sharded mesh.smgm [steps] [k]
//   ranks SHARING a GPU (RCCL refuses that; the GPU test suite has one device): SMG_HOST_COMM_FILE=/tmp/smg_comm instead of SMG_NCCL_ID_FILE --
//   the same closure slot then holds a host reduction through a memory-mapped file (the sums cross the host: what dist.HostReduce does
//   for the Python callers), so smg_solve_sharded's loop runs with world size > 1 from C++ on one device.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../surface_multigrid_code_amd/csrc/mg_api.hpp"

struct Reducer { ncclComm_t comm; long calls; };
// smg_reduce_fn: in-place sum over the ranks of `count` device doubles, on the solve's stream
static int rccl_reduce(double* d, int count, void* hip_stream, void* ctx)
{
    Reducer* r = (Reducer*)ctx;
    r->calls++;
    return ncclAllReduce(d, d, (size_t)count, ncclDouble, ncclSum, r->comm, (hipStream_t)hip_stream) == ncclSuccess ? 0 : 1;
}

// ---- host communicator for ranks that share a device: a memory-mapped file, a generation barrier, sums in rank order (the same bits on every rank)
struct HostComm {
    static constexpr int MAXR = 16, SLOT = 64;
    static constexpr size_t BCAST = (size_t)1 << 20;                    // doubles
    struct Shared { std::atomic<int> count, gen; double slot[MAXR][SLOT]; double bcast[BCAST]; };
    Shared* sh = nullptr;
    int rank = 0, world = 1;
    long calls = 0;
    bool open(const char* path, int r, int w)
    {
        rank = r; world = w;
        if (w > MAXR) return false;
        const int fd = ::open(path, O_RDWR | O_CREAT, 0600);
        if (fd < 0) return false;
        if (ftruncate(fd, (off_t)sizeof(Shared)) != 0) { ::close(fd); return false; }     // a fresh file is all zeros: count = gen = 0
        void* p = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        ::close(fd);
        if (p == MAP_FAILED) return false;
        sh = (Shared*)p;
        return true;
    }
    void barrier()
    {
        const int g = sh->gen.load(std::memory_order_acquire);
        if (sh->count.fetch_add(1, std::memory_order_acq_rel) + 1 == world) { sh->count.store(0, std::memory_order_relaxed); sh->gen.fetch_add(1, std::memory_order_acq_rel); }
        else while (sh->gen.load(std::memory_order_acquire) == g) std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    bool allreduce(double* v, int count)
    {
        if (count > SLOT) return false;
        for (int i = 0; i < count; i++) sh->slot[rank][i] = v[i];
        barrier();
        for (int i = 0; i < count; i++) { double t = 0.0; for (int r = 0; r < world; r++) t += sh->slot[r][i]; v[i] = t; }
        barrier();
        return true;
    }
    bool broadcast(double* buf, size_t n, int owner)
    {
        if (n > BCAST) return false;
        if (rank == owner) std::memcpy(sh->bcast, buf, n * sizeof(double));
        barrier();
        if (rank != owner) std::memcpy(buf, sh->bcast, n * sizeof(double));
        barrier();
        return true;
    }
};
// smg_reduce_fn through the host: wait for the solve stream, sum on the host, put the result back (dist.HostReduce's C++ twin)
static int host_reduce(double* d, int count, void* hip_stream, void* ctx)
{
    HostComm* c = (HostComm*)ctx;
    c->calls++;
    double v[HostComm::SLOT];
    if (count > HostComm::SLOT || hipStreamSynchronize((hipStream_t)hip_stream) != hipSuccess) return 1;
    if (hipMemcpy(v, d, (size_t)count * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess || !c->allreduce(v, count)) return 1;
    return hipMemcpy(d, v, (size_t)count * sizeof(double), hipMemcpyHostToDevice) == hipSuccess ? 0 : 1;
}

// the unique id travels through a file: rank 0 writes it (to a temporary name, then renames), the others wait for it
static bool exchange_id(ncclUniqueId& id, int rank, const char* path)
{
    if (rank == 0) {
        if (ncclGetUniqueId(&id) != ncclSuccess) return false;
        std::string tmp = std::string(path) + ".tmp";
        FILE* f = std::fopen(tmp.c_str(), "wb");
        if (!f) return false;
        const bool ok = std::fwrite(&id, sizeof(id), 1, f) == 1;
        std::fclose(f);
        return ok && std::rename(tmp.c_str(), path) == 0;
    }
    for (int tries = 0; tries < 600; tries++) {
        FILE* f = std::fopen(path, "rb");
        if (f) { const bool ok = std::fread(&id, sizeof(id), 1, f) == 1; std::fclose(f); if (ok) return true; }
        std::this_thread::sleep_for(std::chrono::milliseconds(100));
    }
    return false;
}

int main(int argc, char* argv[])
{
    const char* path = argc > 1 ? argv[1] : "tests/golden/meshes/ogre_sim.smgm";
    const int steps = argc > 2 ? std::atoi(argv[2]) : 2;
    const int k = argc > 3 ? std::atoi(argv[3]) : 8;
    const int rank = std::getenv("RANK") ? std::atoi(std::getenv("RANK")) : 0;
    const int world = std::getenv("WORLD_SIZE") ? std::atoi(std::getenv("WORLD_SIZE")) : 1;
    const int local = std::getenv("LOCAL_RANK") ? std::atoi(std::getenv("LOCAL_RANK")) : rank;
    if (k < 3 || world < 1 || rank < 0 || rank >= world) { std::fprintf(stderr, "bad k / RANK / WORLD_SIZE\n"); return 1; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { std::fprintf(stderr, "no HIP device\n"); return 1; }
    if (hipSetDevice(local % ndev) != hipSuccess) return 1;      // libsmg keeps a handle on the device that is current at its first use

    ncclUniqueId id;
    const char* id_file = std::getenv("SMG_NCCL_ID_FILE");
    const char* host_file = std::getenv("SMG_HOST_COMM_FILE");
    HostComm hc;
    Reducer red{nullptr, 0};
    if (host_file) {
        if (!hc.open(host_file, rank, world)) { std::fprintf(stderr, "rank %d: cannot map %s\n", rank, host_file); return 1; }
    } else {
        if (world > 1) { if (!id_file || !exchange_id(id, rank, id_file)) { std::fprintf(stderr, "rank %d: no unique id (SMG_NCCL_ID_FILE)\n", rank); return 1; } }
        else if (ncclGetUniqueId(&id) != ncclSuccess) return 1;
        if (ncclCommInitRank(&red.comm, world, id, rank) != ncclSuccess) { std::fprintf(stderr, "rank %d: ncclCommInitRank failed\n", rank); return 1; }
    }

    double* Vp = nullptr; int* Fp = nullptr; int nV = 0, nF = 0;
    if (smg_mesh_read(path, &Vp, &nV, &Fp, &nF) != SMG_OK) { std::fprintf(stderr, "%s\n", smg_last_error()); return 1; }
    smg_mesh_normalize_unit_area(Vp, nV, Fp, nF);
    smgDense V(nV, 3); smgDenseI F(nF, 3);
    for (int i = 0; i < nV; i++) for (int c = 0; c < 3; c++) V(i, c) = Vp[3 * i + c];
    for (int i = 0; i < nF; i++) for (int c = 0; c < 3; c++) F(i, c) = Fp[3 * i + c];
    std::vector<mg_data> mg;
    mg_precompute(V, F, 0.25f, 100, 1, mg);                       // every rank builds the same hierarchy (replicated)
    smgSparse L;
    int nnz = 0;
    smg_mesh_cotmatrix(Vp, nV, Fp, nF, &nnz, nullptr, nullptr, nullptr);
    L.rows = L.cols = nV; L.outer.resize(nV + 1); L.inner.resize(nnz); L.values.resize(nnz);
    smg_mesh_cotmatrix(Vp, nV, Fp, nF, nullptr, L.outer.data(), L.inner.data(), L.values.data());

    // all k columns of the state: 3 coordinates + smooth functions of the rest position (every rank can form any column)
    auto column = [&](const smgDense& U3, int c, int i) {
        if (c < 3) return U3(i, c);
        const int q = c - 3;
        return std::sin(0.7 * (q + 1) * V(i, q % 3)) + 0.25 * V(i, (q + 1) % 3);
    };
    const int lo = (int)((long)rank * k / world), hi = (int)((long)(rank + 1) * k / world), kl = hi - lo;   // dist.column_range
    const double delta = 0.01, mg_tol = 5e-7;
    smgDense U = V;                                               // the three coordinate columns, known to every rank
    std::vector<double> Urow((size_t)nV * 3), M(nV);
    min_quad_with_fixed_mg_data solverData;
    smgCoarseSolver coarseSolver;
    coarseSolver.reduce = host_file ? host_reduce : rccl_reduce;  // <- the whole multi-GPU hook
    coarseSolver.reduce_ctx = host_file ? (void*)&hc : (void*)&red;
    for (int s = 0; s < steps; s++) {
        for (int i = 0; i < nV; i++) for (int c = 0; c < 3; c++) Urow[3 * (size_t)i + c] = U(i, c);
        smg_mesh_massmatrix(Urow.data(), nV, Fp, nF, /*voronoi=*/0, M.data());
        smgSparse LHS = L;                                                              // LHS = M - delta * L
        for (int i = 0; i < nV; i++)
            for (int p = LHS.outer[i]; p < LHS.outer[i + 1]; p++) {
                const double x = delta * L.values[p];
                LHS.values[p] = (LHS.inner[p] == i) ? M[i] - x : -x;
            }
        smgDense RHS(nV, kl), Z0(nV, kl), Zl;                                           // this rank's columns only
        for (int c = 0; c < kl; c++) for (int i = 0; i < nV; i++) { Z0(i, c) = column(U, lo + c, i); RHS(i, c) = M[i] * Z0(i, c); }
        min_quad_with_fixed_mg_precompute(LHS, solverData, mg, coarseSolver);
        std::vector<double> rHis;
        const bool ok = min_quad_with_fixed_mg_solve(solverData, RHS, Z0, coarseSolver, mg_tol, mg, Zl, rHis);
        // the coordinate columns come back to every rank (they define the next step's matrix): a broadcast per column from its owner
        std::vector<double> buf((size_t)nV);
        double* dbuf = nullptr;
        if (hipMalloc((void**)&dbuf, (size_t)nV * sizeof(double)) != hipSuccess) return 1;
        for (int c = 0; c < 3; c++) {
            const int owner = (int)(((long)(c + 1) * world - 1) / k);                 // the rank whose range holds column c
            if (owner == rank) for (int i = 0; i < nV; i++) buf[(size_t)i] = Zl(i, c - lo);
            if (host_file) { if (!hc.broadcast(buf.data(), (size_t)nV, owner)) return 1; }
            else {
                (void)hipMemcpy(dbuf, buf.data(), (size_t)nV * sizeof(double), hipMemcpyHostToDevice);
                if (ncclBroadcast(dbuf, dbuf, (size_t)nV, ncclDouble, owner, red.comm, nullptr) != ncclSuccess) return 1;
                (void)hipDeviceSynchronize();
                (void)hipMemcpy(buf.data(), dbuf, (size_t)nV * sizeof(double), hipMemcpyDeviceToHost);
            }
            for (int i = 0; i < nV; i++) U(i, c) = buf[(size_t)i];
        }
        (void)hipFree(dbuf);
        for (int i = 0; i < nV; i++) for (int c = 0; c < 3; c++) Urow[3 * (size_t)i + c] = U(i, c);
        smg_mesh_normalize_unit_area(Urow.data(), nV, Fp, nF);
        for (int i = 0; i < nV; i++) for (int c = 0; c < 3; c++) U(i, c) = Urow[3 * (size_t)i + c];
        double s2 = 0; for (double v : U.data) s2 += v * v;
        if (rank == 0)
            std::printf("step %d: %d columns on %d rank(s) (this rank: %d), converged %d in %d iterations, last residual %.6e, |U|^2 = %.15g\n", s, k, world, kl, (int)ok,
                        (int)rHis.size(), rHis.empty() ? 0.0 : rHis.back(), s2);
        if (!ok) return 2;
    }
    if (rank == 0) std::printf("reductions issued by rank 0: %ld (%s)\n", host_file ? hc.calls : red.calls, host_file ? "host closure" : "RCCL closure");
    if (!host_file) ncclCommDestroy(red.comm);
    smg_free(Vp); smg_free(Fp);
    return 0;
}
