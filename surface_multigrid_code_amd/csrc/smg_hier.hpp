// smg_hier.hpp -- the hierarchy handle behind the C ABI (include/smg.h): host mirror of the reference's
// std::vector<mg_data> + min_quad_with_fixed_mg_data + coarse solver, plus their device images.
#pragma once
#include <memory>
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <cstdint>
#include <string>
#include <vector>

#include "smg_bsr3.hpp"
#include "smg_coarse.hpp"
#include "smg_device.hpp"
#include "smg_mesh.hpp"
#include "smg_order.hpp"
#include "smg_sparse.hpp"
#include "smg_tiled.hpp"
#include "smg_bgs.hpp"
#include "smg_wgs.hpp"
#include "smg_schur.hpp"

namespace smg {

// bytes of device memory currently held by all DevBufs of the process (smg_device_bytes_live(): memory budget reporting)
inline std::atomic<long long>& devbuf_live_bytes() { static std::atomic<long long> v{0}; return v; }

// page-locked host memory (the small transfers of a solve: control block, residual history, the vectors of a small mesh -- a copy to or from
// pageable memory goes through the runtime's own staging and costs tens of microseconds per call, a good part of a 0.4 ms solve)
template <typename T>
struct PinBuf {
    T* p = nullptr;
    size_t n = 0;
    PinBuf() = default;
    PinBuf(const PinBuf&) = delete;
    PinBuf& operator=(const PinBuf&) = delete;
    ~PinBuf() { release(); }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; n = 0; }
    hipError_t ensure(size_t count)
    {
        if (count <= n) return hipSuccess;
        release();
        hipError_t e = hipHostMalloc((void**)&p, count * sizeof(T), hipHostMallocDefault);
        if (e == hipSuccess) n = count; else p = nullptr;
        return e;
    }
};

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept { if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; } return *this; }
    ~DevBuf() { release(); }
    void release() { if (p) { (void)hipFree(p); devbuf_live_bytes() -= (long long)(n * sizeof(T)); } p = nullptr; n = 0; }
    hipError_t alloc(size_t count)
    {
        release();
        if (count == 0) return hipSuccess;
        hipError_t e = hipMalloc((void**)&p, count * sizeof(T));
        if (e == hipSuccess) { n = count; devbuf_live_bytes() += (long long)(count * sizeof(T)); } else p = nullptr;
        return e;
    }
    hipError_t ensure(size_t count) { return count <= n ? hipSuccess : alloc(count); }
    template <class Alloc>
    hipError_t upload(const std::vector<T, Alloc>& v)
    {
        hipError_t e = alloc(v.size());
        if (e != hipSuccess || v.empty()) return e;
        return hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    }
    // upload that keeps the allocation (and with it every pointer captured in a hipGraph) when the size is unchanged; *moved is set
    // when the buffer had to be reallocated -- the caller then owes a drop_graphs()
    template <class Alloc>
    hipError_t upload_in_place(const std::vector<T, Alloc>& v, bool* moved)
    {
        if (p && v.size() == n) return v.empty() ? hipSuccess : hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
        *moved = true;
        return upload(v);
    }
};

struct SellBuf {  // device image of one SELL matrix
    DevBuf<int> slice_row, slice_off, slice_w, col, order;
    DevBuf<double> val;
    SellDev view;
    std::vector<int> color_slice_ptr;
    DevBuf<int> diag_slot;                   // coloured square matrices: slot of a_ii in the value array, per row ...
    int n_first = 0;                         // ... rows of the first colour (they lead the numbering; 0: not available) and
    int n_all = 0;                           // ... all rows (0: some row has no stored diagonal), see FirstColour in smg_device.hpp
    long stored = 0, padded = 0, used = 0;   // CSR entries / allocated slots / slots the kernels read
    hipError_t upload(const Sell& S);          // S.col / S.val empty: the panels are allocated only (filled on the device, launch_sell_fill)
    // long rows kept out of the panels (SellDev::long_*), see csrc/smg_device.hpp
    DevBuf<int> long_row, long_ptr, long_col;
    DevBuf<double> long_val;
    DevBuf<float> long_valf;
    hipError_t upload_long(const std::vector<int>& rows, const std::vector<int>& ptr, const std::vector<int>& col, const std::vector<double>& val);
};

struct Bsr3Buf {  // device image of one block (3 x 3) SELL matrix, smg_bsr3.hpp
    DevBuf<int> slice_row, slice_off, slice_w, col, order;
    DevBuf<double> val;
    DevBuf<float> valf;                         // fp32 image of val (mixed-precision cycle), made on demand
    Bsr3Dev view;
    std::vector<int> color_slice_ptr;
    long stored = 0, blocks = 0, padded = 0;   // scalar CSR entries / 3 x 3 blocks / allocated value slots (9 per panel slot)
    hipError_t upload(const Bsr3Sell& S);
};

struct TiledBuf {  // device image of one overlapped-tiling plan (smg_tiled.hpp)
    DevBuf<int> hdr, ext_rows, pcol, prow, map, mapd;   // map / mapd: value slot / diagonal of a panel row -> index into Level::d_Aval (value-only re-precompute), -1: leave
    DevBuf<double> pval, pdiag;
    TiledDev view;
    long updates = 0;
    std::vector<int> host_map, host_mapd;   // the maps before their first use (uploaded by the first value-only re-precompute)
    bool tried = false;      // a plan was attempted for this (level, sweeps): empty view = the level does not qualify
};

struct BgsBuf {  // device image of the block-sequential Gauss-Seidel plan of a level (smg_bgs.hpp)
    DevBuf<int> hdr, xrow, ugrow, ulrow, eidx, map, mapd;   // map / mapd: value slot / diagonal slot -> index into Level::d_Aval (value-only re-precompute), -1 padding
    DevBuf<double> eval, udiag;
    BgsDev view;
    std::vector<int> color_ptr;      // blocks of colour c
    std::vector<int> host_map, host_mapd;          // the maps before their first use (uploaded by the first value-only re-precompute)
    std::vector<int> host_rows, host_blk_ptr;      // the bgs order (position -> internal row), positions per block: introspection, tests
    double rim = 0.0, fill = 0.0;
    bool tried = false;
};

struct WgsBuf {  // device image of the wave Gauss-Seidel plan of a level (smg_wgs.hpp)
    DevBuf<int> hdr, grow, meta, rim, map, mapd;   // map / mapd: value slot / diagonal slot -> index into Level::d_Aval (value-only re-precompute), -1 padding
    DevBuf<unsigned> eoff;
    DevBuf<double> eval, diag;
    WgsDev view;
    std::vector<int> color_ptr;      // pieces of colour c
    std::vector<int> host_map, host_mapd;          // the maps before their first use (uploaded by the first value-only re-precompute)
    std::vector<int> host_rows, host_piece_ptr;    // the wgs order (position -> internal row), positions per piece: introspection, tests
    double rim_ratio = 0.0, phases_mean = 0.0;
    int phases_max = 0;
    bool tried = false;
};

// one element of std::vector<mg_data> (reference src/mg_data.h:11-27)
struct Level {
    // ---- host, caller numbering: the mg_data fields ----
    std::vector<double> V;  // mg_data::V (optional)
    std::vector<int> F;     // mg_data::F (optional)
    Csr P_full;             // mg_data::P_full
    std::shared_ptr<DecimationLog> dec_log;   // collapses of the step that built this level (smg_mg_precompute_logged), else null
    Csr A;                  // mg_data::A   (unknown-only system matrix of the level)
    std::vector<double> A_diag;  // mg_data::A_diag
    Csr P, PT;              // mg_data::P / PT (unknown-only, maps level lv -> lv-1 / back)
    // ---- device numbering ----
    Ordering ord;           // colour-major numbering of this level's unknowns
    uint64_t ord_key = 0;   // hash of the sparsity pattern `ord` was built for (time-stepping callers re-precompute
                            // with the same pattern every step: the colouring is reused)
    Csr A_int, P_int, PT_int;  // host copies in the internal numbering (introspection / tests)
    // ---- block (3-DOF) hierarchies (smg_hierarchy::bs == 3): the numbering is a VERTEX colouring, ord = its expansion to DOFs 3v+d;
    //      A lives in 3 x 3 blocks (bA / bAT instead of dA / dAT), dP / dPT hold the vertex-level factor of P (x) I_3 ----
    Ordering vord;          // colour-major numbering of the level's vertices
    Csr Pv, PTv;            // vertex-level factor of P = Pv (x) I_3 and its transpose, caller numbering
    Bsr3Buf bA, bAT;
    DevBuf<int> mapB, mapBT;   // value slot of bA / bAT -> index into d_Aval (-1: explicit zero / padding); value-only re-precompute
    SellBuf dA, dP, dPT;
    SellBuf dAT;            // SELL image of A^T, only when A is not bitwise symmetric (Galerkin levels)
    bool gs_on_transpose = false;  // the reference's GS walks COLUMN i of A (src/mg_VCycle.cpp:149-155)
    bool P_device_filled = false, PT_device_filled = false;   // likewise dP / dPT (P_int / PT_int on demand: ensure_P_int)
    bool device_filled = false;    // dA was filled on the device from A and the permutation: A_int is built on demand (ensure_A_int)
    Csr vpat;               // block hierarchies: the n_v x n_v pattern of the 3 x 3 blocks of A (caller's vertex numbering; values unused)
    bool A_bit_symmetric = false;  // A == A^T bit for bit (checked on the host half when the device fill is a candidate)
    TiledBuf tiled[4];      // overlapped-tiling plans of relax(sweeps), sweeps = 1 .. 3 (index = sweeps; built on demand in ensure_work)
    WgsBuf wgs;             // wave Gauss-Seidel plan of a Galerkin level of a decimated hierarchy, 1 - 8 columns (built on demand in ensure_work)
    BgsBuf bgs;             // block-sequential Gauss-Seidel plan for solves with a multiple of 16 columns (BGS_COLS) (built on demand in ensure_work)
    // ---- value-only re-precompute (fixed sparsity, csrc/smg_capi.cpp: fast path of smg_precompute) ----
    std::vector<int> A_int_src;   // A_int entry -> index into A.val
    DevBuf<double> d_Aval;        // values of A in the caller's CSR order: the canonical device copy
    DevBuf<double> d_Tval;        // stage-1 temporary  PT * A_{lv-1}
    DevBuf<int> mapA, mapAT;      // SELL slot -> index into d_Aval (-1 = padding)
    DevBuf<int> r1_ptr, r1_idx, r2_ptr, r2_idx;
    DevBuf<double> r1_coef, r2_coef;
    int nnzT = 0;
    // ---- fp32 images for the mixed-precision V-cycle (values only; slots, columns, slice tables are shared) ----
    DevBuf<float> a32, at32, p32, pt32;
    SellDev dA32, dAT32, dP32, dPT32;
    DevBuf<float> b32, u32, r32;
    // ---- work vectors, internal layout n x kcap ----
    DevBuf<double> b, u, r;
    DevBuf<double> t;       // second iterate of a Jacobi-smoothed level (the sweeps ping-pong between u and t); allocated on demand
    DevBuf<float> t32;
    DevBuf<double> d;       // update vector of a Chebyshev-Jacobi-smoothed level; allocated on demand
    DevBuf<float> d32;
    double lam = 0.0;       // Gershgorin bound of the spectrum of D^-1 A (D^-1 A^T where the smoother streams A^T), device numbering
    int n = 0;
};

// Everything a captured outer iteration bakes in besides device pointers: a cached graph is reused only while the handle's current
// selection compares equal to the key it was captured with.  A new mode that changes the launch sequence or a kernel argument of the
// cycle gets a field HERE (one place), not a hand-written comparison.
struct GraphKey {
    int k = 0, k_user = 0, pre = 0, post = 0, precision = 0, smoother = 0, jacobi_max_rows = 0;
    double omega = 0.0, cheby_fraction = 0.0;
    bool head_fuse = false;
    bool operator==(const GraphKey& o) const
    {
        return k == o.k && k_user == o.k_user && pre == o.pre && post == o.post && precision == o.precision && smoother == o.smoother &&
               jacobi_max_rows == o.jacobi_max_rows && omega == o.omega && cheby_fraction == o.cheby_fraction && head_fuse == o.head_fuse;
    }
    bool operator!=(const GraphKey& o) const { return !(*this == o); }
};

struct ProfScope { std::string name; long count = 0; double ms = 0.0; };
struct ProfRec { int scope; hipEvent_t e0, e1; };

}  // namespace smg

struct smg_hierarchy {
    int n_levels = 0;
    std::vector<smg::Level> lv;
    // ---- min_quad_with_fixed_mg_data (reference src/min_quad_with_fixed_mg.h:22-29) ----
    int n_full = 0;
    bool has_known = false, precomputed = false;
    std::vector<int> known, unknown;
    smg::Csr LHS_unused;  // data.LHS duplicates mg[0].A in the reference; not stored twice here
    smg::Csr Auk;
    smg::DevBuf<int> d_map0;      // internal row i of level 0 -> index in the caller's full-size vectors
    smg::DevBuf<int> d_perm0;     // internal row i of level 0 -> unknown-numbering index
    smg::DevBuf<int> d_unknown, d_known;
    smg::DevBuf<int> d_auk_ptr, d_auk_col;
    smg::DevBuf<double> d_auk_val;
    // ---- fast re-precompute bookkeeping ----
    uint64_t pre_key = 0;          // hash of (pattern of A, known list, P version) of the last full precompute
    int p_version = 0;             // bumped by smg_level_set_prolong
    bool input_canonical = false;  // the caller's CSR rows were sorted and duplicate-free (entry indices are stable)
    bool recipes_built = false, host_stale = false;
    int nnz_input = 0;
    std::vector<int> lhs_src, auk_src;      // LHS / Auk entry -> index into the caller's value array
    smg::DevBuf<int> d_lhs_src, d_auk_src, d_diag_idx;
    smg::DevBuf<long long> d_dense_pos;
    smg::DevBuf<double> d_Afull;
    // the caller's arrays of A on the device, sent beside the host half of a first precompute (transient: consumed or dropped by its device half)
    struct EarlyUpload { smg::DevBuf<int> ptr, col; smg::DevBuf<double> val; bool valid = false; } early0;
    // ---- coarse solver: stands in for Eigen::SimplicialLDLT (factorisation pre-inverted on the device) ----
    int nc = 0, nc_pad = 0;
    smg::DevBuf<double> d_Ainv;
    smg::DevBuf<float> d_Ainv32;
    smg::DevBuf<double> d_sympart;  // (nc_pad/64)^2 x 64 partial products of the symmetric k = 1 coarse solve (also used as float)
    // ... or, for coarsest levels beyond the dense range (smg_coarse.hpp): sparse Cholesky, factored on the host, solved on the device
    bool coarse_sparse = false;
    int coarse_dense_max = 16384;   // smg_hierarchy_set_coarse_dense_max
    bool coarse_dense_max_user = false;   // ... was called: above it the caller gets the sparse factorisation, not the Schur stand-in (schur_wanted)
    bool mem_lean = false;          // smg_hierarchy_set_memory_lean: compact SELL panels (slice_off table) instead of the fixed panel pitch
    int wgs_mode = -1;              // smg_hierarchy_set_wave_gs: -1 automatic (levels the colour launches serve badly: > 5 colours or rows of > 12 entries), 0 never, 1 every Gauss-Seidel level in range
    int bgs_min_rows = -1;          // smg_hierarchy_set_block_gs: levels of at least this many rows sweep block-sequentially when k % 16 == 0, k >= 16 (< 0: never, the default)
    smg::SparseChol chol;
    smg::DevBuf<int> c_perm, c_rptr, c_rcol, c_cptr, c_crow, c_err;
    smg::DevBuf<double> c_rval, c_cval, c_diag, c_work;
    smg::SparseCholDev c_view;
    // ... or, in the upper part of the dense range (smg_schur.hpp): interior blocks of <= 64 rows eliminated exactly, the separator's Schur
    // complement inverted densely -- the factorisation the time-stepping callers repeat costs (separator / n)^3 of the dense inverse's
    bool coarse_schur = false;
    int coarse_schur_when = 2;      // smg_hierarchy_set_coarse_schur: 0 never, 1 from the first smg_precompute on, 2 from the first VALUE-ONLY re-precompute on (a caller
                                    // that sends new values for an old pattern pays the factorisation at every step; one that does not, only the solves)
    int coarse_schur_min = 2048;    // ... for coarsest levels of at least this many unknowns (and within the dense range)
    int coarse_schur_big = 6144;    // under `when` = 2: coarsest levels of at least this many unknowns take it from the FIRST precompute on -- there it is also the cheaper one
                                    // to apply (15 804 unknowns: 38 us and 182 MB against 204 us and 2 GB) and to build
    int coarse_schur_max = 65536;   // ... and up to this size it stands in for the sparse factorisation above coarse_dense_max (only S, ~0.27 n rows there, is dense)
    bool schur_declined = false;    // the current coarsest matrix has no plan (smg_schur.hpp): not tried again until the next full precompute
    smg::SchurPlan schur;
    struct SchurBuf {
        smg::DevBuf<int> irow, bsize, srow, sptr, sidx, aptr, ablk, acol, rptr;
        smg::DevBuf<long long> coff, pos, pos2, ones, rdst, rdst2, rsrc;
        smg::DevBuf<double> arena, g, xs, sym, gj;
        smg::DevBuf<float> arena32, g32, xs32;
        smg::SchurDev view;
        void release() { *this = SchurBuf(); }
    } sch;
    bool f32_valid = false;
    int kcap32 = 0;
    // ---- block (3-DOF) variant: 1 = scalar kernels, 3 = the level matrices live in 3 x 3 blocks ----
    int block_mode = -1;           // smg_hierarchy_set_block_mode: -1 decide at precompute (P = Pv (x) I_3 on every level, no constraints,
                                   // blocks at least half full), 0 never, 3 required (precompute fails if the structure is not there)
    int bs = 1;                    // what the last precompute decided
    // ---- independent meshes in one handle (smg_hierarchy_create_union, csrc/smg_union.cpp) ----
    int union_m = 0;                       // members (0: an ordinary handle)
    std::vector<int> union_off0;           // m + 1: the members' row ranges in the caller's FULL numbering of level 0
    std::vector<int> union_offc;           // m + 1: their row ranges on the coarsest level (that level's caller numbering); set by the precompute
    std::vector<long long> union_moff;     // m: offset of member i's inverse in d_Ainv
    std::vector<int> union_mlda;           // m: its leading dimension
    struct UnionBuf {
        smg::DevBuf<int> rows, rptr, crow_member, mlda, mrow0, nhis, done;
        smg::DevBuf<long long> moff;
        smg::DevBuf<double> ss, his, zsave;
        smg::UnionDev view;
    } un;
    // ---- execution ----
    int device = -1;
    hipStream_t stream = nullptr;
    bool own_stream = false, user_stream = false;
    hipStream_t aux[3] = {nullptr, nullptr, nullptr};   // the precompute's image fills run side by side on these (creating a stream costs milliseconds: made once)
    smg::DevBuf<smg::Ctrl> d_ctrl;
    smg::Ctrl host_ctrl;            // staging of the control block smg_solve_begin uploads (must outlive the asynchronous copy)
    smg::DevBuf<double> d_rhis;     // residual history (Ctrl::r_his points here), at least max_iter entries
    smg::PinBuf<smg::Ctrl> pin_ctrl;   // where the host reads the control block to (read_ctrl)
    smg::PinBuf<double> pin_his;       // ... and the residual history at the end of a solve
    smg::PinBuf<double> pin_vec;       // host-buffer solves of small systems: RHS and z0 on their way in, z on its way out
    smg::DevBuf<double> d_partials;
    int kcap = 0;
    // ---- solve state ----
    bool in_solve = false;
    int k = 0;                      // columns of the internal blocks of the solve in progress (internal_cols(k_user): smg_cycle.cpp)
    int k_user = 0;                 // the caller's column count
    int coarse_cols = 0;            // > 0 during a padded solve: the dense coarse product takes these (the caller's) columns only
    double tol = 1e-3;
    int max_iter = 20, pre = 2, post = 2, verbosity = 0, check_every = 1, use_graph = 1, precision = 0;
    // ---- smoother selection (smg_hierarchy_set_smoother / smg_solve_opts): 0 GS everywhere (reference), 1 Jacobi, 2 hybrid ----
    int smoother = 0;
    double omega = 0.8;
    int jacobi_max_rows = 100000;
    double cheby_fraction = 0.1;   // Chebyshev-Jacobi: the polynomial damps the eigenvalues of D^-1 A in [fraction * lam, lam]
    smg::DevBuf<double> d_lam;     // scratch for launch_gershgorin
    bool lam_valid = false;        // Level::lam belongs to the current matrix values
    int iters_enqueued = 0;
    smg::DevBuf<double> d_stage_rhs, d_stage_z, d_stage_kv, d_tmp_cm;
    smg::DevBuf<double> d_zsave;     // iterate saved by the speculative cycle
    hipGraphExec_t g_spec = nullptr;  // save + V-cycle (no decide)
    const double* cur_kv = nullptr;  // device pointer to known_val (column-major) of the running solve
    int cur_ld_kv = 0;
    // ---- hipGraph cache (one outer iteration; and its two halves for the split-phase API) ----
    hipGraphExec_t g_iter = nullptr, g_resid = nullptr, g_cycle = nullptr;
    hipGraphExec_t g_iter_n = nullptr;    // graph_iters() outer iterations in one graph (smg_cycle.cpp: enqueue_outer_iterations)
    hipGraphExec_t g_rd = nullptr, g_cyc = nullptr;   // residual + break test | the cycle that follows it: an iteration the host looks into (enqueue_checked_iteration)
    double* g_sumsq_ptr = nullptr;   // the buffer g_resid writes / g_cycle reads (the caller's all-reduce buffer, or ctrl->sumsq)
    smg::GraphKey g_key;             // what the cached graphs were captured with (k == 0: nothing cached)
    bool head_fuse = false;          // this solve takes the outer residual out of the first sweep (latched at smg_solve_begin)
    // ---- profc mirror ----
    bool prof_on = false;
    std::vector<smg::ProfScope> scopes;
    std::vector<smg::ProfRec> recs;
    std::vector<hipEvent_t> ev_pool;
};
