// Multigrid hierarchy for the potential block + field-split preconditioner (see amg.hip).
#pragma once
#include <vector>

#include "fedm_internal.hpp"

namespace fedm {

// Scalar sliced-ELL, 64 lanes per slice.  Long rows of small matrices (restrictions, coarse
// operators) are split over `split` = 2^log2_split adjacent lanes (entry e of row r belongs to
// lane-row r*split + e % split) and summed with lane shuffles: more waves, shorter dependent
// gather chains -- the coarse levels are latency-bound, not bandwidth-bound.
struct EllMat {
    int n_rows = 0, n_rows_p = 0, n_cols = 0, n_slices = 0, log2_split = 0;
    int width = 0;  // > 0: every slice stores exactly `width` entries per lane (boff unused)
    int64_t total_bc = 0;
    int *boff = nullptr, *col = nullptr;
    double *val = nullptr, *dinv = nullptr;
    // single = true (set before from_csr): the values are stored in single precision (val32; val stays
    // null) -- the multigrid hierarchy is a preconditioner, a third of its bytes are not needed;
    // vectors, the diagonal's reciprocals and all sums stay double
    bool single = false;
    float *val32 = nullptr;
    int from_csr(const fedm_csr &m, bool want_dinv, int log2_split = -1);  // -1: choose
    void release();
};

struct Amg {
    struct Level {
        EllMat A, P, R;
        double *x = nullptr, *x2 = nullptr, *b = nullptr, *r = nullptr;
        // Composite level (V(1,1), levels whose kernels are bound by launch latency, not bytes):
        // with x = w Dinv b after the first sweep the whole leg down is one product
        //   b_c = C b,            C  = R (I - w A Dinv),
        // and prolongation + second sweep is one product on the concatenated vector [b ; x_c]
        //   x   = G b + Q x_c,    G  = w Dinv (2 I - w A Dinv),   Q = (I - w Dinv A) P
        // (built on the host at set-up, capi.cpp).  Two kernels per level instead of four.  b holds
        // n_rows_p + (rows of the next level) entries and the next level's x is its tail.
        EllMat C, GQ;
        bool composite = false;
        // finest level of a single-GPU hierarchy: only the leg down is composite (b_c = C b, 30 MB
        // instead of the 55 MB of first sweep + residual + restriction); the leg up keeps
        // prolongation and sweep, the prolongation kernel forming x = w Dinv b + P x_c itself
        bool down_composite = false;
        bool x_is_alias = false;   // x points into the previous level's b
        // Polynomial smoother (Amg::poly): k Richardson sweeps x += w_i Dinv (b - A x), weights w in
        // the order of the pre-smoother (the post-smoother runs them backwards: a symmetric cycle).
        // From a zero guess the pre-smoother is one product x = S b, S = (I - prod(I - w_i Dinv A)) Ainv
        // (sparsity of A^(k-1)); composite levels fold S into C and GQ, the finest level streams
        // [S | P] (on [b ; x_c]) and k sweeps of A.
        EllMat S;
        std::vector<double> w;
    };
    std::vector<Level> levels;
    double *coarse_inv = nullptr;  // dense inverse of the coarsest operator, rows padded to ld
    int n_coarse = 0, coarse_ld = 0;
    int nu = 2;
    bool pre_smooth = true;  // false: V(0,nu) cycles (restrict the right-hand side directly)
    // when set (while a Krylov step is being captured): the last level-0 sweep writes
    // out[r * out_stride + out_offset] instead of levels[0].x
    double *out = nullptr;
    int out_stride = 1, out_offset = 0;
    double omega = 0.67;
    bool poly = false;  // Level::w / Level::S in force instead of nu sweeps with one omega
    hipGraphExec_t graph_exec = nullptr;
    // Across GPUs only the finest level is rank-local (smoothed as a distributed operator, with
    // halo exchanges); below it the hierarchy is GLOBAL and replicated on every rank: the rank's
    // level-1 right-hand side goes to its segment [g_offset, g_offset + n_coarse) of d_gb
    // (= global->levels[0].b), d_gb is all-reduced, every rank runs the global cycle, and the
    // rank takes its segment of the result.  Aggregates never cross rank boundaries; everything
    // else is the single-GPU cycle.
    int n_global = 0, g_offset = 0;
    // slices of the finest operator without / with ghost columns (several GPUs): the two scalar
    // halo exchanges of a cycle overlap with the interior ones
    int *d_interior0 = nullptr, *d_boundary0 = nullptr;
    int n_interior0 = 0, n_boundary0 = 0;
    double *d_gb = nullptr;   // alias, owned by `global`
    Amg *global = nullptr;
    // levels[level].b -> levels[level].x; phase 0 whole cycle, 1 down leg, 2 coarse solve + up leg
    void vcycle(Ctx &c, int level, int phase = 0);
    int capture(Ctx &c);             // record the V-cycle once as a hipGraph
    void run(Ctx &c);
    void allreduce_level1(Ctx &c);   // several GPUs: between the cycle's phases 1 and 2
    void release();
};

// y = A x (mode 0), b - A x (1), Jacobi sweep (2), y += A x (3), fused first sweep + residual (4),
// y = w Dinv b + A x (5)
void ell_apply(Ctx &c, const EllMat &A, int mode, const double *x, const double *b, double *y,
               double omega, double *aux = nullptr);
void fieldsplit_setup(Ctx &c);  // species sub-block inverses into c.d_dinv
int fieldsplit_planes_check(Ctx &c, double *out, bool last_fused);   // test hook: planes vs the separate pass
unsigned fieldsplit_zero_species_planes(const Ctx &c);   // the structurally zero species planes, bit r * NS + c
// z = alpha*Minv t.  scatter = false: Amg::out places the potential; with_cycle = false: stop
// after the coupling product (the caller runs the V-cycle, whose right-hand side is ready)
void fieldsplit_apply(Ctx &c, Amg &amg, const double *t, double *z, double alpha, bool scatter = true,
                      bool with_cycle = true);
bool fieldsplit_upper(const Ctx &c);  // upper-triangular order in force (Ctx::fs_upper, split on the right)
void fieldsplit_upper_potential(Ctx &c, Amg &amg, const double *t, double alpha);
void fieldsplit_upper_species(Ctx &c, Amg &amg, const double *t, double *z, double alpha);
// first stage alone, for the vector t: g = Duu^-1 t_u into the sweeps' start vector, b0 = t_phi (Ctx::fs_first_by_producer)
void fieldsplit_first_stage(Ctx &c, Amg &amg, const double *t);
void fieldsplit_scatter(Ctx &c, Amg &amg, double *z);  // potential component of z <- the V-cycle's result
// z = Minv (J v); scatter = false leaves the potential component in amg.levels[0].x
void fieldsplit_apply_operator(Ctx &c, Amg &amg, const double *v, double *t, double *z, bool scatter);
// the same in two parts for the halo overlap: part 0 = SpMV of `slices` only (interior rows),
// part 1 = SpMV of `slices` (boundary rows) followed by the rest of the preconditioner,
// part 2 = the same up to the potential block's right-hand side (the caller runs the V-cycle)
void fieldsplit_apply_operator_part(Ctx &c, Amg &amg, const double *v, double *t, double *z, bool scatter,
                                    int part, const int *slices, int n_slices);
void poisson_precondition(Ctx &c, Amg &amg, const double *r, double *z);
// fs_tiles.hip: the n_sweeps species sweeps after the first stage in as few launches as the tiles' layers allow
// (one GPU).  false: not applicable to this context, nothing was launched.
bool fs_tiles_sweeps(Ctx &c, int n_sweeps, unsigned zmask, const float *g32, float *ping0, float *ping1, double *z,
                     const double *x0, const float *cpl32, double *b0);
int fs_tiles_info(Ctx &c, long long *out);   // 1 + out[10] = {tiles, slices per tile, layers, row width, max vertices, max rows, bytes, threads, all rows, all vertices}
void fs_tiles_release(Ctx &c);
int fs_tiles_host_stats(const fedm_mesh_desc &mesh, int tile_slices, int depth, long long *out);   // no GPU needed
// the finest multigrid level's sweeps behind the polynomial smoother's product as one launch on the same tiles (fs_tiles.hip)
bool mg_tiles_sweeps(Ctx &c, const EllMat &A, const double *b, const double *xin, int n, const double *w, double *out,
                     int ostride, int ooff);
void fs_tiles_prepare(Ctx &c);   // with fieldsplit_setup: the tiles exist before a Krylov step is captured
void fs_tiles_configure(Ctx &c, int mode, int tile_slices, int depth, int threads);

}  // namespace fedm
