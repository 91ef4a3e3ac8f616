// Internal declarations shared by the host side and the HIP kernels of libfedm_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "fedm_hip.h"

namespace fedm {

constexpr int SLICE = 64;  // vertices per matrix slice = one wavefront

// Sliced block-ELL pattern of the P1 vertex graph (host copy).
//   slice S holds vertices [64 S, 64 S + 64); its rows are padded to the widest row of
//   the slice; block column bc = slice_boff[S] + j is the j-th neighbour of every vertex
//   of the slice.  Values of block entry (cr, cc) of (bc, lane) live at
//   val[(bc * NEQ*NEQ + cr*NEQ + cc) * 64 + lane]  -> lanes are contiguous (coalesced).
// One cell as seen from one patch (slice of 64 owned vertices): local vertex ids (< 64 owned
// lane, >= 64 halo), the local block column of every (a, b) pair whose row vertex a is owned
// (0xFF otherwise), boundary tags and the global cell id.
struct PatchCell {
    int32_t cell;
    uint8_t lv[3];
    int8_t tag[3];
    uint8_t j[9];
    uint8_t pad_;
};
static_assert(sizeof(PatchCell) == 20, "PatchCell layout");

struct Pattern {
    int nv = 0, nc = 0, nvp = 0, n_slices = 0;
    int64_t total_bc = 0;    // stored block columns (x 64 lanes = stored blocks incl. padding)
    int64_t nnz_blocks = 0;  // structural blocks without padding
    std::vector<int> slice_boff;        // n_slices + 1
    std::vector<int> colidx;            // total_bc * 64, padded entries point at the row itself
    std::vector<int> row_len;           // nvp
    std::vector<uint32_t> diag_slot;    // nvp : slot of the diagonal block
    std::vector<uint32_t> cell_slots;   // nc * 9 : slot of block (a, b) of every cell
    std::vector<int> colour_ptr;        // n_colours + 1
    std::vector<int> colour_cells;      // cells grouped by colour
    // patch (= slice) lists for the LDS assembly kernel
    std::vector<int> patch_cell_ptr;    // n_slices + 1
    std::vector<PatchCell> patch_cells; // every cell touching a vertex of the slice
    std::vector<int> patch_halo_ptr;    // n_slices + 1
    std::vector<int> patch_halo;        // non-owned vertices referenced by the slice's cells
    int max_patch_width = 0, max_patch_verts = 0, max_patch_cells = 0;
    bool patch_ok = false;              // false: some patch exceeds the 8-bit local indices
};

// allow_rotation: the patch cell order may also turn cells (cyclic rotation of their local vertices) to avoid
// LDS bank clashes; not for models whose tables are indexed by (cell, local node) -- Expression sources
void build_pattern(const fedm_mesh_desc &mesh, Pattern &pat, bool allow_rotation = true);

struct Amg;
struct Comm;
struct GdPrep;
struct FsTiles;

// Optional in-run kernel timing with HIP events on the library's stream (bench.py roofline).
// kinds: 0 assembly F+J, 1 Jacobian SpMV, 2 assembly F only, 3 multigrid V-cycle
struct Prof {
    bool on = false;
    bool all_kinds = false;                        // also time the kernels inside Krylov steps
    bool recording = false;                        // between prof_begin and prof_end
    int stride[8] = {1, 4, 1, 4, 1, 1, 1, 1};      // sample every n-th launch of a kind
    long seen[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<hipEvent_t> ev;
    std::vector<int> kind;
    int used = 0;
    double ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

struct Ctx {
    Amg *amg = nullptr;  // potential-block multigrid (optional)
    // a second cycle for hard systems (stronger smoothing: fedm_amg_setup_poly with as_alternative):
    // swapped with `amg` together with the alternative set of species sweeps (fs_alt_active)
    Amg *amg_alt = nullptr;
    Prof prof;
    Comm *comm = nullptr;  // multi-GPU transport (optional)
    int n_owned = 0;       // owned vertices (== nv on a single GPU)
    // Deep halos: ghost vertices of the inner layers have assembled rows; d_identity lists the ghost
    // vertices with identity rows (the outermost layer).  halo_depth == 1: all ghosts, no list.
    int halo_depth = 1, n_identity = 0;
    int *d_identity = nullptr;
    // one GPU: the boundary facets' terms (species rows) and the Dirichlet / padding rows in ONE launch when no row
    // belongs to both (boundary_rows_disjoint, found at set-up); launch_assemble leaves the facets to launch_finalize
    bool boundary_rows_disjoint = false;
    int boundary_pending = 0;   // 1: residual, 2: residual + Jacobian
    int64_t n_dot = 0;     // vector entries that take part in reductions: n_owned * neq
    int device = 0;
    hipStream_t stream = nullptr;
    int nv = 0, nc = 0, nvp = 0, ns = 0, neq = 0;
    bool poisson = false;
    int64_t n = 0, np = 0;  // nv*neq, nvp*neq
    fedm_model_desc model{};
    int assembly_lean = 3;   // patch assembly generation where it applies: 0 unrolled element, 2 one equation
                             // row at a time (element_lean.hpp), 3 one pass over the cells (assemble3.hip)
    bool xcd_remap = true;   // patch / slice -> workgroup mapping contiguous per XCD
    // Planes (row, col) of the Jacobian that never change: potential-potential, and species planes that
    // are structurally zero.  bit row * neq + col; kept between assemblies once a full one has written them.
    uint32_t const_plane_mask = 0;
    uint32_t zero_plane_mask = 0;   // the structurally zero ones among them (the SpMV skips their bytes)
    bool skip_const_planes = true, const_planes_valid = false;
    bool halo_pending = false;  // several GPUs: ghost entries of d_u are stale (kernels.hip, flush_pending_halo)
    bool assembly_overlap = true;  // ... and the next assembly hides their exchange behind its interior patches
    int model_kind = 0;  // 0: LFA family (fedm_model_desc), 1: LMEA family (fedm_gd_desc)
    fedm_gd_desc gd{};
    fedm_gd_desc *d_gd = nullptr;
    double *d_gd_fields = nullptr;  // [n_fields][nv] nodal coefficient fields
    int gd_n_fields = 0;
    // Jacobian by element matrices + gather (gd.hip): every cell's 3 x 3 blocks, then each stored block
    // sums its contributions (inverse of cell_slots as CSR over the stored positions): each matrix
    // value is written once, coalesced, in a fixed summation order
    double *d_gd_elem = nullptr;          // [9 * neq * neq][nc]
    int *d_gd_inv_ptr = nullptr;          // [total_bc * 64 + 1]
    int *d_gd_inv_idx = nullptr;          // cell * 9 + (a * 3 + b)
    double *d_gd_elemF = nullptr;         // element residuals [3 * neq][nc]
    int *d_gd_vinv_ptr = nullptr, *d_gd_vinv_idx = nullptr;   // vertex -> cell * 3 + local vertex
    int gd_hand_mode = 5;                 // gd.hip, launch_assemble_gd
    uint32_t *d_gd_kpos = nullptr;        // (cell, a, b) -> place in the contributions sorted by matrix position
    GdPrep *gd_prep = nullptr;  // on-device per-step coefficient refresh (LMEA)
    Pattern pat;
    double dt = 1.0, dt_old = 1e30;
    // device mesh
    double *d_coords = nullptr;
    int *d_cells = nullptr;
    int8_t *d_ftags = nullptr;
    uint32_t *d_cell_slots = nullptr;
    int *d_colour_cells = nullptr;
    int *d_patch_cell_ptr = nullptr, *d_patch_halo_ptr = nullptr, *d_patch_halo = nullptr;
    PatchCell *d_patch_cells = nullptr;
    int assembly_kind = 1;  // 0: global colouring (deterministic), 1: LDS patches
    int n_bfacets = 0;      // tagged boundary facets (cell, local facet, tag)
    int *d_bfacets = nullptr;
    std::vector<int> bfacet_colour_ptr;  // facets grouped by colour
    fedm_model_desc *d_model = nullptr;
    double *d_ext[FEDM_MAX_SPECIES] = {nullptr, nullptr, nullptr, nullptr};
    // postfix programs of Expression sources evaluated on the device (fedm_ext_source_program)
    int *d_expr_ops[FEDM_MAX_SPECIES] = {nullptr, nullptr, nullptr, nullptr};
    double *d_expr_consts[FEDM_MAX_SPECIES] = {nullptr, nullptr, nullptr, nullptr};
    int expr_n_ops[FEDM_MAX_SPECIES] = {0, 0, 0, 0}, expr_n_params[FEDM_MAX_SPECIES] = {0, 0, 0, 0};
    // matrix
    int *d_slice_boff = nullptr;
    int *d_colidx = nullptr;
    uint32_t *d_diag_slot = nullptr;
    double *d_val = nullptr;
    double *d_dinv = nullptr;  // sliced: [(slice*NEQ2 + e)*64 + lane]
    // copies for the field-split preconditioner (species_planes_kernel):
    float *d_val32 = nullptr;   // coupling planes in fp32, [(bc*NS + c)*64 + lane]: the potential row's species
                                // columns J_phi,u (lower-triangular order) or the species rows' potential
                                // column J_u,phi (upper-triangular order, fs_upper)
    _Float16 *d_s16 = nullptr;  // Duu^-1 J_uu in fp16: [(bc*64 + lane)*NS*NS + r*NS + c]
    // The one-pass assembly (assemble3.hip) forms these planes from its LDS accumulators while it streams the
    // Jacobian out; planes_fused says the last Jacobian assembly did, and fieldsplit_setup then only redoes the rows
    // that were changed behind the volume kernel: the vertices of boundary-facet cells, of Dirichlet values and
    // the padding (d_planes_rows, listed once).  Opt-in (FEDM_PLANES_FUSED=1); default: the separate pass over the matrix.
    bool planes_fused = false, planes_fuse_ok = false, planes_last_fused = false;
    int *d_planes_rows = nullptr;
    int n_planes_rows = 0;
    // Dirichlet
    int n_dir = 0;
    int *d_dir_dofs = nullptr;
    double *d_dir_vals = nullptr;
    // vectors (np doubles each)
    double *d_u = nullptr, *d_uold = nullptr, *d_uold1 = nullptr, *d_F = nullptr;
    double *d_delta = nullptr, *d_w = nullptr, *d_rhs = nullptr, *d_tmp = nullptr, *d_fs = nullptr, *d_fs_g = nullptr;
    int fs_sweeps = 1;       // Richardson sweeps with block-Jacobi scaling on the species block
    double fs_w[16] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};  // 1 sweep = block Jacobi
    // the two sets of fedm_set_fieldsplit / fedm_set_fieldsplit_alternative; fs_w is the active one
    int fs_main_sweeps = 1, fs_alt_sweeps = 0;
    double fs_main_w[16] = {1}, fs_alt_w[16] = {1};
    bool fs_alt_active = false;
    // one GPU: several species sweeps per launch on tiles of slices with their vertex layers in LDS (fs_tiles.hip);
    // state 0: not looked at yet, 1: in use, -1: not applicable here
    // set for the duration of a GMRES solve (one GPU, field split on the right, sweeps): whoever completes a Krylov
    // vector has formed the preconditioner's first stage for it (g, b0); fieldsplit_apply starts at the sweeps
    bool fs_first_by_producer = false;
    FsTiles *fs_tiles = nullptr;
    int fs_tiles_state = 0;
    bool mg_tiles_off = false;   // (test hook: the multigrid's finest-level sweeps as kernels of their own)
    int fs_tiles_want_slices = 0, fs_tiles_want_depth = 0, fs_tiles_want_threads = 0;   // > 0: set by fs_tiles_configure
    bool fs_halo = true;    // several GPUs: ghost exchange of the species iterate before every sweep
    bool deep_halo = true;  // use the ghost layers (halo_depth > 1) instead of those exchanges; FEDM_DEEP_HALO=0: off
    // lower-triangular order: b_phi -= J_phi,u z_u formed inside the last species sweep from the
    // iterate before it (FEDM_FS_LAGGED_COUPLING=0: separate kernel on the final iterate)
    bool fs_lagged_coupling = true;
    bool red12_local = false;   // several GPUs: d_red[1..2] of the last Newton update are rank-local sums
    // Order of the block-triangular split when it sits on the right of the operator
    // (fedm_set_fieldsplit_order, FEDM_FS_ORDER=lower|upper; the left-preconditioned path, whose
    // first stage is the SpMV's epilogue, is always lower).
    //   lower (default): species sweeps first, potential right-hand side minus J_phi,u z_u, V-cycle.
    //          The true residual (dominated by the species rows, ~1e18 against ~1e-8 for the Poisson
    //          row) then tracks the error: an inexact potential shows up in the species rows through
    //          J_u,phi.
    //   upper: V-cycle on the potential block first, species sweeps on t_u - J_u,phi z_phi.  The
    //          species rows are satisfied for whatever the V-cycle returned, so the residual test
    //          passes after 3-4 instead of 8 Krylov steps late in a streamer run -- with the
    //          potential solved to V-cycle accuracy only.  The ERROR per Krylov step is the same for
    //          both orders (tests/studies/precond_error.py); after 220 steps the state deviates from
    //          a tightly solved run by 4e-3 instead of 6e-6 (tools/fs_order_accuracy.py).  Opt-in.
    bool fs_upper = false;
    double fs_switch_above = 5.0, fs_back_below = 3.5;
    // One GPU: in the hard regime (>= fs_switch_above Krylov steps per Newton system) the set that is
    // actually FASTER is used -- measured: wall time per Newton iteration of the solves under each set
    // (exponential average), the other set probed for two solves (the first re-captures its graphs and
    // is not counted) every fs_probe_every solves.  Which
    // set wins depends on the mesh and the phase of the run (tools/late_sweep.py).  FEDM_FS_POLICY=counts:
    // round 2's rule (the alternative set whenever the count is high); always so across GPUs, where every
    // rank must take the same decision.
    bool fs_measured_policy = true;
    double fs_cost[2] = {0.0, 0.0};     // [main, alternative] ms per Newton iteration (0: unknown)
    int fs_age[2] = {0, 0};             // solves since that set was last measured
    int fs_probe_left = 0;              // > 0: probing the other set for that many more solves
    bool fs_skip_sample = false;        // the next solve re-captures its graphs: its time does not count
    int fs_probe_every = 120;
    long fs_solves[2] = {0, 0};         // Newton solves run under the [main, alternative] set (fedm_fieldsplit_policy)
    double *d_V = nullptr;  // (restart+1) Krylov vectors
    int krylov_cap = 0;
    int krylov_steps_hint = 0;   // Krylov steps of the previous solve: how far ahead steps are queued
    // Flexible GMRES with the field split on the right: z_j = Minv v_j is kept, so the update is
    // Z y and no preconditioner application is spent on the right-hand side.
    double *d_Z = nullptr;
    bool right_precond = true;
    // reductions
    double *d_partials = nullptr;  // [RED_BLOCKS][RED_K]
    double *d_partials_wide = nullptr;  // [8][workgroups of the Jacobian product]: spmv_dots_kernel
    double *d_red = nullptr;       // [RED_K]
    double *h_mail = nullptr;      // pinned, host-mapped: MAIL_SLOTS slots of [RED_K] values + sequence tag
    double *h_red = nullptr;       // the slot of the publication last waited for (wait_red); written
                                   // by the last kernel of a reduction and polled by the host
    unsigned long long mail_seq = 0;      // publications queued so far (host count)
    unsigned long long *d_mail_seq = nullptr;  // same count on the device: replayed graphs cannot
                                               // carry a fresh tag in their kernel arguments
    // one captured hipGraph per Krylov index j: operator + preconditioner + orthogonalisation
    std::vector<hipGraphExec_t> iter_graph;
    std::vector<hipGraphExec_t> iter_graph_interior;  // several GPUs: the interior-rows SpMV of step j
    std::vector<hipGraphExec_t> iter_graph_pre;       // several GPUs, field split on the right: z_j = Minv v_j
    std::vector<hipGraphExec_t> iter_graph_pair;      // one GPU, on the right: steps j and j + 1 as ONE graph
    // the same two kinds without the update of their last step (the step expected to end the solve: its
    // normalised vector is only needed if the solve goes on, and is then formed by a launch of its own)
    std::vector<hipGraphExec_t> iter_graph_last, iter_graph_pair_last;
    bool iter_graphs_ok = true;   // false after a failed capture: plain launches from then on
    bool capturing = false;
    int newton_its_hint = -1;     // Newton iterations of the previous converged solve
    int snap_krylov_steps_hint = 0, snap_newton_its_hint = -1;   // ... as of fedm_state_snapshot
    // relative change of one component, computed with the final residual check of the last Newton solve
    // (fedm_newton_opts::watch_component); dropped by anything that touches the state
    int err_cache_comp = -1;
    double err_cache = 0.0;
    double *h_stage = nullptr;     // pinned staging, np doubles
    void *d_lean3_plan = nullptr;  // assemble3.hip: the model compiled for the one-pass kernels (Lean3Plan), device memory
    int lean3_sig = 0;              // the precompiled structure signature the plan selects (assemble3.hip; 0: none)
    void *lean3_classes = nullptr; // assemble3.hip: patch lists by LDS need (Lean3Classes), built at first use
    double *d_snapshot = nullptr;  // fedm_state_snapshot: u, u_old, u_old1 (3 np doubles, allocated on first use)
};

constexpr int RED_BLOCKS = 512;
constexpr int RED_K = 40;
constexpr int RED_SPARE = RED_K - 3;
constexpr int MAIL_SLOTS = 4;         // publications the host may have unread (GMRES keeps up to three steps in flight)  // a norm that rides along with the next Krylov publication
// per-block partial sums, one contiguous row per slot: the single workgroup that finishes a
// reduction reads a slot's RED_BLOCKS partials as 4 KB of consecutive doubles (with [block][slot]
// it gathered one double per 320-byte stride: most of reduce_finish_kernel's 10 us)
#define PARTIAL_AT(block, slot) ((size_t)(slot) * RED_BLOCKS + (size_t)(block))

// ---- kernel launchers (kernels.hip) -----------------------------------------------------
// mode: 0 = full model, 1 = Poisson row only (species rows become identity)
void launch_assemble(Ctx &c, bool jacobian, int mode);
void launch_assemble_gd(Ctx &c, bool jacobian, int mode);
bool lean3_applies(const Ctx &c);                                   // assemble3.hip
void lean3_release(Ctx &c);
bool launch_assemble_lean3(Ctx &c, bool jacobian, const int *patch_list, int n, uint32_t cmask);
int lean3_signature(const Ctx &c);   // the precompiled model structure the one-pass kernels run with (0: run-time structure)
int gd_prep_setup(Ctx &c, const fedm_csr *mass, int n_tables, const int32_t *tab_ptr,
                  const double *tab_x, const double *tab_y, const fedm_gd_field_prog *progs);
int gd_prep_step(Ctx &c);
void gd_update_mean_energy(Ctx &c);
void gd_prep_release(Ctx &c);
size_t patch_lds_bytes(const Ctx &c, bool jacobian = true);
void launch_finalize(Ctx &c, bool jacobian, int mode);          // Dirichlet + padding rows
void launch_block_inverse(Ctx &c);                              // d_dinv from diagonal blocks
void launch_spmv(Ctx &c, const double *x, double *y, bool scale_dinv, const int *slice_list = nullptr,
                 int n_list = 0);
void launch_ext_source_eval(Ctx &c, int species, const double *params);   // gdprep.hip
void launch_spmv_fieldsplit(Ctx &c, const double *x, double *t, double *z, double *b0, double scale,
                            const int *slice_list = nullptr, int n_list = 0, bool compact32 = false);
void launch_apply_dinv(Ctx &c, const double *x, double *y, double alpha);
void launch_dots(Ctx &c, const double *const *xs, const double *y, int k, bool finish = false);
// (finish = false, several GPUs: the local sums into d_red[0..k) only)   false: not applicable, nothing launched
bool launch_spmv_dots(Ctx &c, const double *x, double *y, const double *const *xs, int k, bool finish = true);
int ensure_spmv_dots(Ctx &c);
// (see cgs_update_fs_kernel, kernels.hip; false: not instantiated for this case)
bool launch_cgs_update_fs(Ctx &c, int k, const double *const *xs, double *y, float *g32, double *b0);
bool launch_scale_copy_fs(Ctx &c, double a, const double *x, double *y, float *g32, double *b0);
void launch_dots_fused(Ctx &c, const double *const *xs, double *y, int k, const double *x0,
                       bool finish = true);
void launch_cgs_finish(Ctx &c, int k);  // finish formulae + publication on d_red[0..k)
void launch_cgs_update(Ctx &c, int k, const double *const *xs, double *y);
// u += delta = sum_i coef_i zs[i] (k <= 8), d_red[1] = |delta|^2, d_red[2] = |u|^2 (owned entries)
void launch_newton_update(Ctx &c, const double *coef_host, int k, const double *const *zs, double *u,
                          double *delta);
void launch_norm2(Ctx &c, const double *x, int slot);                       // d_red[slot] = x.x
void launch_axpy(Ctx &c, double a, const double *x, double *y);             // y += a x
void launch_scale_copy(Ctx &c, double a, const double *x, double *y);       // y = a x
void launch_normalise_copy(Ctx &c, int slot, const double *x, double *y);   // y = x / sqrt(d_red[slot])
void launch_multi_axpy(Ctx &c, const double *coef_host, int k, const double *const *xs,
                       double *y, double sign);                             // y += sign*sum c_i x_i
void launch_field_error(Ctx &c, int comp);  // d_red[0]=|new-old+eps|^2, d_red[1]=|old+eps|^2
void launch_field_error_slots34(Ctx &c, int comp);
void launch_set_dirichlet_state(Ctx &c);    // u[dof] = g
unsigned long long publish_values(Ctx &c, const double *src, int k);   // src[0..k) into the host mailbox, no wait
void publish_values_queued(Ctx &c, const double *src, int k);            // ... the launch alone (stream captures)
void wait_red_seq(Ctx &c, unsigned long long seq);  // a particular publication (steps launched ahead)
void read_red(Ctx &c, int k);
void norm2_read(Ctx &c, const double *x, int slot, int k);  // launch_norm2 + read_red, one kernel fewer on one GPU               // publish d_red[0..k) to h_red and wait for it
void norm2_publish(Ctx &c, const double *x, int slot, int k, int k_sum = 1);  // the launches of norm2_read; wait_red(c) later
void wait_red(Ctx &c);                      // wait for the publication launch_dots(finish) queued

#define FEDM_HIP_CHECK(expr)                                                          \
    do {                                                                              \
        hipError_t e_ = (expr);                                                       \
        if (e_ != hipSuccess) {                                                       \
            fedm::set_error(std::string(#expr) + ": " + hipGetErrorString(e_));        \
            return -1;                                                                \
        }                                                                             \
    } while (0)

void set_error(const std::string &msg);
void iter_graphs_clear(Ctx &c);
int copy_bandwidth(int device, int64_t bytes, int repeats, double *gbs);   // kernels.hip
// Deep halos in force: the mesh carries enough ghost layers for one preconditioner application + Krylov
// product (first stage, fs_sweeps - 1 species sweeps, coupling product, two multigrid smoothings, SpMV:
// fs_sweeps + 2 layers), so a Krylov step exchanges its input vector once and nothing else.
bool deep_halo_active(const Ctx &c);
void prof_begin(Ctx &c, int kind);
void prof_end(Ctx &c);
void prof_collect(Ctx &c);

}  // namespace fedm
