/*
 * fedm_hip.h -- C ABI of libfedm_hip.so: the MI355X (gfx950) replacement for
 * the assemble()/solve() hot path of INP-PM/FEDM.
 *
 * The reference has no FFI; its seam is the DOLFIN subclass protocol
 *     class Problem(df.NonlinearProblem)   F(b, x), J(A, x)   fedm/functions.py:174-202
 * driven by
 *     PETScSNESSolver.solve(problem, x)                       fedm/functions.py:1047
 * Every entry point below names the reference call it replaces.  Plain
 * pointers and sizes only; host pointers are borrowed for the duration of the
 * call; all device memory lives behind the opaque context.
 *
 * Return codes: 0 = ok, <0 = hard error (HIP failure, bad descriptor; see
 * fedm_last_error), >0 = numerical failure (FEDM_DIVERGED_*), which the Python
 * facade turns into RuntimeError so that adaptive_solver's
 * `except Exception` branch (fedm/functions.py:1080-1127) behaves as in the
 * reference.
 *
 * DOF layout: interleaved per vertex, dof = vertex * n_eq + component,
 * species first, potential last (if the model has a Poisson row).
 */
#ifndef FEDM_HIP_H
#define FEDM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FEDM_MAX_SPECIES 4
#define FEDM_MAX_TERMS 6
#define FEDM_MAX_REACTIONS 8
#define FEDM_MAX_TAGS 8
#define FEDM_MAX_QP 32
#define FEDM_MAX_FQP 8
#define FEDM_MAX_EXT_NODES 10

#define FEDM_EQ_REACTION 0                 /* 'reaction'                 functions.py:333 */
#define FEDM_EQ_DIFFUSION_REACTION 1       /* 'diffusion-reaction'                       */
#define FEDM_EQ_DRIFT_DIFFUSION_REACTION 2 /* 'drift-diffusion-reaction'                 */

#define FEDM_BC_ZERO_FLUX 0                /* 'zero flux'                functions.py:472 */
#define FEDM_BC_NEUMANN 1                  /* 'Neumann'                  functions.py:523 */

#define FEDM_DIVERGED_MAX_IT 1
#define FEDM_DIVERGED_NAN 2
#define FEDM_DIVERGED_LINEAR 3

/* f(E) = sum_i c[i] * E^p[i] * exp(q[i] * E^r[i]); normal form of the deck's
 * 'fun:E' strings (file_input/benchmark_model/transport_coefficients/e_Nb.dat:12) */
typedef struct {
    int32_t n_terms;
    int32_t pad_;
    double c[FEDM_MAX_TERMS], p[FEDM_MAX_TERMS], q[FEDM_MAX_TERMS], r[FEDM_MAX_TERMS];
} fedm_termsum;

/* What the weak-form builders of fedm/functions.py:219-528 describe symbolically. */
typedef struct {
    int32_t n_species;                       /* balance equations (log variables)          */
    int32_t poisson;                         /* 1: last component is the potential  :379   */
    int32_t axisymmetric;                    /* 1: r = x[0]; 0: r = 0.5/pi          :251   */
    int32_t n_reactions;
    int32_t eq_type[FEDM_MAX_SPECIES];
    double Z[FEDM_MAX_SPECIES];              /* charge number ("sign")              :232   */
    fedm_termsum mu[FEDM_MAX_SPECIES];       /* mobility(|E|)                               */
    fedm_termsum D[FEDM_MAX_SPECIES];        /* diffusion(|E|)                              */
    int32_t has_drift_w[FEDM_MAX_SPECIES];   /* 1: drift velocity is the constant drift_w   */
    double drift_w[FEDM_MAX_SPECIES][2];
    fedm_termsum k[FEDM_MAX_REACTIONS];      /* rate coefficient(|E|)               :839   */
    int32_t power[FEDM_MAX_REACTIONS][FEDM_MAX_SPECIES];   /* power matrix                  */
    int32_t net[FEDM_MAX_REACTIONS][FEDM_MAX_SPECIES];     /* gain - loss           :841    */
    double charge_over_eps;                  /* e/eps0, fedm-streamer.py:248                */
    int32_t n_tags;
    int32_t bc_kind[FEDM_MAX_TAGS][FEDM_MAX_SPECIES];      /* per boundary tag, species     */
    /* quadrature on the reference triangle / unit interval (FIAT "default") */
    int32_t n_qp;
    int32_t n_fqp;
    double qp_x[FEDM_MAX_QP], qp_y[FEDM_MAX_QP], qp_w[FEDM_MAX_QP];
    double fqp_t[FEDM_MAX_FQP], fqp_w[FEDM_MAX_FQP];
    /* Expression sources (P_k nodal values per cell), fedm-tof.py:116 */
    int32_t ext_nodes[FEDM_MAX_SPECIES];     /* 0: none, else nodes per cell (6 for P2)     */
    double ext_B[FEDM_MAX_QP][FEDM_MAX_EXT_NODES];         /* interpolant at qp_x/qp_y      */
    /* 0: unknowns are ln(n) (weak_form_balance_equation_log_representation, every reference example);
     * 1: unknowns are the densities themselves (weak_form_balance_equation(..., log_representation=False),
     *    Flux(..., logarithm_representation=False), fedm/functions.py:219-237, 350-368) */
    int32_t linear_representation;
    int32_t pad2_;
} fedm_model_desc;

typedef struct {
    int32_t n_vertices;
    int32_t n_cells;
    const double *coords;        /* [n_vertices][2] = (r, z)                               */
    const int32_t *cells;        /* [n_cells][3]                                           */
    const int8_t *facet_tags;    /* [n_cells][3], facet i opposite vertex i; 0 = none      */
    int32_t n_dirichlet;         /* DirichletBC rows, fedm-streamer.py:233                 */
    const int32_t *dirichlet_dofs;
    const double *dirichlet_vals;
    int32_t n_owned_vertices;    /* multi-GPU: vertices [0, n_owned) are owned, the rest are
                                    ghosts of neighbouring partitions; 0 = all owned        */
    /* Deep halos (several ghost layers, fedm_amd/partition.py): the ghost vertices whose rows are
     * identity rows -- the outermost layer.  The rows of all other ghost vertices are assembled
     * like owned rows (redundantly), so that a vector exchanged once stays exact on the owned rows
     * through as many operator applications as there are layers.  NULL: every ghost row is an
     * identity row (one ghost layer).  halo_depth: the number of layers (1 with NULL). */
    int32_t n_identity_vertices;
    const int32_t *identity_vertices;
    int32_t halo_depth;
} fedm_mesh_desc;

typedef struct {
    double rtol, atol, stol;     /* SNES: fedm-streamer.py:295; DOLFIN defaults 1e-10, 1e-16 */
    int32_t max_it;              /* :297                                                    */
    int32_t ksp_restart;         /* GMRES restart (PETSc default 30)                        */
    double ksp_rtol;             /* PETSc default 1e-5, on the preconditioned residual      */
    double ksp_atol;
    int32_t ksp_max_it;          /* PETSc default 10000                                     */
    int32_t watch_component;     /* 1 + the component whose relative change fedm_field_error will be
                                    asked for right after this solve (adaptive_solver's error norm,
                                    fedm/functions.py:1062-1064); 0: none.  Its two sums then ride on the
                                    publication of the final residual check: no extra round trip to the
                                    host.  (One GPU; was padding: 0 keeps the old behaviour.)          */
} fedm_newton_opts;

typedef struct {
    int32_t iterations;          /* Newton iterations taken                                */
    int32_t converged;
    int32_t linear_iterations;   /* total GMRES iterations                                 */
    int32_t reason;              /* 0 or FEDM_DIVERGED_*                                   */
    double fnorm0, fnorm;        /* |F| before / after                                     */
} fedm_newton_report;

/* scalar CSR matrix handed across the boundary (multigrid operators, mass matrix) */
typedef struct {
    int32_t n_rows, n_cols;
    const int64_t *indptr;
    const int32_t *indices;
    const double *values;
} fedm_csr;

/* ---- LMEA model family (glow discharge, examples/glow_discharge/fedm-gd.py) --------------
 * Components: 0 = ln(electron energy density), 1..n_species-1 = ln(number density) of the
 * species with an equation (species 0, the background gas, has none), last = potential.
 * Transport and rate coefficients are nodal P1 fields refreshed by the caller every step
 * (Transport_/Rate_coefficient_interpolation, fedm/functions.py:531-750) and linearised in the
 * mean electron energy on the device (semi_implicit_coefficients, :753-774). */
#define FEDM_GD_MAX_SPECIES 6
#define FEDM_GD_MAX_REACTIONS 16
typedef struct {
    int32_t n_species;                        /* including the background gas            */
    int32_t n_reactions;
    int32_t n_tags;
    int32_t axisymmetric;
    double N0;                                /* gas number density, exp_u[0] in Source_term */
    double charge_over_eps;
    int32_t eq_type[FEDM_GD_MAX_SPECIES];     /* per species (index 0 unused)            */
    int32_t grad_diffusion[FEDM_GD_MAX_SPECIES];
    int32_t is_ion[FEDM_GD_MAX_SPECIES];      /* contributes to Ion_flux, fedm-gd.py:351 */
    double sign[FEDM_GD_MAX_SPECIES];
    double vth[FEDM_GD_MAX_SPECIES];          /* heavy species thermal velocity          */
    double vth_e_coef;                        /* 16 e / (3 pi m_e): vth_e = sqrt(coef * mean energy) */
    int32_t power[FEDM_GD_MAX_REACTIONS][FEDM_GD_MAX_SPECIES];
    int32_t net[FEDM_GD_MAX_REACTIONS][FEDM_GD_MAX_SPECIES];
    double energy_loss[FEDM_GD_MAX_REACTIONS];   /* the decks' two sentinel values stay what they are: 7.77e77 ->
                                                  * (energy_Ei - mean energy), 9.99e99 -> mean energy, :906-909      */
    double ref[FEDM_MAX_TAGS][FEDM_GD_MAX_SPECIES];   /* reflection coefficients         */
    double gamma[FEDM_MAX_TAGS];              /* secondary emission coefficient          */
    double we_secondary;                      /* mean energy of secondary electrons [eV] */
    int32_t n_qp, n_fqp;
    double qp_x[FEDM_MAX_QP], qp_y[FEDM_MAX_QP], qp_w[FEDM_MAX_QP];
    double fqp_t[FEDM_MAX_FQP], fqp_w[FEDM_MAX_FQP];
    /* Energy_Source_term's `Ei` and `mean_energy` arguments (fedm/functions.py:855, 906-909), used by reactions whose
     * energy_loss is a sentinel only.  mean_energy_form: FEDM_GD_ME_UNKNOWN_RATIO = the expression the reference's
     * scripts pass, u[0] / u[n - 1] -- the ratio of the energy and the electron UNKNOWNS at the point, as written
     * (examples/glow_discharge/fedm-gd.py:358) -- differentiated like every other term of the form.  A numeric
     * mean_energy never reaches the device: the host folds it into energy_loss. */
    double energy_Ei;
    int32_t mean_energy_form;
} fedm_gd_desc;
#define FEDM_GD_ME_NONE 0
#define FEDM_GD_ME_UNKNOWN_RATIO 1

/* nodal fields of fedm_gd_set_fields, in this order, each [n_vertices]:
 *   mu[n_species], D[n_species], mu_diff[n_species], D_diff[n_species],
 *   k[n_reactions], k_diff[n_reactions], mean_energy_old, mean_energy, u_e_old           */
#define FEDM_GD_N_FIELDS(ns, nr) (4 * (ns) + 2 * (nr) + 3)

/* Per-step refresh of the LMEA coefficient fields on the device (what the script does on the
 * host between solves, examples/glow_discharge/fedm-gd.py:424-443,452): one "program" per
 * field row of fedm_gd_set_fields. */
#define FEDM_GDP_KEEP 0      /* leave the row as uploaded ('const' dependence, set once)            */
#define FEDM_GDP_TABLE 1     /* np.interp(arg, table) * scale      functions.py:627-630, 739-747    */
#define FEDM_GDP_SCALED_ROW 2 /* scale * fields[src_row]           'ESR': kB*Tgas*mu/e, :633        */
#define FEDM_GDP_ME_OLD 3    /* mean_energy_old <- mean_energy     fedm-gd.py:426-427               */
#define FEDM_GDP_ME 4        /* mean_energy (updated after the solve, fedm-gd.py:452)               */
#define FEDM_GDP_UE_OLD 5    /* ln n_e of the previous step (u_oldV[-1], fedm-gd.py:424)             */
#define FEDM_GDP_ARG_ENERGY 0   /* mean_energy_old */
#define FEDM_GDP_ARG_REDFIELD 1 /* reduced electric field project(1e21*|grad Phi|/N0), :432          */
typedef struct {
    int32_t kind, table, arg, src_row;
    double scale;
} fedm_gd_field_prog;

typedef struct fedm_ctx fedm_ctx;

const char *fedm_last_error(void);
/* Version of this header's structs and entry points; a binding compares it with the constant it was
 * written against and refuses a library of another version (a descriptor that grew would otherwise be
 * read past its end).  2: fedm_model_desc.linear_representation, fedm_newton_opts.watch_component,
 * fedm_pattern_stats out[12], fedm_debug_comm_fault out[6], fedm_pattern_info, fedm_mesh_desc's deep-halo fields.
 * 3: fedm_fieldsplit_tiles_info, fedm_fieldsplit_tiles_stats, fedm_debug_fieldsplit_apply, fedm_debug_fieldsplit_tiles.
 * 4: fedm_state_snapshot, fedm_state_restore; fedm_fieldsplit_tiles_stats out[10]; fedm_comm_stats out[10]; fedm_fieldsplit_policy.
 * 5: fedm_gd_desc.energy_Ei, .mean_energy_form (the sentinel energy losses on the device); fedm_debug_species_planes_check;
 *    fedm_time_kernel kinds 4, 5; fedm_pattern_info out[9]. */
#define FEDM_ABI_VERSION 5
int fedm_abi_version(void);

/* mesh + model -> device: colouring, sliced block-ELL pattern, buffers.
 * Replaces FunctionSpace/derivative/Problem set-up, fedm-streamer.py:133-291. */
int fedm_ctx_create(const fedm_mesh_desc *mesh, const fedm_model_desc *model, int device,
                    fedm_ctx **out);
int fedm_ctx_create_gd(const fedm_mesh_desc *mesh, const fedm_gd_desc *model, int device,
                       fedm_ctx **out);
int fedm_gd_set_fields(fedm_ctx *ctx, const double *fields /* [n_fields][n_vertices] */);
/* mass: consistent P1 mass matrix (Cartesian dx) for project(); tables: concatenated x/y with
 * tab_ptr[n_tables+1]; progs: one per field row */
int fedm_gd_prep_setup(fedm_ctx *ctx, const fedm_csr *mass, int n_tables, const int32_t *tab_ptr,
                       const double *tab_x, const double *tab_y, const fedm_gd_field_prog *progs);
int fedm_gd_prep_step(fedm_ctx *ctx);            /* after fedm_shift_state, before the solve   */
int fedm_gd_update_mean_energy(fedm_ctx *ctx);   /* mean_energy = exp(u_0 - u_e), :452         */
int fedm_gd_get_fields(fedm_ctx *ctx, double *out /* [n_fields][n_vertices] */);
void fedm_ctx_destroy(fedm_ctx *ctx);

/* u_new / u_old / u_old1 (N doubles each, any may be NULL to leave unchanged).
 * Replaces Function.assign / rev_assigner.assign, fedm-streamer.py:282-283,306-307. */
int fedm_set_state(fedm_ctx *ctx, const double *u_new, const double *u_old, const double *u_old1);
int fedm_get_state(fedm_ctx *ctx, double *u_new);
int fedm_get_state_old(fedm_ctx *ctx, double *u_old);
/* u_old1 <- u_old; u_old <- u_new on the device (fedm-streamer.py:306-307) */
int fedm_shift_state(fedm_ctx *ctx);
/* u_new <- u_old (step rejection, fedm/functions.py:1103) */
int fedm_reset_state(fedm_ctx *ctx);
/* The three resident states kept in / brought back from a device-side copy (no host transfer): a
 * checkpoint of the time loop, what a script does with Function.copy(deepcopy=True) before a
 * speculative step.  bench.py repeats its timed window from one. */
int fedm_state_snapshot(fedm_ctx *ctx);
int fedm_state_restore(fedm_ctx *ctx);

/* dt.time_step / dt_old.time_step, fedm/functions.py:350 */
int fedm_set_step(fedm_ctx *ctx, double dt, double dt_old);
/* time-dependent Dirichlet values (time_dependent_arguments, functions.py:1042-1044) */
int fedm_set_dirichlet_values(fedm_ctx *ctx, const double *vals);
/* Expression source of one species for this step: [n_cells][ext_nodes] */
int fedm_set_ext_source(fedm_ctx *ctx, int species, const double *nodal);
/* The same source evaluated on the device.  A DOLFIN Expression string of the arithmetic subset
 * (examples/time_of_flight/fedm-tof.py:116: f = Expression('exp(-(pow(x[1]-w*t, 2)+...)...', D=..., w=...,
 * t=t, degree=2); the script advances f.t before every solve, :145) is handed over once as a postfix
 * program over x[0], x[1], constants and named parameters; fedm_ext_source_eval then fills the
 * species' [n_cells][ext_nodes] table at the P_degree lattice nodes of every cell with the current
 * parameter values -- what fedm_set_ext_source does with a host array, without the host evaluation and
 * the upload.  ops: n_ops pairs (opcode, argument). */
#define FEDM_EXPR_MAX_OPS 256
#define FEDM_EXPR_MAX_PARAMS 16
#define FEDM_EXPR_STACK 24
enum {
    FEDM_OP_CONST = 0, /* push consts[arg]   */
    FEDM_OP_X = 1,     /* push x[arg]        */
    FEDM_OP_PARAM = 2, /* push params[arg]   */
    FEDM_OP_ADD = 3, FEDM_OP_SUB = 4, FEDM_OP_MUL = 5, FEDM_OP_DIV = 6, FEDM_OP_POW = 7, /* binary */
    FEDM_OP_NEG = 8, FEDM_OP_EXP = 9, FEDM_OP_LOG = 10, FEDM_OP_SQRT = 11, FEDM_OP_SIN = 12, FEDM_OP_COS = 13,
    FEDM_OP_TAN = 14, FEDM_OP_FABS = 15, FEDM_OP_TANH = 16, FEDM_OP_ATAN = 17 /* unary */
};
int fedm_ext_source_program(fedm_ctx *ctx, int species, int n_ops, const int32_t *ops /* [n_ops][2] */,
                            int n_consts, const double *consts, int n_params);
int fedm_ext_source_eval(fedm_ctx *ctx, int species, const double *params /* [n_params] */);

/* Problem.F: assemble(F) then bc.apply(b, x)              fedm/functions.py:188-194 */
int fedm_residual(fedm_ctx *ctx, double *F_out /* N or NULL */, double *fnorm /* or NULL */);
/* Problem.J: assemble(J) then bc.apply(A)                 fedm/functions.py:196-202
 * (also assembles F).  csr_* NULL -> stay on device. */
int fedm_jacobian(fedm_ctx *ctx);
/* pattern/values of the assembled Jacobian as scalar CSR, for parity tests */
int64_t fedm_jacobian_nnz(fedm_ctx *ctx);
int fedm_jacobian_csr(fedm_ctx *ctx, int64_t *indptr, int32_t *indices, double *values);
/* y = J x with the assembled Jacobian (KSP mat-vec), for parity tests and bench */
int fedm_spmv(fedm_ctx *ctx, const double *x, double *y);

/* nonlinear_solver.solve(problem, u_new.vector())         fedm/functions.py:1047 */
int fedm_newton_solve(fedm_ctx *ctx, const fedm_newton_opts *opts, fedm_newton_report *rep);
/* Poisson row only, species frozen (initial potential, fedm-streamer.py:205-215) */
int fedm_poisson_solve(fedm_ctx *ctx, double rtol, int max_it, int *iterations);

/* ---- linear solver set-up -----------------------------------------------------------------
 * The reference hands J dx = -F to MUMPS (examples) or to PETSc GMRES + default PC (test
 * harness, tests/integrated_tests/streamer_discharge/fedm_streamer.py:32).  Here: GMRES with
 * point-block Jacobi on the species rows and, when a hierarchy has been installed, one
 * multigrid V-cycle on the (constant) potential block, combined block-triangularly. */

/* scalar CSR of block component (cr, cc) of the assembled Jacobian (n_vertices rows) */
int64_t fedm_block_nnz(fedm_ctx *ctx);
int fedm_block_csr(fedm_ctx *ctx, int cr, int cc, int64_t *indptr, int32_t *indices,
                   double *values);
/* assemble only the Poisson row with the species frozen (identity species rows) */
int fedm_jacobian_poisson_only(fedm_ctx *ctx);
/* install a multigrid hierarchy for the potential block: n_levels operators A[l]
 * (A[0] = fine), n_levels-1 prolongators P[l] (rows of level l, cols of level l+1) and
 * their transposes R[l]; dense inverse of the coarsest operator (NULL when
 * fedm_amg_set_global_hierarchy takes over below the finest level); V(nu,nu) cycles with damped
 * Jacobi smoothing, nu < 0 selects V(0,|nu|) (no pre-smoothing). */
int fedm_amg_setup(fedm_ctx *ctx, int n_levels, const fedm_csr *A, const fedm_csr *P,
                   const fedm_csr *R, const double *coarse_inverse, int nu, double omega);
int fedm_amg_clear(fedm_ctx *ctx);
/* several GPUs: the hierarchy installed with fedm_amg_setup holds the rank's finest level and its
 * level-1 space only (coarse_inverse = NULL there).  The ranks' level-1 spaces are concatenated
 * (n_global unknowns, this rank's start at `offset`) and everything below is ONE global
 * hierarchy, replicated on every rank: A[0] is the Galerkin operator of the undecomposed
 * potential block on that space.  A V-cycle exchanges the finest level's ghost values twice and
 * all-reduces the level-1 right-hand side once. */
int fedm_amg_set_global_hierarchy(fedm_ctx *ctx, int n_global, int offset, int n_levels,
                                  const fedm_csr *A, const fedm_csr *P, const fedm_csr *R,
                                  const double *coarse_inverse, int nu, double omega);
/* The same hierarchy with a polynomial smoother: `degree` Richardson sweeps x += w Dinv (b - A x)
 * per leg, weights[level * degree + i] in the order of the pre-smoother (the post-smoother runs them
 * backwards: the cycle stays symmetric for the Poisson-only CG); Chebyshev roots on
 * [lambda_max / a, lambda_max] of Dinv A are the intended choice.  Costs two more streams of the
 * finest operator per cycle at degree 2 (the coarse levels stay two kernels each: the polynomial is
 * folded into their composite matrices) and contracts 0.38 instead of 0.68 per cycle on the bench
 * mesh.  as_alternative = 0: replaces the hierarchy of fedm_amg_setup.  1: installed next to it and
 * used together with the alternative species sweeps (fedm_set_fieldsplit_alternative: Newton solves
 * that need many Krylov steps), one GPU only. */
int fedm_amg_setup_poly(fedm_ctx *ctx, int n_levels, const fedm_csr *A, const fedm_csr *P,
                        const fedm_csr *R, const double *coarse_inverse, int degree,
                        const double *weights, int as_alternative);
/* Richardson sweeps z += w_k Duu^-1 (r - Juu z) on the species block inside the field split;
 * one weight per sweep (equal weights = damped block Jacobi, Chebyshev roots = polynomial) */
int fedm_set_fieldsplit(fedm_ctx *ctx, int sweeps, const double *weights);
/* A second set of sweeps for hard systems: after a Newton solve that needed at least
 * `switch_above` Krylov steps per Newton iteration the field split uses the alternative set, after
 * one that needed at most `back_below` the set of fedm_set_fieldsplit again (which also resets this).
 * A long polynomial pays while it saves whole Krylov steps: with the potential first in the split
 * (fedm_set_fieldsplit_order, opt-in) the species polynomial decides the Krylov count and the
 * alternative is a HIGHER degree; with the species first (default) the potential block limits the
 * convergence late in a run and the alternative is a cheaper one.  alt_sweeps = 0 switches the rule off. */
int fedm_set_fieldsplit_alternative(fedm_ctx *ctx, int alt_sweeps, const double *alt_weights,
                                    double switch_above, double back_below);
/* Which of the two sets ran: out = {policy (1: the set that is measured faster -- wall time per Newton iteration, one
 * GPU; 0: by the Krylov counts alone), alternative set active now, Newton solves run under the main set, under the
 * alternative set}.  The measured policy makes a run's trajectory of sets depend on the machine; bench.py prints
 * these counts per window so that two runs can be told apart (FEDM_FS_POLICY=counts: reproducible). */
int fedm_fieldsplit_policy(fedm_ctx *ctx, int64_t out[4]);
/* host-side greedy aggregation on a strength graph (set-up helper, no GPU needed) */
int fedm_amg_aggregate(int32_t n, const int64_t *indptr, const int32_t *indices,
                       const uint8_t *strong, int32_t *agg, int32_t *n_agg);

/* ---- multi-GPU (one process per GPU; what DOLFIN/PETSc do implicitly under mpirun,
 * README.md:63-67): ghost-value exchange before SpMV/assembly + all-reduces for dots/norms.
 * Plan in local vertex numbering: for neighbour k (rank nb_rank[k]) send the owned vertices
 * send_idx[send_ptr[k]..send_ptr[k+1]) and receive the ghost vertices
 * n_owned + [recv_ptr[k], recv_ptr[k+1]).  Both sides list a shared vertex set in the same
 * (global id) order. */
typedef void (*fedm_allreduce_fn)(double *host_buf, int32_t n, void *user);
typedef void (*fedm_exchange_fn)(const double *send_host, double *recv_host, int32_t width,
                                 void *user);
int fedm_comm_unique_id(void *out128 /* ncclUniqueId */);
int fedm_comm_init_rccl(fedm_ctx *ctx, int n_neighbours, const int32_t *nb_rank,
                        const int32_t *send_ptr, const int32_t *send_idx, const int32_t *recv_ptr,
                        const void *unique_id, int rank, int n_ranks);
/* host-staged transport owned by the caller (e.g. torch.distributed gloo); tests use it */
int fedm_comm_init_callbacks(fedm_ctx *ctx, int n_neighbours, const int32_t *nb_rank,
                             const int32_t *send_ptr, const int32_t *send_idx,
                             const int32_t *recv_ptr, fedm_allreduce_fn allreduce,
                             fedm_exchange_fn exchange, void *user, int rank, int n_ranks);
/* refresh the ghost entries of u_new, u_old, u_old1 from their owners */
int fedm_sync_ghosts(fedm_ctx *ctx);
/* Transport errors: every RCCL return code is checked; the first failure is latched in the
 * context (message in fedm_last_error), all later exchanges / reductions are skipped and
 * fedm_newton_solve, fedm_poisson_solve, fedm_sync_ghosts and fedm_field_error return -1.
 * fedm_comm_stats: out = {transport kind (0 none, 1 host callbacks, 2 RCCL), ranks, halo
 * exchanges issued, all-reduces issued, failed flag, neighbours, assembly patches without /
 * with ghost vertices (the former are assembled while the state halo travels), bytes this rank has
 * sent in halo exchanges, payload bytes it has contributed to all-reduces}. */
int fedm_comm_stats(fedm_ctx *ctx, int64_t out[10]);
/* latency of the transport's primitives, back to back on the compute stream (all ranks must call
 * it together): kind 0 = halo exchange of a block vector, 1 = of one value per vertex,
 * 2 = all-reduce of 32 doubles.  ms per operation. */
int fedm_time_comm(fedm_ctx *ctx, int kind, int repeats, double *ms_per_op);
/* Test hook (no GPU needed): drives the RCCL code path with a stub transport whose
 * `fail_at`-th call fails.  out = {failed flag latched, transport calls made, calls made after
 * the failing one, comm_failed(), ncclCommAbort calls at teardown, ncclCommDestroy calls at
 * teardown}: a failed communicator is aborted, never destroyed (its unfinished collective would
 * block ncclCommDestroy and every stream synchronisation after it).  Returns 1 when a failure was
 * latched. */
int fedm_debug_comm_fault(int fail_at, int64_t out[6]);
/* One halo exchange of the DOF vector `vec` (N values in the caller's order, ghost entries replaced by
 * what the neighbours sent) and one sum over the ranks of `red[0..k)`, through whatever transport the
 * context has: lets a test drive the transport's data path by itself (e.g. RCCL on one rank that is
 * its own neighbour).  The exchange runs twice (compute stream; communication stream fenced by events),
 * the sum twice (double; single-precision payload, so red comes back rounded to fp32).  red may be NULL. */
int fedm_debug_comm_roundtrip(fedm_ctx *ctx, double *vec, double *red, int k);
/* Host-side preprocessing alone (no GPU needed): what DOLFIN builds with the FunctionSpace /
 * sparsity pattern (fedm/functions.py:192,200).  out = {matrix slices (= assembly patches),
 * max cells per patch, max block columns per slice, max staged vertices per patch, cell visits
 * of all patches, owned (cell, local vertex) pairs, pairs that clash with another cell of their
 * 16-lane group on an LDS accumulator bank (FEDM_PATCH_ORDER, see csrc/prep.cpp), structural blocks,
 * stored blocks of the sliced block-ELL layout (structural + padding), halo vertices staged by all
 * patches, colours of the global cell colouring, (wave of 64 patch cells, local row) pairs in which some
 * cell owns that vertex: what the assembly kernels emit; the others are skipped}. */
int fedm_pattern_stats(const fedm_mesh_desc *mesh, int64_t out[12]);
/* The same for a live context, and which volume-assembly kernels it runs: out = {slices, max cells per
 * patch, max block columns per slice, max staged vertices per patch, cell visits, halo vertices,
 * assembly variant (0 global colouring, 1 LDS patches / unrolled element routine, 2 LDS patches / one
 * equation row at a time, 3 LDS patches / one pass over the cells), threads per patch workgroup, 1 when the one-pass
 * kernels run with the model's STRUCTURE compiled in (a precompiled signature matches it: csrc/assemble3.hip) and only
 * its numbers read at run time}. */
int fedm_pattern_info(fedm_ctx *ctx, int64_t out[9]);
/* The species sweeps of the field split (the Chebyshev polynomial in Duu^-1 Juu that stands for PETSc's
 * sub-solver of the species block; no counterpart in the scripts): on one GPU they run several per launch on
 * tiles of matrix slices whose vertex layers sit in LDS (csrc/fs_tiles.hip).  Returns 1 when this context
 * does so (builds the tiles if need be), 0 when it runs them one by one (several GPUs, more than two species,
 * FEDM_FS_TILES=0); out = {tiles, slices per tile, vertex layers (= sweeps per launch), longest matrix row,
 * most vertices of a tile with its layers, most rows, bytes of the tile tables, threads per tile, rows of all
 * tiles (the tile's own and the layers': the redundancy is this over the vertex count), vertices of all tiles}. */
int fedm_fieldsplit_tiles_info(fedm_ctx *ctx, int64_t out[10]);
/* The tile tables of a mesh built on the host alone (no GPU needed), with a self-check: out = {tiles, longest
 * matrix row, most vertices of a tile with its layers, most rows, rows of all tiles, vertices of all tiles, bytes,
 * violations found (0: every vertex is the own vertex of exactly one tile, the layers nest, every local column
 * number names the vertex the block pattern names), dynamic LDS bytes per workgroup of the species-sweep kernel
 * (two species) and of the multigrid-sweep kernel on these tiles: tiles whose kernels would need more than the
 * device grants a workgroup (160 KiB on gfx950) are refused when a context builds them}. */
int fedm_fieldsplit_tiles_stats(const fedm_mesh_desc *mesh, int tile_slices, int depth, int64_t out[10]);
/* Test hook: z = Minv t with the field-split preconditioner of the CURRENT Jacobian (fedm_jacobian first; its
 * species planes are formed here), host vectors of n_vertices * n_eq doubles -- the operator a Krylov step of
 * fedm_newton_solve applies, alone. */
int fedm_debug_fieldsplit_apply(fedm_ctx *ctx, const double *t, double *z);
/* Test hook: mode 0 = species sweeps one launch each from now on (and the multigrid's finest-level sweeps kernels of
 * their own); 3 = species sweeps on tiles, the multigrid's not; 1 = tiles, rebuilt with `tile_slices` slices
 * per tile, `depth` vertex layers and `threads` threads per tile (0: defaults, FEDM_FS_TILE_SLICES /
 * FEDM_FS_TILE_DEPTH / FEDM_FS_TILE_THREADS). */
int fedm_debug_fieldsplit_tiles(fedm_ctx *ctx, int mode, int tile_slices, int depth, int threads);
/* Test hook: the field split's species planes as they stand (formed by the one-pass assembly itself plus the rows
 * changed behind it, or by the separate pass) against the separate pass over the Jacobian as it stands.
 * out[4] = {max |d Duu^-1| / max |Duu^-1|, max |d S16|, max |d coupling| / max |coupling|, 1 if the last set-up was
 * the fused one}. */
int fedm_debug_species_planes_check(fedm_ctx *ctx, double *out);

/* |new - old + eps| / |old + eps| on one component        fedm/functions.py:1062-1064 */
int fedm_field_error(fedm_ctx *ctx, int component, double *rel_err);

/* timed micro-benchmarks on the resident state (HIP events on the library's stream):
 * kind 0 = residual+Jacobian assembly, 1 = SpMV, 2 = residual only, 3 = one multigrid cycle on the
 * potential block, 4 = the field split's set-up behind an assembly (species planes), 5 = assembly + that set-up.
 * ms per launch. */
int fedm_time_kernel(fedm_ctx *ctx, int kind, int repeats, double *ms_per_launch);
/* Copy rate of the device (GB/s, read + write bytes): a 16-byte-per-lane grid-stride copy between two
 * buffers of `bytes` each -- the practical HBM ceiling bench.py prints next to the specification. */
int fedm_copy_bandwidth(int device, int64_t bytes, int repeats, double *gbs);
/* in-run kernel timing with HIP events on the library's stream.  kind: 0 = assembly F+J,
 * 1 = Jacobian SpMV, 2 = assembly F only, 3 = multigrid V-cycle (whole graph).  Kinds 1 and 3
 * are sampled (every 4th launch carries events; totals are the sampled mean x launches).
 * enable = 1 times the assembly only and leaves the solver untouched; enable = 2 also times
 * kinds 1 and 3, which makes GMRES launch its kernels one by one instead of replaying the
 * captured per-iteration graphs (slower; use it for a separate profiling pass). */
int fedm_profile(fedm_ctx *ctx, int enable);
int fedm_profile_read(fedm_ctx *ctx, int kind, double *ms_total, int64_t *count);
/* assembly kernel: 0 = global graph colouring (bitwise reproducible), 1 = LDS patches
 * (default; each matrix value written once, LDS fp64 atomics) */
int fedm_set_assembly(fedm_ctx *ctx, int kind);
/* side of the field-split preconditioner in the Newton linear solves: 1 = right (flexible GMRES,
 * convergence on the true residual norm; default for LFA models), 0 = left (convergence on the
 * preconditioned residual norm; default for LMEA models; one GPU only: across GPUs the field split
 * is always on the right).  Both solve J delta = -F to ksp_rtol.
 * Environment: FEDM_PRECOND_SIDE=left|right sets the default of new contexts. */
int fedm_set_preconditioner_side(fedm_ctx *ctx, int right);
/* Order of the block-triangular field split when it sits on the right of the operator (the
 * left-preconditioned path is always lower):
 *   0 = lower (default): species sweeps first, then the V-cycle on the potential right-hand side
 *       minus J_phi,u z_u.  The true residual tracks the error of the iterate.
 *   1 = upper: V-cycle on the potential block first, then the species sweeps on t_u - J_u,phi z_phi.
 *       Passes the residual test in 3-4 instead of 8 Krylov steps once a streamer has formed, but
 *       leaves the potential at V-cycle accuracy: the unscaled residual norm (PETSc's and this
 *       library's test) is dominated by the species rows and does not see the Poisson row.  2x
 *       faster late in a run, 4e-3 instead of 6e-6 deviation from a tightly solved run after 220
 *       steps of the bench case (tools/fs_order_accuracy.py): an explicit trade, not the default.
 * Takes effect with the next Jacobian assembly.  Environment: FEDM_FS_ORDER=lower|upper. */
int fedm_set_fieldsplit_order(fedm_ctx *ctx, int upper);
/* value planes of the Jacobian blocks (bit r * n_eq + c = d(row r)/d(unknown c)) that the assembly
 * keeps between assemblies (constant or structurally zero: not recomputed, not written again) and
 * that the Jacobian SpMV skips (structurally zero: no reaction couples the two species): the bytes
 * the roofline model must not count. */
int fedm_plane_masks(fedm_ctx *ctx, uint32_t *kept_planes, uint32_t *zero_planes);
/* sizes the roofline model needs */
int fedm_sizes(fedm_ctx *ctx, int64_t *n_vertices, int64_t *n_cells, int64_t *n_eq,
               int64_t *nnz_blocks, int64_t *stored_blocks, int64_t *n_colours);

#ifdef __cplusplus
}
#endif
#endif /* FEDM_HIP_H */
