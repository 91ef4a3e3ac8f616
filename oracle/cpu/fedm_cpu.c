/* CPU restatement of the assemble/solve hot path -- TEST INFRASTRUCTURE AND REPORTED BASELINE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call
 * this file; nothing under fedm_amd/ does.  "CPU restatement, not FEniCS": the reference's CPU path
 * is DOLFIN's assemble() + PETSc (fedm/functions.py:188-202, :1047), third-party code that is not
 * in this image; what is restated here is the same algorithm the device library runs, in plain
 * C with OpenMP, so that bench.py can time it on the GPU box's host cores beside the HIP path:
 *
 *   element loop over colour classes (cells of a colour share no vertex)      Problem.F / Problem.J
 *     -> block-CSR Jacobian + residual, Dirichlet rows applied after assembly fedm/functions.py:188-202
 *   weak forms: balance equation in log variables with variable-step BDF2     fedm/functions.py:350-368
 *     drift-diffusion flux :219-237, Poisson :401, Neumann boundary flux :523-524,
 *     sources sum_j nu_j k_j(|E|) prod n^P :835-843, exact Gateaux derivative (fedm-streamer.py:289)
 *   Newton with PETSc SNES newtonls/basic rules                                fedm/functions.py:1047
 *   flexible GMRES(m), preconditioner on the right: field split --
 *     Chebyshev/Richardson sweeps with the point-block diagonal on the species block,
 *     smoothed-aggregation V(1,1) cycle on the (constant) potential block, coupled lower-triangularly
 *   error norm of adaptive_solver                                               fedm/functions.py:1062-1064
 *
 * Its residual and Jacobian are checked against oracle/forms.py (numpy) in tests/test_cpu_backend.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXS 4  /* species */
#define MAXR 8  /* reactions */
#define MAXT 6  /* terms of a coefficient function */
#define MAXTAG 8

typedef struct {
    int32_t n_terms, pad_;
    double c[MAXT], p[MAXT], q[MAXT], r[MAXT]; /* sum_i c E^p exp(q E^r) */
} cpu_termsum;

typedef struct {
    int32_t ns, n_reactions, axisymmetric, n_tags;
    int32_t eq_type[MAXS]; /* 0 reaction, 1 diffusion-reaction, 2 drift-diffusion-reaction */
    double Z[MAXS];
    cpu_termsum mu[MAXS], D[MAXS], k[MAXR];
    int32_t power[MAXR][MAXS], net[MAXR][MAXS];
    double charge_over_eps;
    int32_t bc_neumann[MAXTAG][MAXS];
} cpu_model;

typedef struct {
    int32_t n_rows, n_cols;
    const int64_t *indptr;
    const int32_t *indices;
    const double *values;
} cpu_csr;

typedef struct {
    int n, nc;          /* rows, columns of P (= rows of the next level) */
    int64_t *ap; int32_t *ai; double *av; double *dinv;   /* A */
    int64_t *pp; int32_t *pi; double *pv;                 /* P  (n x n_next) */
    int64_t *rp; int32_t *ri; double *rv;                 /* R = P^T */
    double *x, *b, *r;
} amg_level;

typedef struct {
    int nv, nc, ns, neq;
    cpu_model m;
    double *coords; int32_t *cells; int8_t *ftags;
    /* vertex graph, block CSR */
    int64_t *rowptr; int32_t *col; int32_t *diag; /* diag[v] = index of block (v,v) */
    int32_t *slot;   /* nc*9 */
    int n_colours; int32_t *colour_ptr, *colour_cells;
    int n_bf; int32_t *bf; /* cell, local facet, tag */
    double *val, *F;
    int n_dir; int32_t *dir_dofs; double *dir_vals;
    double *u, *uold, *uold1;
    /* solver work */
    int krylov_cap; double *V, *Zv, *w, *delta, *tmp, *g, *dinv_uu; float *dummy;
    /* multigrid on the potential block */
    int n_levels; amg_level *lev; double *coarse_inv; int n_coarse; double omega;
    int cheb_n; double cheb_w[16];
    long spmv_count, vcycle_count;
} cpu_ctx;

/* ---------------------------------------------------------------------------------------- */
static void termsum_eval(const cpu_termsum *ts, double E, double lnE, double *val, double *der) {
    double v = 0.0, d = 0.0;
    for (int i = 0; i < ts->n_terms; ++i) {
        const double c = ts->c[i], p = ts->p[i], q = ts->q[i], r = ts->r[i];
        if (c == 0.0) continue;
        if (p == 0.0 && q == 0.0) { v += c; continue; }
        const double g = q != 0.0 ? q * exp(r * lnE) : 0.0;
        const double t = c * exp(p * lnE + g);
        v += t;
        d += t * (p + r * g) / E;
    }
    *val = v; *der = d;
}

static int cmp_int(const void *a, const void *b) { return *(const int32_t *)a - *(const int32_t *)b; }

/* FIAT's degree-2 triangle rule (oracle/quadrature.py:triangle_rule(2)) and the 2-point
 * Gauss-Legendre rule on a facet (interval_rule(2)) */
static const double QX[3] = {1.0 / 6.0, 1.0 / 6.0, 2.0 / 3.0}, QY[3] = {1.0 / 6.0, 2.0 / 3.0, 1.0 / 6.0};
static const double QW = 1.0 / 6.0;
static const double FT[2] = {0.21132486540518713, 0.78867513459481287}, FW[2] = {0.5, 0.5};

cpu_ctx *cpu_create(int nv, int nc, const double *coords, const int32_t *cells, const int8_t *ftags,
                    const cpu_model *m, int n_dir, const int32_t *dir_dofs, const double *dir_vals) {
    cpu_ctx *c = (cpu_ctx *)calloc(1, sizeof(cpu_ctx));
    c->nv = nv; c->nc = nc; c->m = *m; c->ns = m->ns; c->neq = m->ns + 1;
    const int neq = c->neq;
    c->coords = (double *)malloc(sizeof(double) * 2 * nv);
    memcpy(c->coords, coords, sizeof(double) * 2 * nv);
    c->cells = (int32_t *)malloc(sizeof(int32_t) * 3 * nc);
    memcpy(c->cells, cells, sizeof(int32_t) * 3 * nc);
    c->ftags = (int8_t *)calloc((size_t)3 * nc, 1);
    if (ftags) memcpy(c->ftags, ftags, (size_t)3 * nc);
    /* vertex adjacency */
    int64_t *cnt = (int64_t *)calloc((size_t)nv + 1, sizeof(int64_t));
    for (int k = 0; k < 3 * nc; ++k) cnt[cells[k] + 1] += 3;
    for (int v = 0; v < nv; ++v) cnt[v + 1] += cnt[v];
    int32_t *cand = (int32_t *)malloc(sizeof(int32_t) * (size_t)cnt[nv]);
    int64_t *pos = (int64_t *)malloc(sizeof(int64_t) * nv);
    memcpy(pos, cnt, sizeof(int64_t) * nv);
    for (int e = 0; e < nc; ++e)
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) cand[pos[cells[3 * e + a]]++] = cells[3 * e + b];
    c->rowptr = (int64_t *)calloc((size_t)nv + 1, sizeof(int64_t));
    c->col = (int32_t *)malloc(sizeof(int32_t) * (size_t)cnt[nv]);
    c->diag = (int32_t *)malloc(sizeof(int32_t) * nv);
    int64_t nb = 0;
    for (int v = 0; v < nv; ++v) {
        int32_t *b = cand + cnt[v];
        int64_t n = cnt[v + 1] - cnt[v];
        if (n == 0) { c->col[nb] = v; c->diag[v] = (int32_t)nb; ++nb; c->rowptr[v + 1] = nb; continue; }
        qsort(b, (size_t)n, sizeof(int32_t), cmp_int);
        int32_t last = -1;
        for (int64_t k = 0; k < n; ++k)
            if (b[k] != last) {
                last = b[k];
                if (last == v) c->diag[v] = (int32_t)nb;
                c->col[nb++] = last;
            }
        c->rowptr[v + 1] = nb;
    }
    free(cand); free(pos); free(cnt);
    c->slot = (int32_t *)malloc(sizeof(int32_t) * (size_t)nc * 9);
    for (int e = 0; e < nc; ++e)
        for (int a = 0; a < 3; ++a) {
            const int v = cells[3 * e + a];
            const int32_t *row = c->col + c->rowptr[v];
            const int len = (int)(c->rowptr[v + 1] - c->rowptr[v]);
            for (int b = 0; b < 3; ++b) {
                const int32_t w = cells[3 * e + b];
                int lo = 0, hi = len;
                while (lo < hi) { const int mid = (lo + hi) / 2; if (row[mid] < w) lo = mid + 1; else hi = mid; }
                c->slot[(size_t)e * 9 + a * 3 + b] = (int32_t)(c->rowptr[v] + lo);
            }
        }
    /* greedy colouring: cells of one colour share no vertex */
    uint64_t *used = (uint64_t *)calloc((size_t)nv, sizeof(uint64_t));
    int32_t *colour = (int32_t *)malloc(sizeof(int32_t) * nc);
    int ncol = 0;
    for (int e = 0; e < nc; ++e) {
        const uint64_t mm = used[cells[3 * e]] | used[cells[3 * e + 1]] | used[cells[3 * e + 2]];
        int k = 0;
        while (k < 63 && ((mm >> k) & 1ULL)) ++k;
        colour[e] = k;
        if (k + 1 > ncol) ncol = k + 1;
        for (int a = 0; a < 3; ++a) used[cells[3 * e + a]] |= 1ULL << k;
    }
    c->n_colours = ncol;
    c->colour_ptr = (int32_t *)calloc((size_t)ncol + 1, sizeof(int32_t));
    for (int e = 0; e < nc; ++e) c->colour_ptr[colour[e] + 1]++;
    for (int k = 0; k < ncol; ++k) c->colour_ptr[k + 1] += c->colour_ptr[k];
    c->colour_cells = (int32_t *)malloc(sizeof(int32_t) * nc);
    int32_t *cp = (int32_t *)malloc(sizeof(int32_t) * ncol);
    memcpy(cp, c->colour_ptr, sizeof(int32_t) * ncol);
    for (int e = 0; e < nc; ++e) c->colour_cells[cp[colour[e]]++] = e;
    free(cp); free(colour); free(used);
    /* tagged facets */
    int nbf = 0;
    for (int k = 0; k < 3 * nc; ++k) nbf += c->ftags[k] > 0;
    c->n_bf = nbf;
    c->bf = (int32_t *)malloc(sizeof(int32_t) * 3 * (size_t)(nbf > 0 ? nbf : 1));
    nbf = 0;
    for (int e = 0; e < nc; ++e)
        for (int i = 0; i < 3; ++i)
            if (c->ftags[3 * e + i] > 0) { c->bf[3 * nbf] = e; c->bf[3 * nbf + 1] = i; c->bf[3 * nbf + 2] = c->ftags[3 * e + i]; ++nbf; }
    const size_t N = (size_t)nv * neq;
    c->val = (double *)calloc((size_t)nb * neq * neq, sizeof(double));
    c->F = (double *)calloc(N, sizeof(double));
    c->u = (double *)calloc(N, sizeof(double));
    c->uold = (double *)calloc(N, sizeof(double));
    c->uold1 = (double *)calloc(N, sizeof(double));
    c->w = (double *)calloc(N, sizeof(double));
    c->delta = (double *)calloc(N, sizeof(double));
    c->tmp = (double *)calloc(N, sizeof(double));
    c->g = (double *)calloc(N, sizeof(double));
    c->dinv_uu = (double *)calloc((size_t)nv * c->ns * c->ns, sizeof(double));
    c->n_dir = n_dir;
    c->dir_dofs = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_dir > 0 ? n_dir : 1));
    c->dir_vals = (double *)malloc(sizeof(double) * (size_t)(n_dir > 0 ? n_dir : 1));
    if (n_dir) { memcpy(c->dir_dofs, dir_dofs, sizeof(int32_t) * n_dir); memcpy(c->dir_vals, dir_vals, sizeof(double) * n_dir); }
    c->cheb_n = 1; c->cheb_w[0] = 1.0; c->omega = 0.85;
    return c;
}

int64_t cpu_nnz_blocks(cpu_ctx *c) { return c->rowptr[c->nv]; }
int cpu_n_colours(cpu_ctx *c) { return c->n_colours; }

void cpu_set_state(cpu_ctx *c, const double *u, const double *uold, const double *uold1) {
    const size_t N = (size_t)c->nv * c->neq;
    if (u) memcpy(c->u, u, sizeof(double) * N);
    if (uold) memcpy(c->uold, uold, sizeof(double) * N);
    if (uold1) memcpy(c->uold1, uold1, sizeof(double) * N);
}
void cpu_get_state(cpu_ctx *c, double *u) { memcpy(u, c->u, sizeof(double) * (size_t)c->nv * c->neq); }
void cpu_shift_state(cpu_ctx *c) {
    const size_t N = (size_t)c->nv * c->neq;
    double *t = c->uold1; c->uold1 = c->uold; c->uold = t;
    memcpy(c->uold, c->u, sizeof(double) * N);
}
void cpu_reset_state(cpu_ctx *c) { memcpy(c->u, c->uold, sizeof(double) * (size_t)c->nv * c->neq); }

/* ---- one cell: Re[3][neq], Ke[3][neq][3][neq] -------------------------------------------------
 * mode 0: full model; mode 1: Poisson row only (species rows are made identity by the caller) */
static void element(const cpu_ctx *c, int e, double dt, double dt_old, int jac, int mode, double *Re, double *Ke) {
    const cpu_model *m = &c->m;
    const int ns = c->ns, neq = c->neq, iphi = ns;
    const int32_t *cv = c->cells + 3 * e;
    double x[3][2], U[3][MAXS + 1], Uo[3][MAXS], Uo1[3][MAXS];
    for (int a = 0; a < 3; ++a) {
        x[a][0] = c->coords[2 * cv[a]]; x[a][1] = c->coords[2 * cv[a] + 1];
        for (int s = 0; s < neq; ++s) U[a][s] = c->u[(size_t)cv[a] * neq + s];
        for (int s = 0; s < ns; ++s) { Uo[a][s] = c->uold[(size_t)cv[a] * neq + s]; Uo1[a][s] = c->uold1[(size_t)cv[a] * neq + s]; }
    }
    const double d1x = x[1][0] - x[0][0], d1y = x[1][1] - x[0][1], d2x = x[2][0] - x[0][0], d2y = x[2][1] - x[0][1];
    const double det = d1x * d2y - d1y * d2x, detJ = fabs(det);
    double G[3][2];
    G[0][0] = (x[1][1] - x[2][1]) / det; G[0][1] = (x[2][0] - x[1][0]) / det;
    G[1][0] = (x[2][1] - x[0][1]) / det; G[1][1] = (x[0][0] - x[2][0]) / det;
    G[2][0] = (x[0][1] - x[1][1]) / det; G[2][1] = (x[1][0] - x[0][0]) / det;
    double rn[3];
    for (int a = 0; a < 3; ++a) rn[a] = m->axisymmetric ? x[a][0] : 0.5 / M_PI;
    double E[2] = {0, 0};
    for (int a = 0; a < 3; ++a) { E[0] -= U[a][iphi] * G[a][0]; E[1] -= U[a][iphi] * G[a][1]; }
    const size_t ks = (size_t)3 * neq; /* Ke[a][s][b][t] at ((a*neq+s)*3+b)*neq+t */
#define KE(a, s, b, t) Ke[(((size_t)(a) * neq + (s)) * 3 + (b)) * neq + (t)]
    (void)ks;
    memset(Re, 0, sizeof(double) * 3 * neq);
    if (jac) memset(Ke, 0, sizeof(double) * 9 * neq * neq);
    double GG[3][3];
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) GG[a][b] = G[a][0] * G[b][0] + G[a][1] * G[b][1];
    const double two_pi = 2.0 * M_PI;
    if (mode == 1) {
        for (int q = 0; q < 3; ++q) {
            const double phi[3] = {1.0 - QX[q] - QY[q], QX[q], QY[q]};
            const double W = QW * detJ * two_pi * (rn[0] * phi[0] + rn[1] * phi[1] + rn[2] * phi[2]);
            double rho = 0.0;
            for (int s = 0; s < ns; ++s) rho += m->Z[s] * exp(U[0][s] * phi[0] + U[1][s] * phi[1] + U[2][s] * phi[2]) * m->charge_over_eps;
            for (int a = 0; a < 3; ++a) {
                Re[a * neq + iphi] += W * (-(E[0] * G[a][0] + E[1] * G[a][1]) - rho * phi[a]);
                if (jac) for (int b = 0; b < 3; ++b) KE(a, iphi, b, iphi) += W * GG[a][b];
            }
        }
        return;
    }
    const double Em = sqrt(E[0] * E[0] + E[1] * E[1]), lnE = log(Em);
    double dEm[3];
    for (int b = 0; b < 3; ++b) dEm[b] = -(E[0] * G[b][0] + E[1] * G[b][1]) / Em;
    double muv[MAXS], mud[MAXS], Dv[MAXS], Dd[MAXS], kv[MAXR], kd[MAXR], gradu[MAXS][2];
    for (int s = 0; s < ns; ++s) {
        termsum_eval(&m->mu[s], Em, lnE, &muv[s], &mud[s]);
        termsum_eval(&m->D[s], Em, lnE, &Dv[s], &Dd[s]);
        gradu[s][0] = gradu[s][1] = 0.0;
        for (int a = 0; a < 3; ++a) { gradu[s][0] += U[a][s] * G[a][0]; gradu[s][1] += U[a][s] * G[a][1]; }
    }
    for (int j = 0; j < m->n_reactions; ++j) termsum_eval(&m->k[j], Em, lnE, &kv[j], &kd[j]);
    const double tr = dt / dt_old, trp1 = 1.0 + tr, tr2p1 = 1.0 + 2.0 * tr;
    for (int q = 0; q < 3; ++q) {
        const double phi[3] = {1.0 - QX[q] - QY[q], QX[q], QY[q]};
        const double W = QW * detJ * two_pi * (rn[0] * phi[0] + rn[1] * phi[1] + rn[2] * phi[2]);
        double nq[MAXS];
        for (int s = 0; s < ns; ++s) {
            const double uq = U[0][s] * phi[0] + U[1][s] * phi[1] + U[2][s] * phi[2];
            const double uo = Uo[0][s] * phi[0] + Uo[1][s] * phi[1] + Uo[2][s] * phi[2];
            const double uo1 = Uo1[0][s] * phi[0] + Uo1[1][s] * phi[1] + Uo1[2][s] * phi[2];
            const double n = exp(uq);
            nq[s] = n;
            /* time derivative, fedm/functions.py:350-357 */
            const double u_part = (uq * tr2p1 - trp1 * trp1 * uo + tr * tr * uo1) / trp1;
            const double T = n * u_part / dt, dT = n * (u_part / dt + tr2p1 / (trp1 * dt));
            for (int a = 0; a < 3; ++a) {
                Re[a * neq + s] += W * T * phi[a];
                if (jac) for (int b = 0; b < 3; ++b) KE(a, s, b, s) += W * dT * phi[a] * phi[b];
            }
            /* flux, fedm/functions.py:219-237 */
            if (m->eq_type[s] == 0) continue;
            double vel[2] = {-Dv[s] * gradu[s][0], -Dv[s] * gradu[s][1]};
            const int drift = m->eq_type[s] == 2;
            if (drift) { vel[0] += m->Z[s] * muv[s] * E[0]; vel[1] += m->Z[s] * muv[s] * E[1]; }
            for (int a = 0; a < 3; ++a) {
                const double velGa = vel[0] * G[a][0] + vel[1] * G[a][1];
                Re[a * neq + s] -= W * n * velGa;
                if (!jac) continue;
                for (int b = 0; b < 3; ++b) {
                    KE(a, s, b, s) -= W * n * (phi[b] * velGa - Dv[s] * GG[a][b]);
                    double dv[2] = {-Dd[s] * dEm[b] * gradu[s][0], -Dd[s] * dEm[b] * gradu[s][1]};
                    if (drift) {
                        dv[0] += m->Z[s] * mud[s] * dEm[b] * E[0] - m->Z[s] * muv[s] * G[b][0];
                        dv[1] += m->Z[s] * mud[s] * dEm[b] * E[1] - m->Z[s] * muv[s] * G[b][1];
                    }
                    KE(a, s, b, iphi) -= W * n * (dv[0] * G[a][0] + dv[1] * G[a][1]);
                }
            }
        }
        /* sources, fedm/functions.py:835-843 */
        for (int j = 0; j < m->n_reactions; ++j) {
            double prod = 1.0;
            for (int i = 0; i < ns; ++i) for (int p = 0; p < m->power[j][i]; ++p) prod *= nq[i];
            for (int s = 0; s < ns; ++s) {
                const double nu = (double)m->net[j][s];
                if (nu == 0.0) continue;
                for (int a = 0; a < 3; ++a) {
                    Re[a * neq + s] -= W * nu * kv[j] * prod * phi[a];
                    if (!jac) continue;
                    for (int b = 0; b < 3; ++b) {
                        for (int i = 0; i < ns; ++i)
                            if (m->power[j][i]) KE(a, s, b, i) -= W * nu * kv[j] * m->power[j][i] * prod * phi[a] * phi[b];
                        KE(a, s, b, iphi) -= W * nu * kd[j] * dEm[b] * prod * phi[a];
                    }
                }
            }
        }
        /* Poisson, fedm/functions.py:401 */
        double rho = 0.0;
        for (int s = 0; s < ns; ++s) rho += m->Z[s] * nq[s] * m->charge_over_eps;
        for (int a = 0; a < 3; ++a) {
            Re[a * neq + iphi] += W * (-(E[0] * G[a][0] + E[1] * G[a][1]) - rho * phi[a]);
            if (!jac) continue;
            for (int b = 0; b < 3; ++b) {
                KE(a, iphi, b, iphi) += W * GG[a][b];
                for (int s = 0; s < ns; ++s) KE(a, iphi, b, s) -= W * m->Z[s] * nq[s] * m->charge_over_eps * phi[a] * phi[b];
            }
        }
    }
}

/* Neumann boundary flux of one tagged facet, fedm/functions.py:523-524 */
static void boundary_facet(cpu_ctx *c, int e, int fi, int tag, int jac) {
    const cpu_model *m = &c->m;
    const int ns = c->ns, neq = c->neq, iphi = ns;
    const int32_t *cv = c->cells + 3 * e;
    double x[3][2], U[3][MAXS + 1];
    for (int a = 0; a < 3; ++a) {
        x[a][0] = c->coords[2 * cv[a]]; x[a][1] = c->coords[2 * cv[a] + 1];
        for (int s = 0; s < neq; ++s) U[a][s] = c->u[(size_t)cv[a] * neq + s];
    }
    const double det = (x[1][0] - x[0][0]) * (x[2][1] - x[0][1]) - (x[1][1] - x[0][1]) * (x[2][0] - x[0][0]);
    double G[3][2];
    G[0][0] = (x[1][1] - x[2][1]) / det; G[0][1] = (x[2][0] - x[1][0]) / det;
    G[1][0] = (x[2][1] - x[0][1]) / det; G[1][1] = (x[0][0] - x[2][0]) / det;
    G[2][0] = (x[0][1] - x[1][1]) / det; G[2][1] = (x[1][0] - x[0][0]) / det;
    double E[2] = {0, 0};
    for (int a = 0; a < 3; ++a) { E[0] -= U[a][iphi] * G[a][0]; E[1] -= U[a][iphi] * G[a][1]; }
    const double Em = sqrt(E[0] * E[0] + E[1] * E[1]), lnE = log(Em);
    double dEm[3];
    for (int b = 0; b < 3; ++b) dEm[b] = -(E[0] * G[b][0] + E[1] * G[b][1]) / Em;
    const int j = fi == 0 ? 1 : 0, k = fi == 2 ? 1 : 2;
    const double gi = sqrt(G[fi][0] * G[fi][0] + G[fi][1] * G[fi][1]);
    const double nrm[2] = {-G[fi][0] / gi, -G[fi][1] / gi};
    const double L = sqrt((x[j][0] - x[k][0]) * (x[j][0] - x[k][0]) + (x[j][1] - x[k][1]) * (x[j][1] - x[k][1]));
    const double En = E[0] * nrm[0] + E[1] * nrm[1];
    for (int s = 0; s < ns; ++s) {
        if (m->eq_type[s] != 2 || !m->bc_neumann[tag - 1][s]) continue;
        double muv, mud;
        termsum_eval(&m->mu[s], Em, lnE, &muv, &mud);
        for (int t = 0; t < 2; ++t) {
            double phi[3] = {0, 0, 0};
            phi[j] = 1.0 - FT[t]; phi[k] = FT[t];
            const double rq = m->axisymmetric ? x[0][0] * phi[0] + x[1][0] * phi[1] + x[2][0] * phi[2] : 0.5 / M_PI;
            const double n = exp(U[0][s] * phi[0] + U[1][s] * phi[1] + U[2][s] * phi[2]);
            const double W = FW[t] * L * 2.0 * M_PI * rq;
            for (int a = 0; a < 3; ++a) {
                if (a == fi) continue;
                c->F[(size_t)cv[a] * neq + s] += W * m->Z[s] * muv * En * n * phi[a];
                if (!jac) continue;
                for (int b = 0; b < 3; ++b) {
                    double *blk = c->val + (size_t)c->slot[(size_t)e * 9 + a * 3 + b] * neq * neq;
                    blk[s * neq + s] += W * m->Z[s] * muv * En * n * phi[a] * phi[b];
                    const double dflux = mud * dEm[b] * En - muv * (G[b][0] * nrm[0] + G[b][1] * nrm[1]);
                    blk[s * neq + iphi] += W * m->Z[s] * dflux * n * phi[a];
                }
            }
        }
    }
}

/* F (and J) of the current state; Dirichlet rows afterwards (bc.apply) */
void cpu_assemble(cpu_ctx *c, double dt, double dt_old, int jac, int mode) {
    const int neq = c->neq, neq2 = neq * neq;
    const size_t N = (size_t)c->nv * neq;
    const int64_t nb = c->rowptr[c->nv];
#pragma omp parallel
    {
#pragma omp for schedule(static) nowait
        for (size_t i = 0; i < N; ++i) c->F[i] = 0.0;
        if (jac) {
#pragma omp for schedule(static)
            for (int64_t i = 0; i < nb * neq2; ++i) c->val[i] = 0.0;
        }
#pragma omp barrier
        double Re[3 * (MAXS + 1)], Ke[9 * (MAXS + 1) * (MAXS + 1)];
        for (int k = 0; k < c->n_colours; ++k) {
#pragma omp for schedule(static)
            for (int t = c->colour_ptr[k]; t < c->colour_ptr[k + 1]; ++t) {
                const int e = c->colour_cells[t];
                element(c, e, dt, dt_old, jac, mode, Re, Ke);
                const int32_t *cv = c->cells + 3 * e;
                for (int a = 0; a < 3; ++a) {
                    for (int s = 0; s < neq; ++s) c->F[(size_t)cv[a] * neq + s] += Re[a * neq + s];
                    if (!jac) continue;
                    for (int b = 0; b < 3; ++b) {
                        double *blk = c->val + (size_t)c->slot[(size_t)e * 9 + a * 3 + b] * neq2;
                        for (int s = 0; s < neq; ++s)
                            for (int t2 = 0; t2 < neq; ++t2) blk[s * neq + t2] += Ke[(((size_t)a * neq + s) * 3 + b) * neq + t2];
                    }
                }
            }
        }
    }
    if (mode == 0)
        for (int f = 0; f < c->n_bf; ++f) boundary_facet(c, c->bf[3 * f], c->bf[3 * f + 1], c->bf[3 * f + 2], jac);
    /* frozen species rows in Poisson-only mode */
    if (mode == 1) {
#pragma omp parallel for schedule(static)
        for (int v = 0; v < c->nv; ++v)
            for (int s = 0; s < c->ns; ++s) {
                c->F[(size_t)v * neq + s] = 0.0;
                if (!jac) continue;
                for (int64_t k = c->rowptr[v]; k < c->rowptr[v + 1]; ++k)
                    for (int t = 0; t < neq; ++t) c->val[(size_t)k * neq2 + s * neq + t] = 0.0;
                c->val[(size_t)c->diag[v] * neq2 + s * neq + s] = 1.0;
            }
    }
    for (int i = 0; i < c->n_dir; ++i) {
        const int dof = c->dir_dofs[i], v = dof / neq, s = dof % neq;
        c->F[dof] = c->u[dof] - c->dir_vals[i];
        if (!jac) continue;
        for (int64_t k = c->rowptr[v]; k < c->rowptr[v + 1]; ++k)
            for (int t = 0; t < neq; ++t) c->val[(size_t)k * neq2 + s * neq + t] = 0.0;
        c->val[(size_t)c->diag[v] * neq2 + s * neq + s] = 1.0;
    }
}

void cpu_get_residual(cpu_ctx *c, double *F) { memcpy(F, c->F, sizeof(double) * (size_t)c->nv * c->neq); }

/* scalar CSR of the assembled Jacobian (tests) */
void cpu_jacobian_csr(cpu_ctx *c, int64_t *indptr, int32_t *indices, double *values) {
    const int neq = c->neq, neq2 = neq * neq;
    int64_t pos = 0;
    indptr[0] = 0;
    for (int v = 0; v < c->nv; ++v)
        for (int s = 0; s < neq; ++s) {
            for (int64_t k = c->rowptr[v]; k < c->rowptr[v + 1]; ++k)
                for (int t = 0; t < neq; ++t) { indices[pos] = c->col[k] * neq + t; values[pos] = c->val[(size_t)k * neq2 + s * neq + t]; ++pos; }
            indptr[(size_t)v * neq + s + 1] = pos;
        }
}

/* block (cr, cc) of the Jacobian as a vertex CSR (multigrid set-up) */
void cpu_block_csr(cpu_ctx *c, int cr, int cc, int64_t *indptr, int32_t *indices, double *values) {
    const int neq = c->neq, neq2 = neq * neq;
    for (int v = 0; v <= c->nv; ++v) indptr[v] = c->rowptr[v];
    for (int64_t k = 0; k < c->rowptr[c->nv]; ++k) { indices[k] = c->col[k]; values[k] = c->val[(size_t)k * neq2 + cr * neq + cc]; }
}

/* ---- vectors ---------------------------------------------------------------------------------- */
static double dot(size_t n, const double *x, const double *y) {
    double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
    for (size_t i = 0; i < n; ++i) s += x[i] * y[i];
    return s;
}
static void axpy(size_t n, double a, const double *x, double *y) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) y[i] += a * x[i];
}
static void scale_copy(size_t n, double a, const double *x, double *y) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) y[i] = a * x[i];
}

/* y = J x (block CSR) */
static void spmv(cpu_ctx *c, const double *x, double *y) {
    const int neq = c->neq, neq2 = neq * neq;
    c->spmv_count++;
#pragma omp parallel for schedule(static)
    for (int v = 0; v < c->nv; ++v) {
        double acc[MAXS + 1] = {0};
        for (int64_t k = c->rowptr[v]; k < c->rowptr[v + 1]; ++k) {
            const double *blk = c->val + (size_t)k * neq2, *xj = x + (size_t)c->col[k] * neq;
            for (int s = 0; s < neq; ++s)
                for (int t = 0; t < neq; ++t) acc[s] += blk[s * neq + t] * xj[t];
        }
        for (int s = 0; s < neq; ++s) y[(size_t)v * neq + s] = acc[s];
    }
}

/* ---- multigrid on the potential block --------------------------------------------------------- */
static void csr_copy(const cpu_csr *m, int64_t **ip, int32_t **ii, double **iv) {
    const int64_t nnz = m->indptr[m->n_rows];
    *ip = (int64_t *)malloc(sizeof(int64_t) * ((size_t)m->n_rows + 1));
    *ii = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1));
    *iv = (double *)malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1));
    memcpy(*ip, m->indptr, sizeof(int64_t) * ((size_t)m->n_rows + 1));
    memcpy(*ii, m->indices, sizeof(int32_t) * (size_t)nnz);
    memcpy(*iv, m->values, sizeof(double) * (size_t)nnz);
}

int cpu_set_amg(cpu_ctx *c, int n_levels, const cpu_csr *A, const cpu_csr *P, const cpu_csr *R,
                const double *coarse_inv, double omega) {
    c->n_levels = n_levels;
    c->lev = (amg_level *)calloc((size_t)n_levels, sizeof(amg_level));
    c->omega = omega;
    for (int l = 0; l < n_levels; ++l) {
        amg_level *L = &c->lev[l];
        L->n = A[l].n_rows;
        csr_copy(&A[l], &L->ap, &L->ai, &L->av);
        L->dinv = (double *)malloc(sizeof(double) * L->n);
        for (int i = 0; i < L->n; ++i) {
            L->dinv[i] = 1.0;
            for (int64_t k = L->ap[i]; k < L->ap[i + 1]; ++k) if (L->ai[k] == i && L->av[k] != 0.0) L->dinv[i] = 1.0 / L->av[k];
        }
        if (l + 1 < n_levels) {
            L->nc = P[l].n_cols;
            csr_copy(&P[l], &L->pp, &L->pi, &L->pv);
            csr_copy(&R[l], &L->rp, &L->ri, &L->rv);
        }
        L->x = (double *)calloc((size_t)L->n, sizeof(double));
        L->b = (double *)calloc((size_t)L->n, sizeof(double));
        L->r = (double *)calloc((size_t)L->n, sizeof(double));
    }
    c->n_coarse = A[n_levels - 1].n_rows;
    c->coarse_inv = (double *)malloc(sizeof(double) * (size_t)c->n_coarse * c->n_coarse);
    memcpy(c->coarse_inv, coarse_inv, sizeof(double) * (size_t)c->n_coarse * c->n_coarse);
    return 0;
}

void cpu_set_chebyshev(cpu_ctx *c, int n, const double *w) {
    c->cheb_n = n;
    for (int i = 0; i < n; ++i) c->cheb_w[i] = w[i];
}

static void csr_mv(int n, const int64_t *ip, const int32_t *ii, const double *iv, const double *x, double *y, int add) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int64_t k = ip[i]; k < ip[i + 1]; ++k) s += iv[k] * x[ii[k]];
        y[i] = add ? y[i] + s : s;
    }
}

/* V(1,1), damped Jacobi; lev[l].b holds the right-hand side, the result lands in lev[l].x */
static void vcycle(cpu_ctx *c, int l) {
    amg_level *L = &c->lev[l];
    const int n = L->n;
    if (l == c->n_levels - 1) {
#pragma omp parallel for schedule(static)
        for (int i = 0; i < n; ++i) {
            double s = 0.0;
            const double *row = c->coarse_inv + (size_t)i * n;
            for (int j = 0; j < n; ++j) s += row[j] * L->b[j];
            L->x[i] = s;
        }
        return;
    }
    const double w = c->omega;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) L->x[i] = w * L->dinv[i] * L->b[i];
    /* r = b - A x */
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int64_t k = L->ap[i]; k < L->ap[i + 1]; ++k) s += L->av[k] * L->x[L->ai[k]];
        L->r[i] = L->b[i] - s;
    }
    csr_mv(L->nc, L->rp, L->ri, L->rv, L->r, c->lev[l + 1].b, 0);
    vcycle(c, l + 1);
    csr_mv(n, L->pp, L->pi, L->pv, c->lev[l + 1].x, L->x, 1);
    /* post-smoothing: x += w Dinv (b - A x) */
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int64_t k = L->ap[i]; k < L->ap[i + 1]; ++k) s += L->av[k] * L->x[L->ai[k]];
        L->r[i] = L->x[i] + w * L->dinv[i] * (L->b[i] - s);
    }
    double *t = L->x; L->x = L->r; L->r = t;
}

/* ---- field split: z = Minv t ------------------------------------------------------------------- */
static void fieldsplit_setup(cpu_ctx *c) {
    const int ns = c->ns, neq = c->neq, neq2 = neq * neq;
#pragma omp parallel for schedule(static)
    for (int v = 0; v < c->nv; ++v) {
        double A[MAXS][MAXS], I[MAXS][MAXS];
        const double *blk = c->val + (size_t)c->diag[v] * neq2;
        for (int r = 0; r < ns; ++r) for (int q = 0; q < ns; ++q) { A[r][q] = blk[r * neq + q]; I[r][q] = r == q; }
        for (int k = 0; k < ns; ++k) {
            int piv = k;
            for (int r = k + 1; r < ns; ++r) if (fabs(A[r][k]) > fabs(A[piv][k])) piv = r;
            if (piv != k) for (int q = 0; q < ns; ++q) { double t = A[k][q]; A[k][q] = A[piv][q]; A[piv][q] = t; t = I[k][q]; I[k][q] = I[piv][q]; I[piv][q] = t; }
            const double inv = 1.0 / A[k][k];
            for (int q = 0; q < ns; ++q) { A[k][q] *= inv; I[k][q] *= inv; }
            for (int r = 0; r < ns; ++r) {
                if (r == k) continue;
                const double f = A[r][k];
                for (int q = 0; q < ns; ++q) { A[r][q] -= f * A[k][q]; I[r][q] -= f * I[k][q]; }
            }
        }
        for (int r = 0; r < ns; ++r) for (int q = 0; q < ns; ++q) c->dinv_uu[((size_t)v * ns + r) * ns + q] = I[r][q];
    }
}

static void fieldsplit_apply(cpu_ctx *c, const double *t, double *z) {
    const int ns = c->ns, neq = c->neq, neq2 = neq * neq, nv = c->nv;
    double *g = c->g, *zt = c->tmp;
    /* first stage: g = Duu^-1 t_u, z_u = w0 g */
#pragma omp parallel for schedule(static)
    for (int v = 0; v < nv; ++v)
        for (int r = 0; r < ns; ++r) {
            double s = 0.0;
            for (int q = 0; q < ns; ++q) s += c->dinv_uu[((size_t)v * ns + r) * ns + q] * t[(size_t)v * neq + q];
            g[(size_t)v * neq + r] = s;
            z[(size_t)v * neq + r] = c->cheb_w[0] * s;
        }
    /* sweeps: z <- z + w (g - Duu^-1 J_uu z) */
    for (int sw = 1; sw < c->cheb_n; ++sw) {
        const double w = c->cheb_w[sw];
#pragma omp parallel for schedule(static)
        for (int v = 0; v < nv; ++v) {
            double acc[MAXS] = {0};
            for (int64_t k = c->rowptr[v]; k < c->rowptr[v + 1]; ++k) {
                const double *blk = c->val + (size_t)k * neq2, *zj = z + (size_t)c->col[k] * neq;
                for (int r = 0; r < ns; ++r) for (int q = 0; q < ns; ++q) acc[r] += blk[r * neq + q] * zj[q];
            }
            for (int r = 0; r < ns; ++r) {
                double s = 0.0;
                for (int q = 0; q < ns; ++q) s += c->dinv_uu[((size_t)v * ns + r) * ns + q] * acc[q];
                zt[(size_t)v * neq + r] = z[(size_t)v * neq + r] + w * (g[(size_t)v * neq + r] - s);
            }
        }
#pragma omp parallel for schedule(static)
        for (int v = 0; v < nv; ++v) for (int r = 0; r < ns; ++r) z[(size_t)v * neq + r] = zt[(size_t)v * neq + r];
    }
    /* coupling: b_phi = t_phi - J_phi,u z_u; then one V-cycle */
    double *b0 = c->lev[0].b;
#pragma omp parallel for schedule(static)
    for (int v = 0; v < nv; ++v) {
        double s = 0.0;
        for (int64_t k = c->rowptr[v]; k < c->rowptr[v + 1]; ++k) {
            const double *blk = c->val + (size_t)k * neq2 + ns * neq, *zj = z + (size_t)c->col[k] * neq;
            for (int q = 0; q < ns; ++q) s += blk[q] * zj[q];
        }
        b0[v] = t[(size_t)v * neq + ns] - s;
    }
    c->vcycle_count++;
    vcycle(c, 0);
    const double *x0 = c->lev[0].x;
#pragma omp parallel for schedule(static)
    for (int v = 0; v < nv; ++v) z[(size_t)v * neq + ns] = x0[v];
}

/* ---- flexible GMRES(m), right preconditioning, modified Gram-Schmidt -------------------------- */
static int fgmres(cpu_ctx *c, const double *b, double bscale, double *x, int restart, double rtol, double atol,
                  int max_it, int *its_out, double *rnorm_out) {
    const size_t N = (size_t)c->nv * c->neq;
    const int m = restart;
    if (c->krylov_cap < m + 1) {
        free(c->V); free(c->Zv);
        c->V = (double *)malloc(sizeof(double) * N * (size_t)(m + 1));
        c->Zv = (double *)malloc(sizeof(double) * N * (size_t)m);
        c->krylov_cap = m + 1;
    }
    double *H = (double *)calloc((size_t)(m + 1) * m, sizeof(double)), *cs = (double *)calloc(m, sizeof(double)),
           *sn = (double *)calloc(m, sizeof(double)), *gv = (double *)calloc(m + 1, sizeof(double)), *y = (double *)calloc(m, sizeof(double));
    int its = 0, rc = 3;
    double r0 = -1.0, rnorm = 0.0;
    memset(x, 0, sizeof(double) * N);
    for (int cycle = 0;; ++cycle) {
        double *v0 = c->V;
        if (cycle == 0) scale_copy(N, bscale, b, v0);
        else { spmv(c, x, c->w); scale_copy(N, bscale, b, v0); axpy(N, -1.0, c->w, v0); }
        const double beta = sqrt(dot(N, v0, v0));
        if (!isfinite(beta)) { rc = 2; break; }
        if (cycle == 0) r0 = beta;
        rnorm = beta;
        const double tol = fmax(rtol * r0, atol);
        if (beta <= tol) { rc = 0; break; }
        if (its >= max_it) break;
        scale_copy(N, 1.0 / beta, v0, v0);
        memset(gv, 0, sizeof(double) * (m + 1));
        gv[0] = beta;
        int j = 0, done = 0;
        for (; j < m && its < max_it; ++j) {
            double *vj = c->V + (size_t)j * N, *zj = c->Zv + (size_t)j * N, *wv = c->V + (size_t)(j + 1) * N;
            fieldsplit_apply(c, vj, zj);
            spmv(c, zj, wv);
            for (int i = 0; i <= j; ++i) {
                const double h = dot(N, c->V + (size_t)i * N, wv);
                H[(size_t)i * m + j] = h;
                axpy(N, -h, c->V + (size_t)i * N, wv);
            }
            const double hn = sqrt(dot(N, wv, wv));
            if (!isfinite(hn)) { rc = 2; goto out; }
            H[(size_t)(j + 1) * m + j] = hn;
            if (hn > 0.0) scale_copy(N, 1.0 / hn, wv, wv);
            for (int i = 0; i < j; ++i) {
                const double t = cs[i] * H[(size_t)i * m + j] + sn[i] * H[(size_t)(i + 1) * m + j];
                H[(size_t)(i + 1) * m + j] = -sn[i] * H[(size_t)i * m + j] + cs[i] * H[(size_t)(i + 1) * m + j];
                H[(size_t)i * m + j] = t;
            }
            const double a = H[(size_t)j * m + j], bb = H[(size_t)(j + 1) * m + j], d = hypot(a, bb);
            cs[j] = d > 0.0 ? a / d : 1.0; sn[j] = d > 0.0 ? bb / d : 0.0;
            H[(size_t)j * m + j] = d; H[(size_t)(j + 1) * m + j] = 0.0;
            gv[j + 1] = -sn[j] * gv[j]; gv[j] = cs[j] * gv[j];
            ++its;
            rnorm = fabs(gv[j + 1]);
            if (rnorm <= tol || hn == 0.0) { ++j; done = 1; break; }
        }
        for (int i = j - 1; i >= 0; --i) {
            double s = gv[i];
            for (int l = i + 1; l < j; ++l) s -= H[(size_t)i * m + l] * y[l];
            y[i] = s / H[(size_t)i * m + i];
        }
        for (int i = 0; i < j; ++i) axpy(N, y[i], c->Zv + (size_t)i * N, x);
        if (done) { rc = 0; break; }
        if (its >= max_it) break;
    }
out:
    free(H); free(cs); free(sn); free(gv); free(y);
    *its_out = its; *rnorm_out = rnorm;
    return rc;
}

typedef struct {
    int32_t iterations, converged, linear_iterations, reason;
    double fnorm0, fnorm;
} cpu_report;

/* PETSc SNES newtonls / basic; reason: 0 converged, 1 max_it, 2 NaN, 3 linear solve */
int cpu_newton(cpu_ctx *c, double dt, double dt_old, double rtol, double atol, double stol, int max_it,
               double ksp_rtol, int ksp_restart, int ksp_max_it, cpu_report *rep) {
    const size_t N = (size_t)c->nv * c->neq;
    int it = 0, lin = 0, rc = 0;
    double fnorm = 0.0, fnorm0 = 0.0, snorm = 0.0;
    for (;;) {
        cpu_assemble(c, dt, dt_old, 1, 0);
        fnorm = sqrt(dot(N, c->F, c->F));
        if (!isfinite(fnorm)) { rc = 2; break; }
        if (it == 0) { fnorm0 = fnorm; if (fnorm < atol) break; }
        else {
            if (fnorm < atol || fnorm <= rtol * fnorm0) break;
            if (snorm < stol * sqrt(dot(N, c->u, c->u))) break;
        }
        if (it >= max_it) { rc = 1; break; }
        fieldsplit_setup(c);
        int lits = 0; double lres = 0.0;
        const int lrc = fgmres(c, c->F, -1.0, c->delta, ksp_restart, ksp_rtol, 1e-50, ksp_max_it, &lits, &lres);
        lin += lits;
        if (lrc != 0) { rc = lrc == 2 ? 2 : 3; break; }
        axpy(N, 1.0, c->delta, c->u);
        snorm = sqrt(dot(N, c->delta, c->delta));
        ++it;
    }
    if (rep) { rep->iterations = it; rep->converged = rc == 0; rep->linear_iterations = lin; rep->reason = rc; rep->fnorm0 = fnorm0; rep->fnorm = fnorm; }
    return rc;
}

/* initial Poisson solve (fedm-streamer.py:205-215): species frozen, V-cycle-preconditioned CG on
 * the potential block (Jacobi when no hierarchy is installed yet) */
int cpu_poisson_solve(cpu_ctx *c, double rtol, int max_it, int *iterations) {
    const int neq = c->neq, ns = c->ns, nv = c->nv;
    for (int i = 0; i < c->n_dir; ++i) c->u[c->dir_dofs[i]] = c->dir_vals[i];
    cpu_assemble(c, 1.0, 1.0, 1, 1);
    double *r = (double *)malloc(sizeof(double) * nv), *z = (double *)malloc(sizeof(double) * nv), *p = (double *)malloc(sizeof(double) * nv),
           *q = (double *)malloc(sizeof(double) * nv), *x = (double *)calloc(nv, sizeof(double));
    const int neq2 = neq * neq;
#define KMV(in, out)                                                                                   \
    _Pragma("omp parallel for schedule(static)") for (int v = 0; v < nv; ++v) {                        \
        double s = 0.0;                                                                                \
        for (int64_t k = c->rowptr[v]; k < c->rowptr[v + 1]; ++k) s += c->val[(size_t)k * neq2 + ns * neq + ns] * (in)[c->col[k]]; \
        (out)[v] = s;                                                                                  \
    }
#define PREC(in, out)                                                                                  \
    if (c->n_levels > 0) { memcpy(c->lev[0].b, (in), sizeof(double) * nv); vcycle(c, 0); memcpy((out), c->lev[0].x, sizeof(double) * nv); } \
    else for (int v = 0; v < nv; ++v) (out)[v] = (in)[v] / c->val[(size_t)c->diag[v] * neq2 + ns * neq + ns];
    for (int v = 0; v < nv; ++v) r[v] = -c->F[(size_t)v * neq + ns];
    PREC(r, z)
    memcpy(p, z, sizeof(double) * nv);
    double rz = dot(nv, r, z);
    const double r0 = sqrt(dot(nv, r, r));
    double rn = r0;
    int it = 0;
    while (rn > rtol * r0 && it < max_it && r0 > 0.0) {
        KMV(p, q)
        const double alpha = rz / dot(nv, p, q);
        axpy(nv, alpha, p, x);
        axpy(nv, -alpha, q, r);
        PREC(r, z)
        const double rz_new = dot(nv, r, z);
        rn = sqrt(dot(nv, r, r));
        if (!isfinite(rn)) break;
        const double beta = rz_new / rz;
        rz = rz_new;
#pragma omp parallel for schedule(static)
        for (int v = 0; v < nv; ++v) p[v] = z[v] + beta * p[v];
        ++it;
    }
    for (int v = 0; v < nv; ++v) c->u[(size_t)v * neq + ns] += x[v];
    free(r); free(z); free(p); free(q); free(x);
    if (iterations) *iterations = it;
    return (rn <= rtol * r0 || r0 == 0.0) ? 0 : 3;
}

/* |new - old + eps| / |old + eps| on one component, fedm/functions.py:1062-1064 */
double cpu_field_error(cpu_ctx *c, int comp) {
    const double eps = 3.0e-16;
    double a = 0.0, b = 0.0;
    const int neq = c->neq;
#pragma omp parallel for reduction(+ : a, b) schedule(static)
    for (int v = 0; v < c->nv; ++v) {
        const double n = c->u[(size_t)v * neq + comp], o = c->uold[(size_t)v * neq + comp];
        a += (n - o + eps) * (n - o + eps);
        b += (o + eps) * (o + eps);
    }
    return sqrt(a) / sqrt(b);
}

/* greedy aggregation (Vanek, Mandel, Brezina 1996) on a strength graph: multigrid set-up */
int cpu_aggregate(int32_t n, const int64_t *indptr, const int32_t *indices, const uint8_t *strong, int32_t *agg) {
    int32_t *a = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; ++i) a[i] = -1;
    int32_t na = 0;
    for (int32_t i = 0; i < n; ++i) {
        if (a[i] >= 0) continue;
        int free_nbhd = 1, cnt = 0;
        for (int64_t k = indptr[i]; k < indptr[i + 1]; ++k) {
            const int32_t j = indices[k];
            if (j == i || !strong[k]) continue;
            ++cnt;
            if (a[j] >= 0) { free_nbhd = 0; break; }
        }
        if (!free_nbhd || cnt == 0) continue;
        a[i] = na;
        for (int64_t k = indptr[i]; k < indptr[i + 1]; ++k) if (strong[k] && indices[k] != i) a[indices[k]] = na;
        ++na;
    }
    for (int i = 0; i < n; ++i) agg[i] = a[i];
    for (int32_t i = 0; i < n; ++i) {
        if (a[i] >= 0) continue;
        for (int64_t k = indptr[i]; k < indptr[i + 1]; ++k) {
            const int32_t j = indices[k];
            if (j != i && strong[k] && a[j] >= 0) { agg[i] = a[j]; break; }
        }
    }
    for (int32_t i = 0; i < n; ++i) {
        if (agg[i] >= 0) continue;
        agg[i] = na;
        for (int64_t k = indptr[i]; k < indptr[i + 1]; ++k) {
            const int32_t j = indices[k];
            if (j != i && strong[k] && agg[j] < 0) agg[j] = na;
        }
        ++na;
    }
    free(a);
    return na;
}

void cpu_counters(cpu_ctx *c, int64_t *out) { out[0] = c->spmv_count; out[1] = c->vcycle_count; }
int cpu_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void cpu_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

void cpu_destroy(cpu_ctx *c) {
    if (!c) return;
    free(c->coords); free(c->cells); free(c->ftags); free(c->rowptr); free(c->col); free(c->diag); free(c->slot);
    free(c->colour_ptr); free(c->colour_cells); free(c->bf); free(c->val); free(c->F); free(c->dir_dofs); free(c->dir_vals);
    free(c->u); free(c->uold); free(c->uold1); free(c->V); free(c->Zv); free(c->w); free(c->delta); free(c->tmp); free(c->g);
    free(c->dinv_uu);
    for (int l = 0; l < c->n_levels; ++l) {
        amg_level *L = &c->lev[l];
        free(L->ap); free(L->ai); free(L->av); free(L->dinv); free(L->pp); free(L->pi); free(L->pv);
        free(L->rp); free(L->ri); free(L->rv); free(L->x); free(L->b); free(L->r);
    }
    free(c->lev); free(c->coarse_inv);
    free(c);
}
