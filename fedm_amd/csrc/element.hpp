// Per-cell evaluation of FEDM's LFA weak forms on a P1 triangle (device code, gfx950).
//
// What FFC's generated tabulate_tensor does for the forms of fedm/functions.py:350-368
// (balance equation, log variables, variable-step BDF2), :219-237 (drift-diffusion flux),
// :401 (Poisson) and :523-524 (Neumann boundary flux), with the exact Gateaux derivative
// that `derivative(F, u_new, u)` (fedm-streamer.py:289) produces, hand-derived.
//
// P1 makes grad(u) and grad(Phi) -- hence E, |E| and every |E|-dependent coefficient --
// constant per cell.  The quadrature loop therefore only accumulates weighted moments of
// the P1 basis (sum_q W g(q) phi_a phi_b, sum_q W g(q) phi_a, sum_q W g(q)); the element
// residual and the n_eq x n_eq Jacobian blocks are closed-form combinations of them.
#pragma once
#include "fedm_internal.hpp"

namespace fedm {

__device__ __forceinline__ int sym6(int a, int b) {
    // (0,0)=0 (1,1)=1 (2,2)=2 (0,1)=3 (0,2)=4 (1,2)=5
    return a == b ? a : (a + b + 2);
}

// value and d/dE of  sum_i c_i E^p_i exp(q_i E^r_i).  E^r needs no transcendental for the
// integer r the decks use (exp(-2.73e7/E_m): r = -1); divisions are multiplications by 1/E.
__device__ __forceinline__ void termsum_eval(const fedm_termsum &ts, double E, double invE,
                                             double lnE, double &val, double &der) {
    val = 0.0;
    der = 0.0;
    for (int i = 0; i < ts.n_terms; ++i) {
        const double c = ts.c[i], p = ts.p[i], q = ts.q[i], r = ts.r[i];
        if (c == 0.0) continue;
        if (p == 0.0 && q == 0.0) {
            val += c;
            continue;
        }
        double g = 0.0;
        if (q != 0.0) {
            if (r == -1.0) g = q * invE;
            else if (r == 1.0) g = q * E;
            else if (r == -2.0) g = q * invE * invE;
            else if (r == 2.0) g = q * E * E;
            else g = q * exp(r * lnE);
        }
        const double t = c * exp(p * lnE + g);
        val += t;
        der += t * (p + r * g) * invE;
    }
}

// Variable-step BDF2 in the log variable, fedm/functions.py:350-357:
//   u_part = (u*(1+2w) - (1+w)^2 u_old + w^2 u_old1) / (1+w),   w = dt/dt_old.
// The history part is linear in the nodal values, so it is folded per node into
//   hist = c_old * u_old + c_old1 * u_old1        and       u_part = c_new * u + hist.
struct StepCoef {
    double dt, inv_dt, c_new, c_old, c_old1;
};

__host__ __device__ inline StepCoef step_coef(double dt, double dt_old) {
    StepCoef s;
    const double tr = dt / dt_old;
    const double trp1 = 1.0 + tr;
    s.dt = dt;
    s.inv_dt = 1.0 / dt;
    s.c_new = (1.0 + 2.0 * tr) / trp1;
    s.c_old = -(trp1 * trp1) / trp1;
    s.c_old1 = (tr * tr) / trp1;
    return s;
}

// cell geometry: gradients of the P1 basis, |det J|, vertex radii
struct CellGeom {
    double G[3][2];
    double detJ, inv_det;
    double rn[3];
    // idet_known != 0: 1/det from an earlier evaluation of the same cell (the division is a couple
    // of dozen fp64 instructions; the row-at-a-time routine revisits a cell once per equation row)
    __device__ __forceinline__ void init(const double x[3][2], int axisymmetric, double idet_known = 0.0) {
        const double d1x = x[1][0] - x[0][0], d1y = x[1][1] - x[0][1];
        const double d2x = x[2][0] - x[0][0], d2y = x[2][1] - x[0][1];
        const double det = d1x * d2y - d1y * d2x;
        const double idet = idet_known != 0.0 ? idet_known : 1.0 / det;
        inv_det = idet;
        detJ = fabs(det);
        G[0][0] = (x[1][1] - x[2][1]) * idet;
        G[0][1] = (x[2][0] - x[1][0]) * idet;
        G[1][0] = (x[2][1] - x[0][1]) * idet;
        G[1][1] = (x[0][0] - x[2][0]) * idet;
        G[2][0] = (x[0][1] - x[1][1]) * idet;
        G[2][1] = (x[1][0] - x[0][0]) * idet;
#pragma unroll
        for (int a = 0; a < 3; ++a) rn[a] = axisymmetric ? x[a][0] : 0.5 / 3.14159265358979323846;
    }
};

// NR: compile-time bound on the number of reactions (keeps their coefficients in registers).
// CACHE: keep exp(u) and the BDF term at the (<= 3) quadrature points in registers and emit
// the element tensors one equation row at a time, so that only one row's moments are live
// (n_eq = 3: 22 instead of 60 doubles); without it every row pass re-evaluates exp(u).
// CACHE = 2 additionally bakes FIAT's degree-2 triangle rule (the only 3-point rule the decks
// produce) into the code: every model scalar read in the inner loops is a scalar-cache round trip
// that the few resident waves cannot hide, and the constant weights fold into the arithmetic.
// LIN: the unknowns are the densities themselves (fedm_model_desc::linear_representation); a
// compile-time switch -- as a run-time flag it costs the logarithmic kernels up to 300 bytes of scratch.
template <int NS, bool PO, int NR, int CACHE, bool LIN = false>
struct Element {
    static constexpr int NEQ = NS + (PO ? 1 : 0);
    static constexpr int IPHI = NEQ - 1;
    static constexpr int NQC = 3;
    static constexpr bool CACHED = CACHE != 0, QSTD = CACHE == 2;

    __device__ __forceinline__ static double qx(const fedm_model_desc *__restrict__ md, int q) {
        return QSTD ? (q == 2 ? 2.0 / 3.0 : 1.0 / 6.0) : md->qp_x[q];
    }
    __device__ __forceinline__ static double qy(const fedm_model_desc *__restrict__ md, int q) {
        return QSTD ? (q == 1 ? 2.0 / 3.0 : 1.0 / 6.0) : md->qp_y[q];
    }
    __device__ __forceinline__ static double qw(const fedm_model_desc *__restrict__ md, int q) {
        return QSTD ? 1.0 / 6.0 : md->qp_w[q];
    }

    // cell constants
    double G[3][2], detJ, rn[3];
    double gradPhi[2], E[2], invEm;
    double Dv[NS], Dd[NS], muv[NS], mud[NS], vel[NS][2], gradu[NS][2];
    double kv[NR > 0 ? NR : 1], kd[NR > 0 ? NR : 1];
    double nq_c[NS][NQC], up_c[NS][NQC];
    bool flux[NS], fdrift[NS], full;
    static constexpr bool lin = LIN;
    int nreac;
    // moments of the row being emitted (m01 = sum W, m1w[b] = sum W phi_b: the linear representation's
    // counterparts of m0n, m1n)
    double m2[NS][6], m1h[3], m1n[3], m0n, m1sp[3], m01, m1w[3];
    // per cell: G_a.G_b (sym6 order) and d|E|/dPhi_b;  per row (row_prepare): the factors that
    // every (a, b) entry shares, so that an entry costs two FMAs instead of a dozen:
    //   J[s][s](a,b)   = m2 + DGm*gg(a,b) - velG[a]*m1n[b]
    //   J[s][phi](a,b) = Kd*gg(a,b) - dE[b]*T[a]
    double gg[6], dE[3];
    double velG[3], T[3], DGm, Kd;

    __device__ __forceinline__ double dEm(int b) const {
        return -(E[0] * G[b][0] + E[1] * G[b][1]) * invEm;
    }
    __device__ __forceinline__ double GG(int a, int b) const {
        return G[a][0] * G[b][0] + G[a][1] * G[b][1];
    }

    __device__ void setup(const fedm_model_desc *__restrict__ md, const double x[3][2],
                          const double Uc[3][NEQ], const double Hc[3][NS], const StepCoef sc,
                          int mode) {
        CellGeom cg;
        cg.init(x, md->axisymmetric);
        detJ = cg.detJ;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            G[a][0] = cg.G[a][0];
            G[a][1] = cg.G[a][1];
            rn[a] = cg.rn[a];
        }
        double Em = 1.0;
        E[0] = E[1] = gradPhi[0] = gradPhi[1] = 0.0;
        invEm = 0.0;
        full = (mode == 0);
        if (PO) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                gradPhi[0] += Uc[a][IPHI] * G[a][0];
                gradPhi[1] += Uc[a][IPHI] * G[a][1];
            }
            E[0] = -gradPhi[0];
            E[1] = -gradPhi[1];
            if (full) {
                Em = sqrt(E[0] * E[0] + E[1] * E[1]);
                invEm = 1.0 / Em;
            }
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            dE[a] = dEm(a);
#pragma unroll
            for (int b = a; b < 3; ++b) gg[sym6(a, b)] = GG(a, b);
        }
        const double lnE = log(Em);
        const double invEm_ = full && PO ? invEm : 1.0;
        nreac = full ? md->n_reactions : 0;
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            kv[j] = kd[j] = 0.0;
            if (j < nreac) termsum_eval(md->k[j], Em, invEm_, lnE, kv[j], kd[j]);
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            gradu[s][0] = gradu[s][1] = 0.0;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                gradu[s][0] += Uc[a][s] * G[a][0];
                gradu[s][1] += Uc[a][s] * G[a][1];
            }
            muv[s] = mud[s] = Dv[s] = Dd[s] = 0.0;
            flux[s] = full && md->eq_type[s] != FEDM_EQ_REACTION;
            fdrift[s] = false;
            vel[s][0] = vel[s][1] = 0.0;
            if (flux[s]) {
                termsum_eval(md->D[s], Em, invEm_, lnE, Dv[s], Dd[s]);
                vel[s][0] = -Dv[s] * gradu[s][0];
                vel[s][1] = -Dv[s] * gradu[s][1];
                if (md->eq_type[s] == FEDM_EQ_DRIFT_DIFFUSION_REACTION) {
                    if (md->has_drift_w[s]) {
                        vel[s][0] += md->drift_w[s][0];
                        vel[s][1] += md->drift_w[s][1];
                    } else if (PO) {
                        termsum_eval(md->mu[s], Em, invEm_, lnE, muv[s], mud[s]);
                        vel[s][0] += md->Z[s] * muv[s] * E[0];
                        vel[s][1] += md->Z[s] * muv[s] * E[1];
                        fdrift[s] = true;
                    }
                }
            }
        }
        if (CACHED) {
#pragma unroll
            for (int q = 0; q < NQC; ++q) {
                const double xq = qx(md, q), yq = qy(md, q);
                const double p0 = 1.0 - xq - yq;
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const double u = Uc[0][s] * p0 + Uc[1][s] * xq + Uc[2][s] * yq;
                    nq_c[s][q] = lin ? u : exp(u);
                    up_c[s][q] = sc.c_new * u + (Hc[0][s] * p0 + Hc[1][s] * xq + Hc[2][s] * yq);
                }
            }
        }
    }

    // moments of equation row `row` (compile-time constant after unrolling)
    __device__ __forceinline__ void row_moments(const fedm_model_desc *__restrict__ md, int row,
                                                const double Uc[3][NEQ], const double Hc[3][NS],
                                                const StepCoef sc, const double *const ext[NS]) {
        const double two_pi = 6.283185307179586476925286766559;
#pragma unroll
        for (int a = 0; a < 3; ++a) m1h[a] = m1n[a] = m1sp[a] = m1w[a] = 0.0;
#pragma unroll
        for (int i = 0; i < NS; ++i)
#pragma unroll
            for (int k = 0; k < 6; ++k) m2[i][k] = 0.0;
        m0n = m01 = 0.0;
        const int nq = QSTD ? 3 : md->n_qp;
        if (CACHED) {
            // compile-time quadrature index: the cached values stay in registers
#pragma unroll
            for (int q = 0; q < NQC; ++q) {
                if (q >= nq) break;
                double n[NS];
#pragma unroll
                for (int i = 0; i < NS; ++i) n[i] = nq_c[i][q];
                point(md, row, q, n, up_c[row < NS ? row : 0][q], sc, ext);
            }
        } else {
            for (int q = 0; q < nq; ++q) {
                const double xq = md->qp_x[q], yq = md->qp_y[q], p0 = 1.0 - xq - yq;
                double n[NS];
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    const double ui = Uc[0][i] * p0 + Uc[1][i] * xq + Uc[2][i] * yq;
                    n[i] = lin ? ui : exp(ui);
                }
                const int s = row < NS ? row : 0;
                const double u = Uc[0][s] * p0 + Uc[1][s] * xq + Uc[2][s] * yq;
                const double u_part = sc.c_new * u + (Hc[0][s] * p0 + Hc[1][s] * xq + Hc[2][s] * yq);
                point(md, row, q, n, u_part, sc, ext);
            }
        }
    }

    // contribution of quadrature point q to the moments of equation row `row`
    __device__ __forceinline__ void point(const fedm_model_desc *__restrict__ md, int row, int q,
                                          const double n[NS], double u_part, const StepCoef sc,
                                          const double *const ext[NS]) {
        const double two_pi = 6.283185307179586476925286766559;
        const bool phi_row = PO && row == IPHI;
        const double xq = qx(md, q), yq = qy(md, q);
        const double phi[3] = {1.0 - xq - yq, xq, yq};
        const double rq = rn[0] * phi[0] + rn[1] * phi[1] + rn[2] * phi[2];
        const double W = qw(md, q) * detJ * two_pi * rq;
        double g[NS], h = 0.0, sp = 0.0;
#pragma unroll
        for (int i = 0; i < NS; ++i) g[i] = 0.0;
        if (phi_row) {
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const double cz = md->Z[i] * n[i] * md->charge_over_eps;
                h -= cz;
                if (full) g[i] = lin ? -md->Z[i] * md->charge_over_eps : -cz;   // d n_i / d u_i = 1 (linear) or n_i
            }
        } else if (full) {
            const int s = row < NS ? row : 0;
            if (lin) {  // expu_or_1 = 1, fedm/functions.py:352
                h = u_part * sc.inv_dt;
                g[s] = sc.c_new * sc.inv_dt;
            } else {
                h = n[s] * u_part * sc.inv_dt;
                g[s] = n[s] * (u_part + sc.c_new) * sc.inv_dt;
            }
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                if (j >= nreac) break;
                const double nu = (double)md->net[j][s];
                if (nu == 0.0) continue;
                double prod = 1.0;
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    const int P = md->power[j][i];
                    for (int e = 0; e < P; ++e) prod *= n[i];
                }
                h -= nu * kv[j] * prod;
                sp += nu * kd[j] * prod;
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    const int P = md->power[j][i];
                    if (!P) continue;
                    if (!lin) {
                        g[i] -= nu * kv[j] * (double)P * prod;
                    } else {  // d prod / d u_i = P u_i^(P-1) prod_(k != i) u_k^P_k
                        double dp = (double)P;
                        for (int e = 1; e < P; ++e) dp *= n[i];
#pragma unroll
                        for (int k2 = 0; k2 < NS; ++k2) {
                            if (k2 == i) continue;
                            for (int e = 0; e < md->power[j][k2]; ++e) dp *= n[k2];
                        }
                        g[i] -= nu * kv[j] * dp;
                    }
                }
            }
            const int nn = md->ext_nodes[s];
            if (nn && ext[s]) {
                double f = 0.0;
                for (int m = 0; m < nn; ++m) f += ext[s][m] * md->ext_B[q][m];
                h -= f;
            }
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) m1h[a] += W * phi[a] * h;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = a; b < 3; ++b) {
                const double pp = W * phi[a] * phi[b];
#pragma unroll
                for (int i = 0; i < NS; ++i) m2[i][sym6(a, b)] += pp * g[i];
            }
        m01 += W;
        if (!phi_row) {
            const double ns_ = n[row < NS ? row : 0];
            m0n += W * ns_;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                m1n[a] += W * phi[a] * ns_;
                m1sp[a] += W * phi[a] * sp;
                m1w[a] += W * phi[a];
            }
        }
    }

    // shared factors of the row's entries -- after row_moments(row), before residual/block_row.
    // Logarithmic representation: Gamma = n vel, vel = -D grad(u) + Z mu E.  Linear: Gamma =
    // -D grad(u) + Z mu E u, so the diffusive part carries sum W (m01) instead of sum W n (m0n) and
    // d n / d u_b is phi_b instead of n phi_b (m1w instead of m1n).
    double PG[3], QG[3], Dcur, zmucur;   // grad(u).G_a, E.G_a (or w.G_a for a constant drift), D, Z mu (linear form)
    __device__ __forceinline__ void row_prepare(const fedm_model_desc *__restrict__ md, int row) {
        if (PO && row == IPHI) return;
        const int s = row < NS ? row : 0;
        DGm = 0.0;
        Kd = 0.0;
        Dcur = zmucur = 0.0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            velG[a] = 0.0;
            T[a] = m1sp[a];
            PG[a] = QG[a] = 0.0;
        }
        if (!flux[s]) return;
        const double zmud = fdrift[s] ? md->Z[s] * mud[s] : 0.0;
        if (!lin) {
            DGm = Dv[s] * m0n;
            if (fdrift[s]) Kd = md->Z[s] * muv[s] * m0n;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                velG[a] = vel[s][0] * G[a][0] + vel[s][1] * G[a][1];
                if (PO) {
                    const double Pa = gradu[s][0] * G[a][0] + gradu[s][1] * G[a][1];
                    const double Qa = E[0] * G[a][0] + E[1] * G[a][1];
                    T[a] += (zmud * Qa - Dd[s] * Pa) * m0n;
                }
            }
            return;
        }
        Dcur = Dv[s];
        DGm = Dv[s] * m01;
        const bool wdrift = md->eq_type[s] == FEDM_EQ_DRIFT_DIFFUSION_REACTION && md->has_drift_w[s];
        zmucur = fdrift[s] ? md->Z[s] * muv[s] : (wdrift ? 1.0 : 0.0);
        if (fdrift[s]) Kd = md->Z[s] * muv[s] * m0n;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            PG[a] = gradu[s][0] * G[a][0] + gradu[s][1] * G[a][1];
            QG[a] = fdrift[s] ? E[0] * G[a][0] + E[1] * G[a][1]
                              : (wdrift ? md->drift_w[s][0] * G[a][0] + md->drift_w[s][1] * G[a][1] : 0.0);
            // integral of Gamma . G_a = -D P_a sum W + Z mu Q_a sum W u
            velG[a] = -Dcur * PG[a] * m01 + zmucur * QG[a] * m0n;
            if (PO) T[a] += zmud * QG[a] * m0n - Dd[s] * PG[a] * m01;
        }
    }

    // residual entry (a, row)
    __device__ __forceinline__ double residual(int row, int a) const {
        if (PO && row == IPHI) return (gradPhi[0] * G[a][0] + gradPhi[1] * G[a][1]) * m01 + m1h[a];
        return lin ? m1h[a] - velG[a] : m1h[a] - velG[a] * m0n;
    }

    // entries d R[a][row] / d U[b][0..NEQ)
    __device__ __forceinline__ void block_row(const fedm_model_desc *__restrict__, int row, int a,
                                              int b, double B[NEQ]) const {
        const int k = sym6(a, b);
#pragma unroll
        for (int i = 0; i < NS; ++i) B[i] = m2[i][k];
        if (PO && row == IPHI) {
            B[IPHI] = gg[k] * m01;
            return;
        }
        const int s = row < NS ? row : 0;
        if (lin) B[s] += DGm * gg[k] - zmucur * QG[a] * m1w[b];
        else B[s] += DGm * gg[k] - velG[a] * m1n[b];
        if (PO) B[IPHI] = Kd * gg[k] - dE[b] * T[a];
    }
};

// ---------------------------------------------------------------------------------------------
// Neumann boundary flux of one tagged facet (fedm/functions.py:523-524):
//   2*pi * (Z mu E.n) exp(u) v r ds     on the edge opposite local vertex `fi`.
// Contributions are added with global fp64 atomics (a few thousand facets; adjacent facets
// share a vertex).  R: [3][NEQ] residual, blocks via `add(a, b, s_row, s_col, value)`.
// ---------------------------------------------------------------------------------------------
template <int NS, class AddR, class AddJ>
__device__ void boundary_facet(const fedm_model_desc *__restrict__ md, const double x[3][2],
                               const double Uc[3][NS + 1], int fi, int tag, bool jacobian,
                               AddR addR, AddJ addJ) {
    constexpr int NEQ = NS + 1;
    const double two_pi = 6.283185307179586476925286766559;
    CellGeom cg;
    cg.init(x, md->axisymmetric);
    double E[2] = {0.0, 0.0};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        E[0] -= Uc[a][NS] * cg.G[a][0];
        E[1] -= Uc[a][NS] * cg.G[a][1];
    }
    const double Em = sqrt(E[0] * E[0] + E[1] * E[1]);
    const double lnE = log(Em);
    double dEm[3];
#pragma unroll
    for (int b = 0; b < 3; ++b) dEm[b] = -(E[0] * cg.G[b][0] + E[1] * cg.G[b][1]) / Em;
    const int j = (fi == 0) ? 1 : 0, k = (fi == 2) ? 1 : 2;
    const double gi = sqrt(cg.G[fi][0] * cg.G[fi][0] + cg.G[fi][1] * cg.G[fi][1]);
    const double nrm[2] = {-cg.G[fi][0] / gi, -cg.G[fi][1] / gi};
    const double ex = x[j][0] - x[k][0], ey = x[j][1] - x[k][1];
    const double L = sqrt(ex * ex + ey * ey);
    const double En = E[0] * nrm[0] + E[1] * nrm[1];
    for (int s = 0; s < NS; ++s) {
        if (md->eq_type[s] != FEDM_EQ_DRIFT_DIFFUSION_REACTION || md->has_drift_w[s] ||
            md->bc_kind[tag - 1][s] != FEDM_BC_NEUMANN)
            continue;
        double muv, mud;
        termsum_eval(md->mu[s], Em, 1.0 / Em, lnE, muv, mud);
        double EM1[3] = {0.0, 0.0, 0.0}, EM2[3][3] = {{0.0}};
        for (int t = 0; t < md->n_fqp; ++t) {
            double phi[3] = {0.0, 0.0, 0.0};
            phi[j] = 1.0 - md->fqp_t[t];
            phi[k] = md->fqp_t[t];
            const double rq = cg.rn[0] * phi[0] + cg.rn[1] * phi[1] + cg.rn[2] * phi[2];
            const double uq = Uc[0][s] * phi[0] + Uc[1][s] * phi[1] + Uc[2][s] * phi[2];
            const bool lin = md->linear_representation != 0;
            const double n = lin ? uq : exp(uq);
            const double W0 = md->fqp_w[t] * L * two_pi * rq;
            const double We = W0 * n, Wd = lin ? W0 : We;   // d n / d u_b = phi_b (linear) or n phi_b
            for (int a = 0; a < 3; ++a) {
                EM1[a] += We * phi[a];
                for (int b = 0; b < 3; ++b) EM2[a][b] += Wd * phi[a] * phi[b];
            }
        }
        const double zm = md->Z[s] * muv, zd = md->Z[s] * mud;
        for (int a = 0; a < 3; ++a) {
            if (a == fi) continue;  // phi_a vanishes on the facet
            addR(a, s, zm * En * EM1[a]);
            if (!jacobian) continue;
            for (int b = 0; b < 3; ++b) {
                addJ(a, b, s, s, zm * En * EM2[a][b]);
                const double gbn = cg.G[b][0] * nrm[0] + cg.G[b][1] * nrm[1];
                addJ(a, b, s, NEQ - 1, (zd * dEm[b] * En - zm * gbn) * EM1[a]);
            }
        }
    }
}

}  // namespace fedm
