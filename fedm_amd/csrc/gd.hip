// LMEA (glow-discharge) model family on the device: residual and exact Jacobian of
//   * particle balance equations in log variables with nodal, semi-implicitly linearised
//     transport/rate coefficients            fedm/functions.py:350-368, 753-774, 777-843
//   * the electron energy balance with Joule heating      fedm/functions.py:845-912,
//                                                         examples/glow_discharge/fedm-gd.py:354-359
//   * 'flux source' walls with reflection and secondary emission    fedm/functions.py:514-522
//   * Poisson                                              fedm/functions.py:401
// The residual is written once on "value + spatial gradient" objects over a forward-mode
// dual scalar; the element Jacobian is obtained column by column (one dual direction per
// local dof), which is exact like UFL's `derivative` (fedm-gd.py:402) without a hand
// derivation of the many cross terms.  Cells of one colour per launch (conflict-free
// read-modify-write, bitwise reproducible); one thread per cell for the residual, one thread
// per (cell, local dof) for the Jacobian -- a colour alone has too few cells to fill the chip.
#include "fedm_internal.hpp"

#include <algorithm>
#include <cstdlib>
#include <vector>

namespace fedm {

// -DFEDM_GD_ROW_TIMING: how long the waves of gd_jacobian_rows_kernel spend on each equation row (100 MHz wall clock,
// lane 0 of every wave, summed over the workgroups; [NEQ]: the set-up in front of the rows): tools/gd_row_time.py
#ifdef FEDM_GD_ROW_TIMING
__device__ unsigned long long g_gd_row[8];
extern "C" void fedm_debug_gd_rows(unsigned long long *out, int reset) {
    hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gd_row), sizeof(unsigned long long) * 8);
    if (reset) {
        unsigned long long z[8] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(g_gd_row), z, sizeof(z));
    }
}
#define GD_ROW_T(k) if ((threadIdx.x & 63) == 0) { const unsigned long long now_ = wall_clock64(); atomicAdd(&g_gd_row[k], now_ - t_prev_); t_prev_ = now_; }
#else
#define GD_ROW_T(k)
#endif

struct Dual {
    double v, d;
};
__device__ __forceinline__ Dual mk(double v) { return {v, 0.0}; }
__device__ __forceinline__ Dual operator+(Dual a, Dual b) { return {a.v + b.v, a.d + b.d}; }
__device__ __forceinline__ Dual operator-(Dual a, Dual b) { return {a.v - b.v, a.d - b.d}; }
__device__ __forceinline__ Dual operator-(Dual a) { return {-a.v, -a.d}; }
__device__ __forceinline__ Dual operator*(Dual a, Dual b) { return {a.v * b.v, a.d * b.v + a.v * b.d}; }
__device__ __forceinline__ Dual operator*(double a, Dual b) { return {a * b.v, a * b.d}; }
__device__ __forceinline__ Dual operator*(Dual a, double b) { return {a.v * b, a.d * b}; }
__device__ __forceinline__ Dual operator+(Dual a, double b) { return {a.v + b, a.d}; }
__device__ __forceinline__ Dual operator-(Dual a, double b) { return {a.v - b, a.d}; }
__device__ __forceinline__ Dual operator/(Dual a, Dual b) {
    const double inv = 1.0 / b.v;
    return {a.v * inv, (a.d - a.v * inv * b.d) * inv};
}
__device__ __forceinline__ Dual dexp(Dual a) {
    const double e = exp(a.v);
    return {e, e * a.d};
}
__device__ __forceinline__ Dual dsqrt(Dual a) {
    const double s = sqrt(a.v);
    return {s, 0.5 * a.d / s};
}
__device__ __forceinline__ Dual dabs(Dual a) { return a.v < 0.0 ? -a : a; }

// value and spatial gradient of a scalar field at a quadrature point
struct SG {
    Dual v, gx, gy;
};
__device__ __forceinline__ SG operator+(SG a, SG b) { return {a.v + b.v, a.gx + b.gx, a.gy + b.gy}; }
__device__ __forceinline__ SG operator-(SG a, SG b) { return {a.v - b.v, a.gx - b.gx, a.gy - b.gy}; }
__device__ __forceinline__ SG operator*(SG a, SG b) {
    return {a.v * b.v, a.gx * b.v + a.v * b.gx, a.gy * b.v + a.v * b.gy};
}
__device__ __forceinline__ SG operator*(SG a, double b) { return {a.v * b, a.gx * b, a.gy * b}; }
__device__ __forceinline__ SG operator/(SG a, SG b) {
    const Dual inv = mk(1.0) / b.v;
    return {a.v * inv, (a.gx - a.v * inv * b.gx) * inv, (a.gy - a.v * inv * b.gy) * inv};
}
__device__ __forceinline__ SG sexp(SG a) {
    const Dual e = dexp(a.v);
    return {e, e * a.gx, e * a.gy};
}

constexpr int GS = FEDM_GD_MAX_SPECIES;
constexpr int GR = FEDM_GD_MAX_REACTIONS;

struct GdCell {
    double G[3][2], detJ, rn[3], x[3][2];
    int v[3];
};

// nodal field f (index `fi` in the field table) at the point with P1 weights phi
__device__ __forceinline__ SG nodal_sg(const double *__restrict__ fields, int nv, int fi,
                                       const GdCell &c, const double phi[3]) {
    const double *f = fields + (size_t)fi * nv;
    const double a0 = f[c.v[0]], a1 = f[c.v[1]], a2 = f[c.v[2]];
    SG r;
    r.v = mk(a0 * phi[0] + a1 * phi[1] + a2 * phi[2]);
    r.gx = mk(a0 * c.G[0][0] + a1 * c.G[1][0] + a2 * c.G[2][0]);
    r.gy = mk(a0 * c.G[0][1] + a1 * c.G[1][1] + a2 * c.G[2][1]);
    return r;
}

template <int NEQ>
__device__ __forceinline__ SG unknown_sg(const Dual Uc[3][NEQ], int comp, const GdCell &c,
                                         const double phi[3]) {
    SG r;
    r.v = Uc[0][comp] * phi[0] + Uc[1][comp] * phi[1] + Uc[2][comp] * phi[2];
    r.gx = Uc[0][comp] * c.G[0][0] + Uc[1][comp] * c.G[1][0] + Uc[2][comp] * c.G[2][0];
    r.gy = Uc[0][comp] * c.G[0][1] + Uc[1][comp] * c.G[1][1] + Uc[2][comp] * c.G[2][1];
    return r;
}

// drift-diffusion flux, fedm/functions.py:219-237
__device__ __forceinline__ void gd_flux(double sign, SG ulog, SG D, SG mu, Dual Ex, Dual Ey,
                                        bool grad_diffusion, Dual &gx, Dual &gy) {
    const SG ue = sexp(ulog);
    if (grad_diffusion) {
        const SG De = D * ue;
        gx = -De.gx;
        gy = -De.gy;
    } else {
        gx = -(D.v * ue.gx);
        gy = -(D.v * ue.gy);
    }
    gx = gx + sign * (mu.v * Ex * ue.v);
    gy = gy + sign * (mu.v * Ey * ue.v);
}

struct GdPoint {  // everything the integrands need at one point
    SG u[GS + 1];        // unknowns: 0 energy, 1..ns-1 species, ns potential
    SG mu[GS], D[GS];    // semi-implicit coefficients per species
    Dual n[GS];          // densities exp(u_i), i >= 1
    Dual Ex, Ey;
    SG me;               // current mean-energy Function (thermal velocity of electrons)
};

// (the species count is NEQ - 1: loops over it unroll and the per-species arrays stay in registers)
template <int NEQ>
__device__ __forceinline__ void gd_point(const fedm_gd_desc *__restrict__ md, const double *__restrict__ fields,
                                         int nv, const GdCell &c, const Dual Uc[3][NEQ], const double phi[3],
                                         GdPoint &p, SG &dme_out) {
    constexpr int ns = NEQ - 1;
    const int nr = md->n_reactions;
#pragma unroll
    for (int i = 0; i < NEQ; ++i) p.u[i] = unknown_sg<NEQ>(Uc, i, c, phi);
    const int F_MU = 0, F_D = ns, F_MUD = 2 * ns, F_DD = 3 * ns, F_K = 4 * ns, F_KD = 4 * ns + nr,
              F_MEO = 4 * ns + 2 * nr, F_ME = F_MEO + 1, F_UEO = F_MEO + 2;
    const SG meo = nodal_sg(fields, nv, F_MEO, c, phi);
    p.me = nodal_sg(fields, nv, F_ME, c, phi);
    const SG ueo = nodal_sg(fields, nv, F_UEO, c, phi);
    p.Ex = -p.u[NEQ - 1].gx;
    p.Ey = -p.u[NEQ - 1].gy;
#pragma unroll
    for (int i = 1; i < ns; ++i) p.n[i] = dexp(p.u[i].v);
    // mean energy of the new state, fedm-gd.py:215
    const SG dme = (sexp(p.u[0]) - sexp(p.u[ns - 1]) * meo) / sexp(ueo);
#pragma unroll
    for (int i = 0; i < ns; ++i) {
        p.mu[i] = nodal_sg(fields, nv, F_MU + i, c, phi) + nodal_sg(fields, nv, F_MUD + i, c, phi) * dme;
        p.D[i] = nodal_sg(fields, nv, F_D + i, c, phi) + nodal_sg(fields, nv, F_DD + i, c, phi) * dme;
    }
    dme_out = dme;
    (void)nr;
    (void)F_K;
    (void)F_KD;
}

// semi-implicit rate coefficient of reaction j at the point (value part only is used)
__device__ __forceinline__ Dual gd_rate_coefficient(const double *__restrict__ fields, int nv, int ns, int nr,
                                                    int j, const GdCell &c, const double phi[3], const SG &dme) {
    const int F_K = 4 * ns, F_KD = 4 * ns + nr;
    const double *fk = fields + (size_t)(F_K + j) * nv, *fd = fields + (size_t)(F_KD + j) * nv;
    const double kv = fk[c.v[0]] * phi[0] + fk[c.v[1]] * phi[1] + fk[c.v[2]] * phi[2];
    const double kd = fd[c.v[0]] * phi[0] + fd[c.v[1]] * phi[1] + fd[c.v[2]] * phi[2];
    return mk(kv) + kd * dme.v;
}

// element residual R[a][comp] for the given (dual) nodal unknowns
template <int NEQ>
__device__ void gd_element(const fedm_gd_desc *__restrict__ md, const double *__restrict__ fields,
                           int nv, const GdCell &c, const Dual Uc[3][NEQ], const double Uo[3][NEQ],
                           const double Uo1[3][NEQ], double dt, double dt_old,
                           const int8_t tags[3], int mode, Dual R[3][NEQ]) {
    const double two_pi = 6.283185307179586476925286766559;
    constexpr int ns = NEQ - 1, IPHI = NEQ - 1;
    const int nr = md->n_reactions;
    const double tr = dt / dt_old, trp1 = 1.0 + tr, tr2p1 = 1.0 + 2.0 * tr;
    for (int a = 0; a < 3; ++a)
        for (int s = 0; s < NEQ; ++s) R[a][s] = mk(0.0);
    const bool full = (mode == 0);

    for (int q = 0; q < md->n_qp; ++q) {
        const double phi[3] = {1.0 - md->qp_x[q] - md->qp_y[q], md->qp_x[q], md->qp_y[q]};
        const double rq = c.rn[0] * phi[0] + c.rn[1] * phi[1] + c.rn[2] * phi[2];
        const double W = md->qp_w[q] * c.detJ * two_pi * rq;
        GdPoint p;
        SG dme;
        gd_point<NEQ>(md, fields, nv, c, Uc, phi, p, dme);
        // Poisson row
        Dual rho = mk(0.0);
#pragma unroll
        for (int i = 1; i < ns; ++i) rho = rho + (md->sign[i] * md->charge_over_eps) * p.n[i];
        for (int a = 0; a < 3; ++a)
            R[a][IPHI] = R[a][IPHI] + W * ((p.u[IPHI].gx * c.G[a][0] + p.u[IPHI].gy * c.G[a][1]) - rho * phi[a]);
        if (!full) continue;
        // reaction rates, Source_term / Energy_Source_term (functions.py:835-843, 901-912)
        Dual f[ns], f_en = mk(0.0);
#pragma unroll
        for (int i = 0; i < ns; ++i) f[i] = mk(0.0);
        // Energy_Source_term's mean_energy where the scripts pass u[0] / u[n - 1] (fedm-gd.py:358): the sentinel losses
        Dual me_arg = mk(0.0);
        if (md->mean_energy_form == FEDM_GD_ME_UNKNOWN_RATIO) me_arg = p.u[0].v / p.u[ns - 1].v;
        for (int j = 0; j < nr; ++j) {
            Dual rate = gd_rate_coefficient(fields, nv, ns, nr, j, c, phi, dme);
#pragma unroll
            for (int i = 0; i < ns; ++i)
                for (int e = 0; e < md->power[j][i]; ++e) rate = (i == 0) ? rate * md->N0 : rate * p.n[i];
#pragma unroll
            for (int i = 0; i < ns; ++i) f[i] = f[i] + (double)md->net[j][i] * rate;
            const double lj = md->energy_loss[j];   // functions.py:905-911
            if (lj > 7e77 && lj < 8e77) f_en = f_en - (mk(md->energy_Ei) - me_arg) * rate;
            else if (lj > 9e99 && lj < 1e100) f_en = f_en - me_arg * rate;
            else f_en = f_en - lj * rate;
        }
        constexpr int ie = ns - 1;  // electrons are the last species
        Dual gex, gey;
        gd_flux(md->sign[ie], p.u[ie], p.D[ie], p.mu[ie], p.Ex, p.Ey, md->grad_diffusion[ie] != 0, gex, gey);
        f_en = f_en - (gex * p.Ex + gey * p.Ey);  // Joule heating, fedm-gd.py:359
#pragma unroll
        for (int comp = 0; comp < ns; ++comp) {
            // comp 0: energy equation with 5/3 of the electron coefficients (fedm-gd.py:354)
            const int sp = (comp == 0) ? ie : comp;
            const int et = md->eq_type[sp];
            Dual gx = mk(0.0), gy = mk(0.0), src;
            if (comp == 0) {
                gd_flux(md->sign[ie], p.u[0], p.D[ie] * (5.0 / 3.0), p.mu[ie] * (5.0 / 3.0), p.Ex, p.Ey,
                        md->grad_diffusion[ie] != 0, gx, gy);
                src = f_en;
            } else {
                src = f[comp];
                if (et == FEDM_EQ_DRIFT_DIFFUSION_REACTION) {
                    if (comp == ie) {
                        gx = gex;
                        gy = gey;
                    } else {
                        gd_flux(md->sign[comp], p.u[comp], p.D[comp], p.mu[comp], p.Ex, p.Ey,
                                md->grad_diffusion[comp] != 0, gx, gy);
                    }
                } else if (et == FEDM_EQ_DIFFUSION_REACTION) {  // -grad(D exp(u)), functions.py:362-364
                    const SG De = p.D[comp] * sexp(p.u[comp]);
                    gx = -De.gx;
                    gy = -De.gy;
                }
            }
            const Dual ulog = p.u[comp].v;
            double uo = 0.0, uo1 = 0.0;
            for (int a = 0; a < 3; ++a) {
                uo += Uo[a][comp] * phi[a];
                uo1 += Uo1[a][comp] * phi[a];
            }
            const Dual u_part = (ulog * tr2p1 - (trp1 * trp1) * uo + (tr * tr) * uo1) * (1.0 / trp1);
            const Dual T = dexp(ulog) * u_part * (1.0 / dt);
            for (int a = 0; a < 3; ++a)
                R[a][comp] = R[a][comp] + W * (T * phi[a] - (gx * c.G[a][0] + gy * c.G[a][1]) - src * phi[a]);
        }
    }
    if (!full) return;

    // 'flux source' boundary facets, functions.py:514-522
    for (int i = 0; i < 3; ++i) {
        const int tag = tags[i];
        if (tag <= 0) continue;
        const int j = (i == 0) ? 1 : 0, kk = (i == 2) ? 1 : 2;
        const double gi = sqrt(c.G[i][0] * c.G[i][0] + c.G[i][1] * c.G[i][1]);
        const double nx = -c.G[i][0] / gi, ny = -c.G[i][1] / gi;
        const double ex = c.x[j][0] - c.x[kk][0], ey = c.x[j][1] - c.x[kk][1];
        const double L = sqrt(ex * ex + ey * ey);
        constexpr int ie = ns - 1;
        for (int t = 0; t < md->n_fqp; ++t) {
            double phi[3] = {0.0, 0.0, 0.0};
            phi[j] = 1.0 - md->fqp_t[t];
            phi[kk] = md->fqp_t[t];
            const double rq = c.rn[0] * phi[0] + c.rn[1] * phi[1] + c.rn[2] * phi[2];
            const double W = md->fqp_w[t] * L * two_pi * rq;
            GdPoint p;
            SG dme;
            gd_point<NEQ>(md, fields, nv, c, Uc, phi, p, dme);
            const Dual En = p.Ex * nx + p.Ey * ny;
            Dual ion = mk(0.0);  // Ion_flux = sum Max(Gamma_i . n, 0), fedm-gd.py:351
#pragma unroll
            for (int s = 1; s < ns; ++s) {
                if (!md->is_ion[s]) continue;
                Dual gx, gy;
                gd_flux(md->sign[s], p.u[s], p.D[s], p.mu[s], p.Ex, p.Ey, md->grad_diffusion[s] != 0, gx, gy);
                const Dual gn = gx * nx + gy * ny;
                ion = ion + (gn + dabs(gn)) * 0.5;
            }
            const Dual vth_e = dsqrt(md->vth_e_coef * p.me.v);
#pragma unroll
            for (int comp = 0; comp < ns; ++comp) {
                const int sp = (comp == 0) ? ie : comp;
                const double ref = md->ref[tag - 1][sp];
                const double fac = (1.0 - ref) / (1.0 + ref);
                const int et = md->eq_type[sp];
                Dual val = mk(0.0);
                if (et == FEDM_EQ_REACTION) continue;
                const double vth_s = (sp == ie) ? 0.0 : md->vth[sp];
                Dual vth = (sp == ie) ? vth_e : mk(vth_s);
                double mu_scale = 1.0, gam = md->gamma[tag - 1];
                if (comp == 0) {  // energy: 5/3 mu, 1.3333 vth, gamma * mean energy of secondaries
                    vth = vth * 1.3333;
                    mu_scale = 5.0 / 3.0;
                    gam = gam * md->we_secondary;
                }
                const Dual dens = dexp(p.u[comp].v);
                if (et == FEDM_EQ_DIFFUSION_REACTION) {
                    val = fac * (0.5 * vth * dens);
                } else {
                    val = fac * ((0.5 * vth + dabs((md->sign[sp] * mu_scale) * (p.mu[sp].v * En))) * dens);
                    if (sp == ie) val = val - (2.0 * gam / (1.0 + ref)) * ion;
                }
                R[j][comp] = R[j][comp] + (W * phi[j]) * val;
                R[kk][comp] = R[kk][comp] + (W * phi[kk]) * val;
            }
        }
    }
}

template <int NEQ>
__global__ __launch_bounds__(128) void gd_assemble_kernel(
    const fedm_gd_desc *__restrict__ md, const double *__restrict__ fields, int nv,
    const int *__restrict__ cell_list, int n_cells, const int *__restrict__ cells,
    const double *__restrict__ coords, const int8_t *__restrict__ ftags,
    const uint32_t *__restrict__ cell_slots, const double *__restrict__ u,
    const double *__restrict__ uold, const double *__restrict__ uold1, double dt, double dt_old,
    double *__restrict__ val, double *__restrict__ F, int jacobian, int mode) {
    constexpr int NEQ2 = NEQ * NEQ, NCOL = 3 * NEQ;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int ci = jacobian ? t / NCOL : t;       // cell within the colour
    const int col = jacobian ? t - ci * NCOL : 0;  // local dof = element-matrix column
    if (ci >= n_cells) return;
    const int cidx = cell_list[ci];
    GdCell c;
    int8_t tags[3];
    Dual Uc[3][NEQ];
    double Uo[3][NEQ], Uo1[3][NEQ];
    for (int a = 0; a < 3; ++a) {
        c.v[a] = cells[3 * cidx + a];
        tags[a] = ftags[3 * cidx + a];
        c.x[a][0] = coords[2 * c.v[a]];
        c.x[a][1] = coords[2 * c.v[a] + 1];
        c.rn[a] = md->axisymmetric ? c.x[a][0] : 0.5 / 3.14159265358979323846;
        for (int s = 0; s < NEQ; ++s) {
            Uc[a][s] = mk(u[(size_t)c.v[a] * NEQ + s]);
            Uo[a][s] = uold[(size_t)c.v[a] * NEQ + s];
            Uo1[a][s] = uold1[(size_t)c.v[a] * NEQ + s];
        }
    }
    {
        const double d1x = c.x[1][0] - c.x[0][0], d1y = c.x[1][1] - c.x[0][1];
        const double d2x = c.x[2][0] - c.x[0][0], d2y = c.x[2][1] - c.x[0][1];
        const double det = d1x * d2y - d1y * d2x;
        c.detJ = fabs(det);
        c.G[0][0] = (c.x[1][1] - c.x[2][1]) / det;
        c.G[0][1] = (c.x[2][0] - c.x[1][0]) / det;
        c.G[1][0] = (c.x[2][1] - c.x[0][1]) / det;
        c.G[1][1] = (c.x[0][0] - c.x[2][0]) / det;
        c.G[2][0] = (c.x[0][1] - c.x[1][1]) / det;
        c.G[2][1] = (c.x[1][0] - c.x[0][0]) / det;
    }
    Dual R[3][NEQ];
    if (!jacobian) {
        gd_element<NEQ>(md, fields, nv, c, Uc, Uo, Uo1, dt, dt_old, tags, mode, R);
        for (int a = 0; a < 3; ++a)
            for (int s = 0; s < NEQ; ++s) F[(size_t)c.v[a] * NEQ + s] += R[a][s].v;
        return;
    }
    // one dual direction per local dof -> one column of the element Jacobian per thread
    const int b = col / NEQ, sc = col - b * NEQ;
    const bool carries_residual = (b == 0 && sc == NEQ - 1);  // any pass has the values
    if (mode == 1 && sc != NEQ - 1) return;  // Poisson-only: potential columns
#pragma unroll
    for (int bb = 0; bb < 3; ++bb)
#pragma unroll
        for (int s = 0; s < NEQ; ++s) Uc[bb][s].d = (bb == b && s == sc) ? 1.0 : 0.0;
    gd_element<NEQ>(md, fields, nv, c, Uc, Uo, Uo1, dt, dt_old, tags, mode, R);
    for (int a = 0; a < 3; ++a) {
        const uint32_t slot = cell_slots[(size_t)cidx * 9 + a * 3 + b];
        double *dst = val + ((size_t)(slot >> 6) * NEQ2) * SLICE + (slot & 63);
        for (int sr = 0; sr < NEQ; ++sr) dst[(size_t)(sr * NEQ + sc) * SLICE] += R[a][sr].d;
        if (carries_residual)
            for (int sr = 0; sr < NEQ; ++sr) F[(size_t)c.v[a] * NEQ + sr] += R[a][sr].v;
    }
}


// =============================================================================================
// Hand-derived element Jacobian.
//
// At a point the integrand of equation row `row` is  W (S phi_a - Gx G_a,x - Gy G_a,y)  with
// S = T - source (time term minus sources) and G the flux; S, Gx, Gy are functions of the point values
// and gradients of the unknowns.  Their partial derivatives are written down by hand -- the
// dual-number kernel above is the cross-check (tests: both against the oracle) -- and combined into
// DIRECTIONAL derivatives along the basis function of one column vertex b: for column field s the
// triple (tS, tX, tY) = d(S, Gx, Gy)/dU[b][s], so that
//   dR[a] / dU[b][s] = W (tS phi_a - tX G_a,x - tY G_a,y).
// The semi-implicit coefficients depend on u_0 (energy) and u_e (electrons) through the change of
// the mean energy dme = (exp(u_0) - exp(u_e) me_old) / exp(u_e,old)  (value and gradient: three
// "channels" c1, c2, c3); a derivative with respect to a channel reaches the columns 0 and e through
// the channels' own directional derivatives (GdChannels::k*).
// Work split: a workgroup takes 64 cells and one column vertex b; its waves are the equation rows
// (wave-uniform row: scalar branches, no lane evaluates another row's integrands).  The nodal
// coefficient fields and unknowns of the 64 cells are staged in LDS by all waves together.  One
// evaluation of the point functions per (cell, row, b, point) instead of fifteen dual passes of the
// whole element per cell and fifteen threads holding 512 registers each.
// All cells in one launch (the cells of one colour do not fill the chip: a launch per colour lasts as
// long as its slowest thread, nine times per assembly).
// =============================================================================================
struct N3 {
    double v, gx, gy;
};
// `fl`: the cell's column of the staged field values [field][vertex][cell]: field fi at vertex a is
// fl[(3 fi + a) * SLICE] (lanes = cells: consecutive LDS words, no bank conflicts)
__device__ __forceinline__ N3 nodal3(const double *fl, int fi, const GdCell &c, const double phi[3]) {
    const double a0 = fl[(3 * fi) * SLICE], a1 = fl[(3 * fi + 1) * SLICE], a2 = fl[(3 * fi + 2) * SLICE];
    return {a0 * phi[0] + a1 * phi[1] + a2 * phi[2], a0 * c.G[0][0] + a1 * c.G[1][0] + a2 * c.G[2][0],
            a0 * c.G[0][1] + a1 * c.G[1][1] + a2 * c.G[2][1]};
}
__device__ __forceinline__ double nodal1(const double *fl, int fi, const double phi[3]) {
    return fl[(3 * fi) * SLICE] * phi[0] + fl[(3 * fi + 1) * SLICE] * phi[1] + fl[(3 * fi + 2) * SLICE] * phi[2];
}

// dme = (c1; c2, c3) and the derivatives of the three channels along the basis function of the
// column vertex for the columns 0 (energy) and e (electrons)
struct GdChannels {
    double c1, c2, c3;
    // d c1 / d v_0 = d c2 / d gx_0 = d c3 / d gy_0 = r0;  d c1 / d v_e = d c2 / d gx_e = d c3 / d gy_e = rem
    double r0, rem, c2_v0, c3_v0, c2_ve, c3_ve;
};

// drift-diffusion flux of one species (gd_flux above) with its partial derivatives
struct GdFluxD {
    double Gx, Gy;      // dG/dv_own = G itself
    double dg;          // dGx/dgx_own = dGy/dgy_own
    double dE;          // dGx/dEx = dGy/dEy
    double x_c1, x_c2;  // dGx/dc1, dGx/dc2
    double y_c1, y_c3;  // dGy/dc1, dGy/dc3
};
// coefficient fields f_mu, f_d (values) and f_mud, f_dd (derivatives with respect to the mean
// energy); scale: 5/3 for the energy flux
__device__ __forceinline__ GdFluxD gd_flux_partials(const double *fl, int f_mu, int f_mud, int f_d, int f_dd,
                                                    const GdCell &c, const double phi[3], const GdChannels &ch,
                                                    double sign, double scale, bool drift, bool grad_diffusion,
                                                    double v, double gx, double gy, double Ex, double Ey,
                                                    double e_known = -1.0) {
    const N3 Da = nodal3(fl, f_d, c, phi), Db = nodal3(fl, f_dd, c, phi);
    // D = Da + Db * dme as value + gradient (product rule of the SG type above)
    const double Dv = scale * (Da.v + Db.v * ch.c1);
    const double Dgx = scale * (Da.gx + Db.gx * ch.c1 + Db.v * ch.c2);
    const double Dgy = scale * (Da.gy + Db.gy * ch.c1 + Db.v * ch.c3);
    double muv = 0.0, mub = 0.0;
    if (drift) {
        mub = scale * nodal1(fl, f_mud, phi);
        muv = scale * nodal1(fl, f_mu, phi) + mub * ch.c1;
    }
    const double e = e_known >= 0.0 ? e_known : exp(v);   // exp(v) of the workgroup's table, when it has one
    GdFluxD F;
    const double zm = sign * muv * e;
    if (grad_diffusion) {
        F.Gx = -(Dgx * e + Dv * e * gx) + zm * Ex;
        F.Gy = -(Dgy * e + Dv * e * gy) + zm * Ey;
    } else {
        F.Gx = -(Dv * e * gx) + zm * Ex;
        F.Gy = -(Dv * e * gy) + zm * Ey;
    }
    F.dg = -Dv * e;
    F.dE = zm;
    const double dDv = scale * Db.v;   // dDv/dc1 (= dDgx/dc2 = dDgy/dc3)
    const double dmu = sign * mub * e;  // d(zm)/dc1
    F.x_c1 = -e * gx * dDv + dmu * Ex;
    F.y_c1 = -e * gy * dDv + dmu * Ey;
    F.x_c2 = 0.0;
    F.y_c3 = 0.0;
    if (grad_diffusion) {
        F.x_c1 -= e * scale * Db.gx;
        F.y_c1 -= e * scale * Db.gy;
        F.x_c2 = -e * dDv;
        F.y_c3 = -e * dDv;
    }
    return F;
}

// STORE: 2 = the element blocks go to `val` = the element buffer [(a * 3 + b) * NEQ * NEQ + row * NEQ + s]
// [cell] (lanes = cells: coalesced) and gd_gather_kernel sums them into the matrix -- every matrix
// value written once, fixed summation order; 1 = fp64 atomics straight into the matrix (the fall-back
// when the buffer cannot be allocated).  The residual is added with atomics.
// STORE 0: residual only (launched with one column vertex: gridDim.y = 1).
// NRC, NQC > 0: the numbers of reactions and of quadrature points at compile time (the loops over them
// unroll, the model's scalars -- weights, powers, points -- are loaded once instead of by a dependent
// scalar load and a wait in every pass of the innermost loops); 0: taken from the descriptor.
// BALL (round 4): the three column vertices side by side -- 45 + 9 accumulators, the point functions evaluated ONCE
// per quadrature point instead of once per column vertex.  That is the variant that spilled 600 bytes at two waves
// per SIMD; here it runs at ONE wave per SIMD with the whole 512-entry register file: a
// third of the instructions per wave for half the resident waves.
template <int NEQ, int STORE, int NRC = 0, int NQC = 0, bool BALL = false, int WV = 0>
__global__ __launch_bounds__(64 * (NEQ - 1)) __attribute__((amdgpu_waves_per_eu(BALL ? 1 : 2, BALL ? 1 : 2))) void gd_jacobian_rows_kernel(
    const fedm_gd_desc *__restrict__ md, const double *__restrict__ fields, int nv,
    const int *__restrict__ cell_list, int n_cells, const int *__restrict__ cells,
    const double *__restrict__ coords, const int8_t *__restrict__ ftags,
    const uint32_t *__restrict__ cell_slots, const double *__restrict__ u,
    const double *__restrict__ uold, const double *__restrict__ uold1, double dt, double dt_old,
    double *__restrict__ val, double *__restrict__ F, int mode, double *__restrict__ elemF, int exp_table) {
#ifdef FEDM_GD_ROW_TIMING
    unsigned long long t_prev_ = wall_clock64();
#endif
    constexpr int NEQ2 = NEQ * NEQ, ns = NEQ - 1, IPHI = NEQ - 1, ie = ns - 1;
    // waves of a workgroup: one per species/energy row (WV = 0), or WV waves that take the rows w, w + WV, ... in turn
    // (two for the glow-discharge model: the set-up is shared by two rows a wave, two workgroups fit a CU, and the
    // rows balance: energy + ion against metastables + electrons + Poisson)
    constexpr int NWV = WV > 0 ? WV : NEQ - 1;
    const double two_pi = 6.283185307179586476925286766559;
    const int lc = threadIdx.x & (SLICE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ci = blockIdx.x * SLICE + lc;
    const int nr = NRC > 0 ? NRC : md->n_reactions;
    const int nqp = NQC > 0 ? NQC : md->n_qp;
    const int NF = 4 * ns + 2 * nr + 3;
    extern __shared__ double gd_lds[];
    int *lds_vtx = reinterpret_cast<int *>(gd_lds);                 // [64][3] global vertex ids (-1: no cell)
    double *lds_f = gd_lds + (3 * SLICE + 1) / 2;                   // [64][NF][3]
    double *lds_u = lds_f + (size_t)SLICE * NF * 3;                 // [3][NEQ][64]
    // One workgroup is resident per CU (four waves of 400 registers), so nothing hides the latency of this set-up:
    // it is ONE round trip to memory after the cell's vertex ids -- every lane reads the ids of its own cell (the waves
    // redundantly: no LDS hop, no barrier), then the cell's coordinates, the history of the wave's row and the wave's
    // share of the (field, vertex) pairs are all requested before the first of them is used (12.9 of the
    // workgroup's 51 us were this set-up -- a dependent trip for the ids, per batch of fields, for the coordinates and for
    // the history -- now 8.2 of 45: tools/gd_row_time.py; F + J 290 -> 264 us at 200 k DOFs).
    (void)lds_vtx;
    int vt[3] = {-1, -1, -1};
    if (ci < n_cells) {
        const int cc = cell_list ? cell_list[ci] : ci;
#pragma unroll
        for (int a = 0; a < 3; ++a) vt[a] = cells[3 * cc + a];
    }
    double xy0[3][2], hist0[3][2];   // coordinates; (u_old, u_old1) of the wave's first row at the cell's vertices
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int v = vt[a] >= 0 ? vt[a] : 0;
        xy0[a][0] = coords[2 * v];
        xy0[a][1] = coords[2 * v + 1];
        hist0[a][0] = uold[(size_t)v * NEQ + wave];
        hist0[a][1] = uold1[(size_t)v * NEQ + wave];
    }
    {
        // lane = cell, the waves share the (field, vertex) pairs: no division by a run-time number
        const int nw = blockDim.x >> 6;
        constexpr int CH = 32;             // pairs requested before the first is stored (33 fields: 25 a wave)
        constexpr int CU = (3 * NEQ + NWV - 1) / NWV;           // the wave's share of the unknowns
        double *dstu = lds_u + lc;
        double tmpu[CU];
#pragma unroll
        for (int k = 0; k < CU; ++k) {
            const int f = wave + k * nw, a = f / NEQ, sidx = f - a * NEQ;
            const int vtx = a == 0 ? vt[0] : a == 1 ? vt[1] : vt[2];
            tmpu[k] = (f < 3 * NEQ && vtx >= 0) ? u[(size_t)vtx * NEQ + sidx] : 0.0;
        }
        double *dstf = lds_f + lc;
        for (int f0 = wave; f0 < NF * 3; f0 += nw * CH) {
            double tmp[CH];
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                const int f = f0 + k * nw, fi = f / 3, a = f - 3 * fi;
                const int vtx = a == 0 ? vt[0] : a == 1 ? vt[1] : vt[2];
                tmp[k] = (f < NF * 3 && vtx >= 0) ? fields[(size_t)fi * nv + vtx] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                const int f = f0 + k * nw;
                if (f < NF * 3) dstf[(size_t)f * SLICE] = tmp[k];
            }
        }
#pragma unroll
        for (int k = 0; k < CU; ++k) {
            const int f = wave + k * nw;
            if (f < 3 * NEQ) dstu[(size_t)f * SLICE] = tmpu[k];
        }
    }
    __syncthreads();
    // exp(u_s) of the species/energy unknowns and exp(-u_e,old) at the quadrature points depend on the
    // cell only: every row and every column-vertex pass needs them (eight exponentials per point and
    // pass, half of the kernel's vector instructions).  Wave s tabulates unknown s for its 64 cells
    // once: [point][unknown | NEQ-1: exp(-u_e,old)][cell].  exp_table = 0: no room in LDS, evaluate.
    double *lds_e = lds_u + (size_t)SLICE * 3 * NEQ;
    // per row: weight of reaction j (net gain of the species, minus the energy loss for row 0) and its
    // powers packed 4 bits per species (-1: the reaction does not enter the row)
    double *lds_rw = lds_e + (exp_table ? (size_t)nqp * NEQ * SLICE : 0);
    int *lds_rp = reinterpret_cast<int *>(lds_rw + ns * FEDM_GD_MAX_REACTIONS);
    if (lc < nr) {
        int pk = 0;
        for (int i = 0; i < ns; ++i) pk |= (md->power[lc][i] & 15) << (4 * i);
        for (int rw = wave; rw < ns; rw += NWV) {     // (the rows of this wave)
            const double w = (rw == 0) ? -md->energy_loss[lc] : (double)md->net[lc][rw];
            // energy row: a loss that is a sentinel of the decks (1: Ei - mean energy, 2: mean energy, functions.py:906-909)
            // takes the reaction out of the row's list and into a list of its own (bits 28-29: the kind), walked at the
            // quadrature points only when the model has such reactions -- the glow-discharge deck has none
            const int kind = rw != 0 ? 0 : (-w > 7e77 && -w < 8e77) ? 1 : (-w > 9e99 && -w < 1e100) ? 2 : 0;
            lds_rw[rw * FEDM_GD_MAX_REACTIONS + lc] = w;
            lds_rp[rw * FEDM_GD_MAX_REACTIONS + lc] = (w == 0.0 || kind) ? -1 : pk;
            if (rw == 0) lds_rp[ns * FEDM_GD_MAX_REACTIONS + lc] = kind ? (pk | kind << 28) : -1;
        }
    }
    if (!exp_table) __syncthreads();
    if (exp_table) {
        const double *flw = lds_f + lc, *ulw = lds_u + lc;
        const int f_ueo = 4 * ns + 2 * nr + 2;
        for (int q = 0; q < nqp; ++q) {
            const double p0 = 1.0 - md->qp_x[q] - md->qp_y[q], p1 = md->qp_x[q], p2 = md->qp_y[q];
            for (int sw = wave; sw < ns; sw += NWV)
                lds_e[((size_t)q * NEQ + sw) * SLICE + lc] = exp(ulw[sw * SLICE] * p0 + ulw[(NEQ + sw) * SLICE] * p1 + ulw[(2 * NEQ + sw) * SLICE] * p2);
            if (wave == 0)
                lds_e[((size_t)q * NEQ + ns) * SLICE + lc] =
                    exp(-(flw[(3 * f_ueo) * SLICE] * p0 + flw[(3 * f_ueo + 1) * SLICE] * p1 + flw[(3 * f_ueo + 2) * SLICE] * p2));
        }
        __syncthreads();
    }
    if (ci >= n_cells) return;
    const int cidx = cell_list ? cell_list[ci] : ci;   // no list: the mesh's own cell order
    const double *fl = lds_f + lc;   // (re-based at every quadrature point, see there)
    const double *ul = lds_u + lc;   // [a][s][cell]
    GdCell c;
    {
        double x[3][2];   // (not kept: the facet terms read the two coordinates they need again)
        for (int a = 0; a < 3; ++a) {
            c.v[a] = vt[a];
            x[a][0] = xy0[a][0];
            x[a][1] = xy0[a][1];
            c.rn[a] = md->axisymmetric ? x[a][0] : 0.5 / 3.14159265358979323846;
        }
        const double d1x = x[1][0] - x[0][0], d1y = x[1][1] - x[0][1];
        const double d2x = x[2][0] - x[0][0], d2y = x[2][1] - x[0][1];
        const double det = d1x * d2y - d1y * d2x;
        c.detJ = fabs(det);
        c.G[0][0] = (x[1][1] - x[2][1]) / det;
        c.G[0][1] = (x[2][0] - x[1][0]) / det;
        c.G[1][0] = (x[2][1] - x[0][1]) / det;
        c.G[1][1] = (x[0][0] - x[2][0]) / det;
        c.G[2][0] = (x[0][1] - x[1][1]) / det;
        c.G[2][1] = (x[1][0] - x[0][0]) / det;
    }
    const double tr = dt / dt_old, trp1 = 1.0 + tr, c_new = (1.0 + 2.0 * tr) / trp1;
    const int F_MU = 0, F_D = ns, F_MUD = 2 * ns, F_DD = 3 * ns, F_K = 4 * ns, F_KD = 4 * ns + nr,
              F_MEO = 4 * ns + 2 * nr, F_ME = F_MEO + 1, F_UEO = F_MEO + 2;
    // One wave per species/energy row; the (cheap) Poisson row is a second pass of wave 1, so that a
    // workgroup is NEQ - 1 waves (four for the glow-discharge model: two workgroups per CU at two waves
    // per SIMD; five-wave workgroups left three of eight wave slots empty).
    GD_ROW_T(NEQ)
    const int n_own = (ns - wave + NWV - 1) / NWV;                            // species/energy rows of this wave
    const int n_pass = n_own + (wave == (ns > 1 && NWV > 1 ? 1 : 0) ? 1 : 0);   // ... and the Poisson row
    for (int pass = 0; pass < n_pass; ++pass) {
    const int row = pass < n_own ? wave + pass * NWV : IPHI;
    if (mode == 1 && row != IPHI) continue;   // Poisson-only: the other rows are identity rows
    double Hrow[3];
    for (int a = 0; a < 3; ++a) {
        const double ho = pass == 0 ? hist0[a][0] : uold[(size_t)c.v[a] * NEQ + row];
        const double ho1 = pass == 0 ? hist0[a][1] : uold1[(size_t)c.v[a] * NEQ + row];
        Hrow[a] = (-(trp1 * trp1) * ho + (tr * tr) * ho1) / trp1;
    }
    // One column vertex b at a time (a rolled loop around the quadrature loop): 15 accumulators instead
    // of 45.  The point functions are evaluated once per column vertex -- more arithmetic, but the 45
    // accumulators next to the live coefficients did not fit the register file: 92 spilled dwords per
    // lane, re-read and re-written at every quadrature point, were what the kernel's time went into.
    // NB column vertices per pass over the quadrature points: all three (BALL) or one (a rolled loop of three passes)
    constexpr int NB = (BALL && STORE != 0) ? 3 : 1;
#pragma unroll 1
    for (int b = 0; b < (STORE == 0 || BALL ? 1 : 3); ++b) {
    // The columns a row writes are known at compile time (energy 0, electrons ie, potential, the
    // species of the sources) except its OWN column: that one has accumulators of its own, added to
    // Jacc[.][row] after the loops -- no run-time register selection inside them.
    double Racc[3] = {0.0, 0.0, 0.0}, Jacc[NB][3][NEQ], Jown[NB][3];   // [column vertex][row vertex a][column field s]
#pragma unroll
    for (int bb = 0; bb < NB; ++bb)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            Jown[bb][a] = 0.0;
#pragma unroll
            for (int s = 0; s < NEQ; ++s) Jacc[bb][a][s] = 0.0;
        }
    // the column vertex of accumulator set bb (compile-time bb; b is the rolled loop's vertex when NB = 1)
    auto col_vertex = [&](int bb) { return NB == 3 ? bb : b; };

    // value and gradient of unknown s at a point (s is wave-uniform or a compile-time index)
    auto value = [&](int s, const double phi[3]) {
        return ul[s * SLICE] * phi[0] + ul[(NEQ + s) * SLICE] * phi[1] + ul[(2 * NEQ + s) * SLICE] * phi[2];
    };
    auto grad_x = [&](int s) { return ul[s * SLICE] * c.G[0][0] + ul[(NEQ + s) * SLICE] * c.G[1][0] + ul[(2 * NEQ + s) * SLICE] * c.G[2][0]; };
    auto grad_y = [&](int s) { return ul[s * SLICE] * c.G[0][1] + ul[(NEQ + s) * SLICE] * c.G[1][1] + ul[(2 * NEQ + s) * SLICE] * c.G[2][1]; };
    const double gx0 = grad_x(0), gy0 = grad_y(0), gxe = grad_x(ie), gye = grad_y(ie);
    const double Ex = -grad_x(IPHI), Ey = -grad_y(IPHI);

    // exp(u_s) at volume quadrature point q from the table (q < 0: a facet point, evaluated)
    auto expu = [&](int s, int q, const double phi[3]) {
        return (exp_table && q >= 0) ? lds_e[((size_t)q * NEQ + s) * SLICE + lc] : exp(value(s, phi));
    };
    auto channels = [&](const double phi[3], int q) {
        const N3 meo = nodal3(fl, F_MEO, c, phi), ueo = nodal3(fl, F_UEO, c, phi);
        const double ieo = (exp_table && q >= 0) ? lds_e[((size_t)q * NEQ + ns) * SLICE + lc] : exp(-ueo.v);
        const double r0 = expu(0, q, phi) * ieo, re = expu(ie, q, phi) * ieo;
        GdChannels ch;
        ch.c1 = r0 - re * meo.v;
        ch.c2 = r0 * gx0 - re * (gxe * meo.v + meo.gx) - ch.c1 * ueo.gx;
        ch.c3 = r0 * gy0 - re * (gye * meo.v + meo.gy) - ch.c1 * ueo.gy;
        ch.r0 = r0;
        ch.rem = -re * meo.v;
        ch.c2_v0 = r0 * (gx0 - ueo.gx);
        ch.c3_v0 = r0 * (gy0 - ueo.gy);
        ch.c2_ve = -re * (gxe * meo.v + meo.gx - meo.v * ueo.gx);
        ch.c3_ve = -re * (gye * meo.v + meo.gy - meo.v * ueo.gy);
        return ch;
    };
    // W (tS phi_a - tX G_a,x - tY G_a,y) into column s of the accumulators (s: wave-uniform)
    // (source-only and flux-only forms: a product with a literal zero is not folded without fast-math)
    // s: a compile-time column where these are called (after inlining and unrolling), or OWN
    constexpr int OWN = -1;
    auto addS = [&](int bb, int s, double W, const double phi[3], double tS) {
        if (STORE == 0) return;
        const double w = W * tS;
        if (s == OWN) {
#pragma unroll
            for (int a = 0; a < 3; ++a) Jown[bb][a] += w * phi[a];
            return;
        }
#pragma unroll
        for (int k = 0; k < NEQ; ++k)
            if (k == s) {
#pragma unroll
                for (int a = 0; a < 3; ++a) Jacc[bb][a][k] += w * phi[a];
            }
    };
    auto addG = [&](int bb, int s, double W, double tX, double tY) {
        if (STORE == 0) return;
        const double wx = W * tX, wy = W * tY;
        if (s == OWN) {
#pragma unroll
            for (int a = 0; a < 3; ++a) Jown[bb][a] -= wx * c.G[a][0] + wy * c.G[a][1];
            return;
        }
#pragma unroll
        for (int k = 0; k < NEQ; ++k)
            if (k == s) {
#pragma unroll
                for (int a = 0; a < 3; ++a) Jacc[bb][a][k] -= wx * c.G[a][0] + wy * c.G[a][1];
            }
    };
    // derivatives of the channels along the basis function of column vertex b (value w_v, gradient w_x, w_y)
    struct Seeds {
        double w_v, w_x, w_y, k1_0, k2_0, k3_0, k1_e, k2_e, k3_e;
    };
    auto seeds = [&](int bb, const double phi[3], const GdChannels &ch) {
        const int bv = col_vertex(bb);
        Seeds sd;
        sd.w_v = bv == 0 ? phi[0] : bv == 1 ? phi[1] : phi[2];
        sd.w_x = bv == 0 ? c.G[0][0] : bv == 1 ? c.G[1][0] : c.G[2][0];
        sd.w_y = bv == 0 ? c.G[0][1] : bv == 1 ? c.G[1][1] : c.G[2][1];
        sd.k1_0 = ch.r0 * sd.w_v;
        sd.k2_0 = ch.c2_v0 * sd.w_v + ch.r0 * sd.w_x;
        sd.k3_0 = ch.c3_v0 * sd.w_v + ch.r0 * sd.w_y;
        sd.k1_e = ch.rem * sd.w_v;
        sd.k2_e = ch.c2_ve * sd.w_v + ch.rem * sd.w_x;
        sd.k3_e = ch.c3_ve * sd.w_v + ch.rem * sd.w_y;
        return sd;
    };
    auto species_flux = [&](int s, double scale, int own, const double phi[3], const GdChannels &ch, int q) {
        const int et = md->eq_type[s];
        const bool drift = et == FEDM_EQ_DRIFT_DIFFUSION_REACTION;
        const bool gdf = drift ? md->grad_diffusion[s] != 0 : true;   // diffusion-reaction: -grad(D exp(u))
        return gd_flux_partials(fl, F_MU + s, F_MUD + s, F_D + s, F_DD + s, c, phi, ch, md->sign[s], scale, drift, gdf,
                                value(own, phi), grad_x(own), grad_y(own), Ex, Ey,
                                (exp_table && q >= 0) ? lds_e[((size_t)q * NEQ + own) * SLICE + lc] : -1.0);
    };
    // directional derivative of a flux (column vertex b) into the flux columns of the accumulators
    auto flux_columns = [&](int bb, const GdFluxD &Fl, int own, const Seeds &sd, double W, const double phi[3]) {
        addG(bb, own, W, Fl.Gx * sd.w_v + Fl.dg * sd.w_x, Fl.Gy * sd.w_v + Fl.dg * sd.w_y);
        addG(bb, IPHI, W, -Fl.dE * sd.w_x, -Fl.dE * sd.w_y);   // E = -grad Phi
        addG(bb, 0, W, Fl.x_c1 * sd.k1_0 + Fl.x_c2 * sd.k2_0, Fl.y_c1 * sd.k1_0 + Fl.y_c3 * sd.k3_0);
        addG(bb, ie, W, Fl.x_c1 * sd.k1_e + Fl.x_c2 * sd.k2_e, Fl.y_c1 * sd.k1_e + Fl.y_c3 * sd.k3_e);
    };
    // ... of factor * (w . G) (Joule heating: w = E; wall flux: w = n) into the source columns
    auto flux_dot_columns = [&](int bb, const GdFluxD &Fl, int own, const Seeds &sd, double wx, double wy, double factor,
                                double W, const double phi[3]) {
        addS(bb, own, W, phi, factor * ((Fl.Gx * wx + Fl.Gy * wy) * sd.w_v + Fl.dg * (wx * sd.w_x + wy * sd.w_y)));
        addS(bb, IPHI, W, phi, -factor * Fl.dE * (wx * sd.w_x + wy * sd.w_y));
        const double o1 = factor * (Fl.x_c1 * wx + Fl.y_c1 * wy), o2 = factor * Fl.x_c2 * wx, o3 = factor * Fl.y_c3 * wy;
        addS(bb, 0, W, phi, o1 * sd.k1_0 + o2 * sd.k2_0 + o3 * sd.k3_0);
        addS(bb, ie, W, phi, o1 * sd.k1_e + o2 * sd.k2_e + o3 * sd.k3_e);
    };

    for (int q = 0; q < nqp; ++q) {
        // The staged nodal values do not depend on q or b: left alone, the compiler hoists their ~100
        // LDS reads out of both loops and spills them.  An opaque lane offset per point keeps every read
        // where it is used.
        {
            int lcq = lc;
            asm volatile("" : "+v"(lcq));
            fl = lds_f + lcq;
            ul = lds_u + lcq;
        }
        const double phi[3] = {1.0 - md->qp_x[q] - md->qp_y[q], md->qp_x[q], md->qp_y[q]};
        const double rq = c.rn[0] * phi[0] + c.rn[1] * phi[1] + c.rn[2] * phi[2];
        const double W = md->qp_w[q] * c.detJ * two_pi * rq;
        if (row == IPHI) {
            // 2 pi r (grad Phi . grad v - rho v): S = -rho, G = -grad Phi
            double rho = 0.0;
#pragma unroll
            for (int i = 1; i < ns; ++i) {
                const double ni = (md->sign[i] * md->charge_over_eps) * expu(i, q, phi);
                rho += ni;
#pragma unroll
                for (int bb = 0; bb < NB; ++bb) {
                    const int bv = col_vertex(bb);
                    addS(bb, i, W, phi, -ni * (bv == 0 ? phi[0] : bv == 1 ? phi[1] : phi[2]));
                }
            }
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) {
                const int bv = col_vertex(bb);
                addG(bb, IPHI, W, -(bv == 0 ? c.G[0][0] : bv == 1 ? c.G[1][0] : c.G[2][0]),
                     -(bv == 0 ? c.G[0][1] : bv == 1 ? c.G[1][1] : c.G[2][1]));
            }
            if (b == 0) {
#pragma unroll
                for (int a = 0; a < 3; ++a) Racc[a] += W * (-rho * phi[a] - (Ex * c.G[a][0] + Ey * c.G[a][1]));
            }
            continue;
        }
        const GdChannels ch = channels(phi, q);
        double n[ns];
#pragma unroll
        for (int i = 1; i < ns; ++i) n[i] = expu(i, q, phi);
        // source of this row: sum_j w_j rate_j with w_j = net[j][row] (species) or -loss_j (energy)
        double src = 0.0, src_c1 = 0.0, src_v[ns];
#pragma unroll
        for (int i = 0; i < ns; ++i) src_v[i] = 0.0;
        for (int j = 0; j < nr; ++j) {
            // (weight, packed powers) of reaction j for this row: one broadcast LDS read, returned with
            // the rate-coefficient reads that follow, instead of dependent scalar loads from the descriptor
            const int pk = __builtin_amdgcn_readfirstlane(lds_rp[row * FEDM_GD_MAX_REACTIONS + j]);
            if (pk < 0) continue;   // the reaction does not enter this row
            const double w = lds_rw[row * FEDM_GD_MAX_REACTIONS + j];
            const double kv = nodal1(fl, F_K + j, phi), kd = nodal1(fl, F_KD + j, phi);
            double prod = 1.0;
#pragma unroll
            for (int i = 0; i < ns; ++i) {
                // x^p for the small integer powers of a reaction scheme without a loop over p
                const int pw = (pk >> (4 * i)) & 15;
                const double x = (i == 0) ? md->N0 : n[i];
                if (pw <= 3) prod *= pw == 0 ? 1.0 : pw == 1 ? x : pw == 2 ? x * x : x * x * x;
                else
                    for (int e = 0; e < pw; ++e) prod *= x;
            }
            const double rate = (kv + kd * ch.c1) * prod;
            src += w * rate;
            src_c1 += w * kd * prod;
#pragma unroll
            for (int i = 1; i < ns; ++i) src_v[i] += w * (double)((pk >> (4 * i)) & 15) * rate;
        }
        // time term, fedm/functions.py:350-357
        const double hq = Hrow[0] * phi[0] + Hrow[1] * phi[1] + Hrow[2] * phi[2];
        const double vr = value(row, phi), er = expu(row, q, phi);
        const double T = er * (vr * c_new + hq) / dt;
        double S = T - src;
        const double dT = T + er * c_new / dt;
        double Gx = 0.0, Gy = 0.0;
        if (row == 0) {
            // energy: 5/3 of the electron coefficients on u_0 (fedm-gd.py:354), Joule heating -Gamma_e . E
            const GdFluxD Fw = species_flux(ie, 5.0 / 3.0, 0, phi, ch, q);
            Gx = Fw.Gx;
            Gy = Fw.Gy;
            const GdFluxD Fe = species_flux(ie, 1.0, ie, phi, ch, q);
            S += Fe.Gx * Ex + Fe.Gy * Ey;
            // Sentinel losses: weight m - Ei or -m with m = u_0 / u_e at the point, the expression the scripts pass as
            // mean_energy (fedm-gd.py:358).  sm_0, sm_e: d(-src)/du_0 and /du_e through m.
            double sm_0 = 0.0, sm_e = 0.0;
            if (md->mean_energy_form == FEDM_GD_ME_UNKNOWN_RATIO) {
                const double inv_ve = 1.0 / value(ie, phi), m = value(0, phi) * inv_ve;
                double src_m = 0.0;   // d src / d m
                for (int j = 0; j < nr; ++j) {
                    const int pk = __builtin_amdgcn_readfirstlane(lds_rp[ns * FEDM_GD_MAX_REACTIONS + j]);
                    if (pk < 0) continue;
                    const bool ei = ((pk >> 28) & 3) == 1;
                    const double w = ei ? m - md->energy_Ei : -m;
                    const double kv = nodal1(fl, F_K + j, phi), kd = nodal1(fl, F_KD + j, phi);
                    double prod = 1.0;
#pragma unroll
                    for (int i = 0; i < ns; ++i) {
                        const double x = (i == 0) ? md->N0 : n[i];
                        for (int e = 0; e < ((pk >> (4 * i)) & 15); ++e) prod *= x;
                    }
                    const double rate = (kv + kd * ch.c1) * prod;
                    S -= w * rate;
                    src_c1 += w * kd * prod;
                    src_m += ei ? rate : -rate;
#pragma unroll
                    for (int i = 1; i < ns; ++i) src_v[i] += w * (double)((pk >> (4 * i)) & 15) * rate;
                }
                sm_0 = -src_m * inv_ve;
                sm_e = src_m * m * inv_ve;
            }
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) {
                const Seeds sd = seeds(bb, phi, ch);
                addS(bb, 0, W, phi, (dT + sm_0) * sd.w_v - src_c1 * sd.k1_0);
#pragma unroll
                for (int i = 1; i < ns; ++i) addS(bb, i, W, phi, -src_v[i] * sd.w_v);
                addS(bb, ie, W, phi, sm_e * sd.w_v - src_c1 * sd.k1_e);
                flux_columns(bb, Fw, 0, sd, W, phi);
                flux_dot_columns(bb, Fe, ie, sd, Ex, Ey, 1.0, W, phi);
                addS(bb, IPHI, W, phi, -(Fe.Gx * sd.w_x + Fe.Gy * sd.w_y));   // d(G . E)/dE = G, E = -grad Phi
            }
        } else {
            const bool has_flux = md->eq_type[row] != FEDM_EQ_REACTION;
            GdFluxD Fl = {};
            if (has_flux) {
                Fl = species_flux(row, 1.0, row, phi, ch, q);
                Gx = Fl.Gx;
                Gy = Fl.Gy;
            }
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) {
                const Seeds sd = seeds(bb, phi, ch);
                addS(bb, OWN, W, phi, dT * sd.w_v);
#pragma unroll
                for (int i = 1; i < ns; ++i) addS(bb, i, W, phi, -src_v[i] * sd.w_v);
                addS(bb, 0, W, phi, -src_c1 * sd.k1_0);
                addS(bb, ie, W, phi, -src_c1 * sd.k1_e);
                if (has_flux) flux_columns(bb, Fl, OWN, sd, W, phi);
            }
        }
        if (b == 0) {
#pragma unroll
            for (int a = 0; a < 3; ++a) Racc[a] += W * (S * phi[a] - (Gx * c.G[a][0] + Gy * c.G[a][1]));
        }
    }

    // 'flux source' boundary facets, functions.py:514-522
    if (mode == 0 && row != IPHI) {
        for (int i = 0; i < 3; ++i) {
            const int tag = ftags[3 * cidx + i];
            if (tag <= 0) continue;
            const int sp = (row == 0) ? ie : row;
            const int et = md->eq_type[sp];
            if (et == FEDM_EQ_REACTION) continue;
            const int j = (i == 0) ? 1 : 0, kk = (i == 2) ? 1 : 2;   // the facet's two vertices
            const double Gi0 = i == 0 ? c.G[0][0] : i == 1 ? c.G[1][0] : c.G[2][0];
            const double Gi1 = i == 0 ? c.G[0][1] : i == 1 ? c.G[1][1] : c.G[2][1];
            const double gi = sqrt(Gi0 * Gi0 + Gi1 * Gi1);
            const double nx = -Gi0 / gi, ny = -Gi1 / gi;
            const int vj = j == 0 ? c.v[0] : c.v[1], vk = kk == 1 ? c.v[1] : c.v[2];
            const double ex = coords[2 * vj] - coords[2 * vk], ey = coords[2 * vj + 1] - coords[2 * vk + 1];
            const double L = sqrt(ex * ex + ey * ey);
            const double ref = md->ref[tag - 1][sp], fac = (1.0 - ref) / (1.0 + ref);
            for (int tq = 0; tq < md->n_fqp; ++tq) {
                const double tj = 1.0 - md->fqp_t[tq], tk = md->fqp_t[tq];
                // phi is tj at vertex j, tk at vertex kk, 0 at vertex i
                const double phi[3] = {i == 0 ? 0.0 : (j == 0 ? tj : tk), i == 1 ? 0.0 : (j == 1 ? tj : tk),
                                       i == 2 ? 0.0 : tk};
                const double rq = c.rn[0] * phi[0] + c.rn[1] * phi[1] + c.rn[2] * phi[2];
                const double W = md->fqp_w[tq] * L * two_pi * rq;
                const double dens = exp(value(row, phi));
                double vth = (sp == ie) ? sqrt(md->vth_e_coef * nodal1(fl, F_ME, phi)) : md->vth[sp];
                double mu_scale = 1.0, gam = md->gamma[tag - 1];
                if (row == 0) {  // energy: 5/3 mu, 1.3333 vth, gamma * mean energy of secondaries
                    vth *= 1.3333;
                    mu_scale = 5.0 / 3.0;
                    gam *= md->we_secondary;
                }
                double wall;
                if (et == FEDM_EQ_DIFFUSION_REACTION) {
                    wall = fac * (0.5 * vth * dens);
#pragma unroll
                    for (int bb = 0; bb < NB; ++bb) {
                        const int bv = col_vertex(bb);
                        addS(bb, OWN, W, phi, wall * (bv == 0 ? phi[0] : bv == 1 ? phi[1] : phi[2]));
                    }
                } else {
                    const GdChannels ch = channels(phi, -1);
                    const double En = Ex * nx + Ey * ny;
                    const double mub = nodal1(fl, F_MUD + sp, phi);
                    const double muv = nodal1(fl, F_MU + sp, phi) + mub * ch.c1;
                    const double zs = md->sign[sp] * mu_scale;
                    const double qd = zs * (muv * En), sg = qd < 0.0 ? -1.0 : 1.0;
                    wall = fac * ((0.5 * vth + sg * qd) * dens);
                    const double k = fac * dens * sg * zs;
#pragma unroll
                    for (int bb = 0; bb < NB; ++bb) {
                        const Seeds sd = seeds(bb, phi, ch);
                        addS(bb, OWN, W, phi, wall * sd.w_v);
                        addS(bb, 0, W, phi, k * mub * En * sd.k1_0);
                        addS(bb, ie, W, phi, k * mub * En * sd.k1_e);
                        addS(bb, IPHI, W, phi, -k * muv * (nx * sd.w_x + ny * sd.w_y));
                    }
                    if (sp == ie) {
                        // - 2 gamma / (1 + r) * sum over ions of Max(Gamma_s . n, 0), fedm-gd.py:351
                        const double cI = 2.0 * gam / (1.0 + ref);
                        for (int s = 1; s < ns; ++s) {
                            if (!md->is_ion[s]) continue;
                            const GdFluxD Fs = species_flux(s, 1.0, s, phi, ch, -1);
                            const double gn = Fs.Gx * nx + Fs.Gy * ny;
                            if (gn < 0.0) continue;
                            wall -= cI * gn;
#pragma unroll
                            for (int bb = 0; bb < NB; ++bb) flux_dot_columns(bb, Fs, s, seeds(bb, phi, ch), nx, ny, -cI, W, phi);
                        }
                    }
                }
                if (b == 0) {
#pragma unroll
                    for (int a = 0; a < 3; ++a) Racc[a] += W * wall * phi[a];
                }
            }
        }
    }

#pragma unroll
    for (int bb = 0; bb < NB; ++bb)
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int k = 0; k < NEQ; ++k) Jacc[bb][a][k] += (k == row) ? Jown[bb][a] : 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (b == 0) {
            if (elemF) elemF[(size_t)(a * NEQ + row) * n_cells + cidx] = Racc[a];   // summed by gd_gather_residual_kernel
            else unsafeAtomicAdd(&F[(size_t)c.v[a] * NEQ + row], Racc[a]);
        }
        if (STORE == 0) continue;
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) {
        const int bv = col_vertex(bb);
        if (STORE == 3) {
            // element buffer in the order of the DESTINATIONS: [row * NEQ + s][k], k = this (cell, a, b)'s place in the
            // list of contributions sorted by stored matrix position (`cell_slots` carries that table here): the gather
            // kernel then reads consecutive addresses for consecutive matrix rows, and these stores, scattered over
            // the few hundred positions around a workgroup's 64 cells, meet in the L2
            const size_t n_k = (size_t)9 * n_cells;
            double *dst = val + (size_t)(row * NEQ) * n_k + cell_slots[(size_t)cidx * 9 + a * 3 + bv];
#pragma unroll
            for (int s = 0; s < NEQ; ++s) dst[(size_t)s * n_k] = Jacc[bb][a][s];
            continue;
        }
        if (STORE == 2) {   // element buffer [(a * 3 + b) * NEQ2 + row * NEQ + s][cell]: lanes = cells, coalesced
            double *dst = val + ((size_t)(a * 3 + bv) * NEQ2 + row * NEQ) * n_cells + cidx;
#pragma unroll
            for (int s = 0; s < NEQ; ++s) dst[(size_t)s * n_cells] = Jacc[bb][a][s];
            continue;
        }
        const uint32_t slot = cell_slots[(size_t)cidx * 9 + a * 3 + bv];
        double *dst = val + ((size_t)(slot >> 6) * NEQ2) * SLICE + (slot & 63);
#pragma unroll
        for (int s = 0; s < NEQ; ++s) unsafeAtomicAdd(&dst[(size_t)(row * NEQ + s) * SLICE], Jacc[bb][a][s]);
        }
    }
    }   // column vertex b
    GD_ROW_T(row)
    }   // pass
}

// every stored block position p = (block column, lane) of the sliced block-ELL matrix sums the
// element blocks that land on it (inv_ptr / inv_idx: inverse of cell_slots), in the order of the
// list, and writes its NEQ * NEQ planes: one coalesced store per matrix value, padding included
template <int NEQ>
__global__ __launch_bounds__(256) void gd_gather_kernel(int n_pos, const int *__restrict__ inv_ptr,
                                                        const int *__restrict__ inv_idx,
                                                        const double *__restrict__ elem, double *__restrict__ val,
                                                        int row_first, int row_last, int n_cells) {
    constexpr int NEQ2 = NEQ * NEQ;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pos) return;
    double acc[NEQ2];
#pragma unroll
    for (int e = 0; e < NEQ2; ++e) acc[e] = 0.0;
    for (int k = inv_ptr[p]; k < inv_ptr[p + 1]; ++k) {
        const int ce = inv_idx[k], cell = ce / 9, ab = ce - 9 * cell;
        const double *src = elem + (size_t)ab * NEQ2 * n_cells + cell;
#pragma unroll
        for (int e = 0; e < NEQ2; ++e) acc[e] += src[(size_t)e * n_cells];
    }
    double *dst = val + ((size_t)(p >> 6) * NEQ2) * SLICE + (p & 63);
#pragma unroll
    for (int e = 0; e < NEQ2; ++e)
        if (e / NEQ >= row_first && e / NEQ <= row_last) dst[(size_t)e * SLICE] = acc[e];
}

// the same for the element buffer in destination order (STORE 3): position p's contributions are the entries
// [inv_ptr[p], inv_ptr[p + 1]) of every plane -- consecutive positions read consecutive addresses
template <int NEQ>
__global__ __launch_bounds__(256) void gd_gather_dest_kernel(int n_pos, const int *__restrict__ inv_ptr,
                                                             const double *__restrict__ elem, double *__restrict__ val,
                                                             int row_first, int row_last, int n_cells) {
    constexpr int NEQ2 = NEQ * NEQ;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pos) return;
    const size_t n_k = (size_t)9 * n_cells;
    double acc[NEQ2];
#pragma unroll
    for (int e = 0; e < NEQ2; ++e) acc[e] = 0.0;
    for (int k = inv_ptr[p]; k < inv_ptr[p + 1]; ++k) {
#pragma unroll
        for (int e = 0; e < NEQ2; ++e) acc[e] += elem[(size_t)e * n_k + k];
    }
    double *dst = val + ((size_t)(p >> 6) * NEQ2) * SLICE + (p & 63);
#pragma unroll
    for (int e = 0; e < NEQ2; ++e)
        if (e / NEQ >= row_first && e / NEQ <= row_last) dst[(size_t)e * SLICE] = acc[e];
}

// The same gather with a thread per (stored position, equation row): NEQ accumulators and NEQ loads per
// contribution and thread instead of NEQ^2 -- five times the threads in flight for the same loads, so a wave whose
// diagonal lanes walk 6-8 contributions holds the others up for a fifth of the work (FEDM_GD_GATHER=rows).
template <int NEQ>
__global__ __launch_bounds__(256) void gd_gather_dest_rows_kernel(int n_pos, const int *__restrict__ inv_ptr,
                                                                  const double *__restrict__ elem, double *__restrict__ val,
                                                                  int row_first, int row_last, int n_cells) {
    constexpr int NEQ2 = NEQ * NEQ;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = row_first + (int)blockIdx.y;
    if (p >= n_pos || r > row_last) return;
    const size_t n_k = (size_t)9 * n_cells;
    double acc[NEQ];
#pragma unroll
    for (int s = 0; s < NEQ; ++s) acc[s] = 0.0;
    const double *src = elem + (size_t)(r * NEQ) * n_k;
    for (int k = inv_ptr[p]; k < inv_ptr[p + 1]; ++k) {
#pragma unroll
        for (int s = 0; s < NEQ; ++s) acc[s] += src[(size_t)s * n_k + k];
    }
    double *dst = val + ((size_t)(p >> 6) * NEQ2 + r * NEQ) * SLICE + (p & 63);
#pragma unroll
    for (int s = 0; s < NEQ; ++s) dst[(size_t)s * SLICE] = acc[s];
}

// every vertex sums the element residuals of its cells (inverse of the connectivity), fixed order
template <int NEQ>
__global__ __launch_bounds__(256) void gd_gather_residual_kernel(int nv, const int *__restrict__ inv_ptr,
                                                                 const int *__restrict__ inv_idx,
                                                                 const double *__restrict__ elemF,
                                                                 double *__restrict__ F, int row_first, int n_cells) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nv) return;
    double acc[NEQ];
#pragma unroll
    for (int r = 0; r < NEQ; ++r) acc[r] = 0.0;
    for (int k = inv_ptr[v]; k < inv_ptr[v + 1]; ++k) {
        const int ca = inv_idx[k], cell = ca / 3, a = ca - 3 * cell;
#pragma unroll
        for (int r = 0; r < NEQ; ++r)
            if (r >= row_first) acc[r] += elemF[(size_t)(a * NEQ + r) * n_cells + cell];
    }
#pragma unroll
    for (int r = 0; r < NEQ; ++r)
        if (r >= row_first) F[(size_t)v * NEQ + r] = acc[r];
}

// element buffers and the inverse maps (cell_slots, connectivity), built the first time they are needed
static int gd_elem_setup(Ctx &c) {
    if (c.d_gd_elem) return 0;
    const size_t n_pos = (size_t)c.pat.total_bc * SLICE, n_e = (size_t)c.nc * 9;
    std::vector<int> ptr(n_pos + 1, 0), idx(n_e);
    for (size_t e = 0; e < n_e; ++e) ++ptr[c.pat.cell_slots[e] + 1];
    for (size_t p = 0; p < n_pos; ++p) ptr[p + 1] += ptr[p];
    std::vector<int> fill(ptr.begin(), ptr.end() - 1);
    for (size_t e = 0; e < n_e; ++e) idx[fill[c.pat.cell_slots[e]]++] = (int)e;
    std::vector<uint32_t> kpos(n_e);   // (cell, a, b) -> its place in that list (the destination-ordered buffer)
    for (size_t k = 0; k < n_e; ++k) kpos[idx[k]] = (uint32_t)k;

    // vertex -> (cell, local vertex): read back from the device copy of the connectivity
    std::vector<int> cells_h((size_t)c.nc * 3), vptr((size_t)c.nv + 1, 0), vidx((size_t)c.nc * 3);
    if (hipMemcpy(cells_h.data(), c.d_cells, sizeof(int) * cells_h.size(), hipMemcpyDeviceToHost) != hipSuccess) {
        hipGetLastError();
        return -1;
    }
    for (size_t e = 0; e < cells_h.size(); ++e) ++vptr[cells_h[e] + 1];
    for (int v = 0; v < c.nv; ++v) vptr[v + 1] += vptr[v];
    {
        std::vector<int> vfill(vptr.begin(), vptr.end() - 1);
        for (size_t e = 0; e < cells_h.size(); ++e) vidx[vfill[cells_h[e]]++] = (int)e;
    }
    if (hipMalloc((void **)&c.d_gd_vinv_ptr, sizeof(int) * vptr.size()) != hipSuccess ||
        hipMalloc((void **)&c.d_gd_vinv_idx, sizeof(int) * vidx.size()) != hipSuccess ||
        hipMalloc((void **)&c.d_gd_elemF, sizeof(double) * (size_t)c.nc * 3 * c.neq) != hipSuccess ||
        hipMemcpy(c.d_gd_vinv_ptr, vptr.data(), sizeof(int) * vptr.size(), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(c.d_gd_vinv_idx, vidx.data(), sizeof(int) * vidx.size(), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemset(c.d_gd_elemF, 0, sizeof(double) * (size_t)c.nc * 3 * c.neq) != hipSuccess ||
        hipMalloc((void **)&c.d_gd_kpos, sizeof(uint32_t) * kpos.size()) != hipSuccess ||
        hipMemcpy(c.d_gd_kpos, kpos.data(), sizeof(uint32_t) * kpos.size(), hipMemcpyHostToDevice) != hipSuccess ||
        hipMalloc((void **)&c.d_gd_inv_ptr, sizeof(int) * ptr.size()) != hipSuccess ||
        hipMalloc((void **)&c.d_gd_inv_idx, sizeof(int) * idx.size()) != hipSuccess ||
        hipMalloc((void **)&c.d_gd_elem, sizeof(double) * n_e * c.neq * c.neq) != hipSuccess ||
        hipMemcpy(c.d_gd_inv_ptr, ptr.data(), sizeof(int) * ptr.size(), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(c.d_gd_inv_idx, idx.data(), sizeof(int) * idx.size(), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemset(c.d_gd_elem, 0, sizeof(double) * n_e * c.neq * c.neq) != hipSuccess) {
        hipGetLastError();
        for (void *q : {(void *)c.d_gd_inv_ptr, (void *)c.d_gd_inv_idx, (void *)c.d_gd_elem, (void *)c.d_gd_vinv_ptr,
                        (void *)c.d_gd_vinv_idx, (void *)c.d_gd_elemF, (void *)c.d_gd_kpos})
            if (q) hipFree(q);
        c.d_gd_kpos = nullptr;
        c.d_gd_inv_ptr = c.d_gd_inv_idx = c.d_gd_vinv_ptr = c.d_gd_vinv_idx = nullptr;
        c.d_gd_elem = c.d_gd_elemF = nullptr;
        return -1;   // the caller falls back to the atomics
    }
    return 0;
}

void launch_assemble_gd(Ctx &c, bool jacobian, int mode) {
    hipMemsetAsync(c.d_F, 0, sizeof(double) * c.np, c.stream);
    const int ncol = (int)c.pat.colour_ptr.size() - 1;
    // Ctx::gd_hand_mode (FEDM_GD_HAND when the context is created): 0 = dual numbers, a launch per
    // colour (the cross-check of the hand-derived blocks), 2 = hand-derived blocks, fp64 atomics into the
    // matrix, 3 (default) = hand-derived blocks, element buffer in destination order + gather, 4 = the buffer in cell order
    const int hand_mode = c.gd_hand_mode;
    if (jacobian && hand_mode >= 2) {
        // in Poisson-only mode the other rows keep the zeros of the memset (identity rows follow)
        const int n = c.nc;
        const bool gather = hand_mode >= 3 && gd_elem_setup(c) == 0;
        const bool dest_order = hand_mode == 3 || hand_mode == 5;   // (4: the buffer in cell order, round 2's layout)
        const bool ball = hand_mode == 5;
        // the gather's thread mapping: a thread per (position, equation row) by default (measured at 200 k DOFs, F + J:
        // 353 us with a thread per position, 287-296 us per (position, row), the same per (position, plane));
        // FEDM_GD_GATHER=positions|rows
        const char *gd_waves_env = std::getenv("FEDM_GD_WAVES");     // (per launch: the tests switch it)
        const char gd_waves_mode = gd_waves_env ? gd_waves_env[0] : 'a';
        static const char gather_kind = [] {
            const char *e = std::getenv("FEDM_GD_GATHER");
            return e ? e[0] : 'r';
        }();
        const bool gather_rows = gather_kind == 'r';
        if (!gather || mode != 0)   // (the gather writes every value of the rows it covers)
            hipMemsetAsync(c.d_val, 0, sizeof(double) * (size_t)c.pat.total_bc * SLICE * c.neq * c.neq, c.stream);
        const int cpb = SLICE, nf = c.gd_n_fields;
        size_t lds_h = sizeof(double) * ((size_t)(3 * cpb + 1) / 2 + (size_t)cpb * nf * 3 + (size_t)cpb * 3 * c.neq);
        // table of exp(u) at the quadrature points, while two workgroups still fit a CU's 160 KB
        const size_t lds_table = sizeof(double) * (size_t)c.gd.n_qp * c.neq * cpb;
        const int exp_table = lds_h + lds_table <= 80 * 1024 ? 1 : 0;
        if (exp_table) lds_h += lds_table;
        lds_h += (size_t)(c.neq - 1) * FEDM_GD_MAX_REACTIONS * 12 + FEDM_GD_MAX_REACTIONS * 4 + 8;   // the rows' reaction weights and powers, the sentinel list
        const dim3 gh((unsigned)((n + cpb - 1) / cpb), 1), bh(SLICE * (c.neq - 1));
        const int n_pos = (int)(c.pat.total_bc * SLICE);
        const int row_first = mode == 1 ? c.neq - 1 : 0, row_last = c.neq - 1;
#define FEDM_GD_HAND_LAUNCH(NEQ, NRC, NQC)                                                                               \
    do {                                                                                                          \
        static bool lds_attr_set = false;   /* per instantiation: a property of these kernels */            \
        if (lds_h > 64 * 1024 && !lds_attr_set) {   /* more dynamic LDS than the default limit */               \
            hipFuncSetAttribute(reinterpret_cast<const void *>(&gd_jacobian_rows_kernel<NEQ, 3, NRC, NQC>),                 \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_h);                          \
            hipFuncSetAttribute(reinterpret_cast<const void *>(&gd_jacobian_rows_kernel<NEQ, 2, NRC, NQC>),                 \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_h);                          \
            hipFuncSetAttribute(reinterpret_cast<const void *>(&gd_jacobian_rows_kernel<NEQ, 1, NRC, NQC>),                 \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_h);                          \
            hipFuncSetAttribute(reinterpret_cast<const void *>(&gd_jacobian_rows_kernel<NEQ, 0, NRC, NQC>),                 \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_h);                          \
            lds_attr_set = true;                                                                                  \
        }                                                                                                         \
        if (gather && dest_order) {                                                                               \
            if (ball) {                                                                                           \
                /* two waves a workgroup, each taking two rows in turn, two workgroups a CU: the set-up is paid once for two   \
                 * rows and the rows balance -- 509 against 558 us at 402 k DOFs (2 500 workgroups), 268 against 260 us at      \
                 * 200 k (1 243: the rounds of workgroups quantise the same way): from six workgroups a CU on                  \
                 * (FEDM_GD_WAVES=rows / two: a wave per row / two waves, whatever the size) */                                 \
                constexpr int WV2 = (NEQ - 1) % 2 == 0 && NEQ > 3 ? 2 : 0;                                        \
                if (WV2 && (gd_waves_mode == 't' || (gd_waves_mode != 'r' && (long long)gh.x >= 6LL * 256))) {   \
                    static bool ball2_attr_set = false;                                                           \
                    if (lds_h > 64 * 1024 && !ball2_attr_set) {                                                   \
                        hipFuncSetAttribute(reinterpret_cast<const void *>(&gd_jacobian_rows_kernel<NEQ, 3, NRC, NQC, true, WV2>), \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_h);              \
                        ball2_attr_set = true;                                                                    \
                    }                                                                                             \
                    hipLaunchKernelGGL((gd_jacobian_rows_kernel<NEQ, 3, NRC, NQC, true, WV2>), gh, dim3(SLICE * (WV2 ? WV2 : 1)), lds_h, c.stream, c.d_gd, c.d_gd_fields, \
                                       c.nv, (const int *)nullptr, n, c.d_cells, c.d_coords, c.d_ftags, c.d_gd_kpos, \
                                       c.d_u, c.d_uold, c.d_uold1, c.dt, c.dt_old, c.d_gd_elem, c.d_F, mode, c.d_gd_elemF, exp_table); \
                } else {                                                                                          \
                static bool ball_attr_set = false;                                                                \
                if (lds_h > 64 * 1024 && !ball_attr_set) {                                                        \
                    hipFuncSetAttribute(reinterpret_cast<const void *>(&gd_jacobian_rows_kernel<NEQ, 3, NRC, NQC, true>), \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_h);                  \
                    ball_attr_set = true;                                                                         \
                }                                                                                                 \
                hipLaunchKernelGGL((gd_jacobian_rows_kernel<NEQ, 3, NRC, NQC, true>), gh, bh, lds_h, c.stream, c.d_gd, c.d_gd_fields, \
                                   c.nv, (const int *)nullptr, n, c.d_cells, c.d_coords, c.d_ftags, c.d_gd_kpos,  \
                                   c.d_u, c.d_uold, c.d_uold1, c.dt, c.dt_old, c.d_gd_elem, c.d_F, mode, c.d_gd_elemF, exp_table); \
                }                                                                                                 \
            } else                                                                                                \
            hipLaunchKernelGGL((gd_jacobian_rows_kernel<NEQ, 3, NRC, NQC>), gh, bh, lds_h, c.stream, c.d_gd, c.d_gd_fields, \
                               c.nv, (const int *)nullptr, n, c.d_cells, c.d_coords, c.d_ftags, c.d_gd_kpos,      \
                               c.d_u, c.d_uold, c.d_uold1, c.dt, c.dt_old, c.d_gd_elem, c.d_F, mode, c.d_gd_elemF, exp_table); \
            if (gather_rows)                                                                                      \
                hipLaunchKernelGGL((gd_gather_dest_rows_kernel<NEQ>), dim3((n_pos + 255) / 256, row_last - row_first + 1), \
                                   dim3(256), 0, c.stream, n_pos, c.d_gd_inv_ptr, c.d_gd_elem, c.d_val, row_first, row_last, n); \
            else                                                                                                  \
            hipLaunchKernelGGL((gd_gather_dest_kernel<NEQ>), dim3((n_pos + 255) / 256), dim3(256), 0, c.stream, n_pos, \
                               c.d_gd_inv_ptr, c.d_gd_elem, c.d_val, row_first, row_last, n);                     \
            hipLaunchKernelGGL((gd_gather_residual_kernel<NEQ>), dim3((c.nv + 255) / 256), dim3(256), 0, c.stream, \
                               c.nv, c.d_gd_vinv_ptr, c.d_gd_vinv_idx, c.d_gd_elemF, c.d_F, row_first, n);        \
        } else if (gather) {                                                                                      \
            hipLaunchKernelGGL((gd_jacobian_rows_kernel<NEQ, 2, NRC, NQC>), gh, bh, lds_h, c.stream, c.d_gd, c.d_gd_fields, \
                               c.nv, (const int *)nullptr, n, c.d_cells, c.d_coords, c.d_ftags, c.d_cell_slots,   \
                               c.d_u, c.d_uold, c.d_uold1, c.dt, c.dt_old, c.d_gd_elem, c.d_F, mode, c.d_gd_elemF, exp_table); \
            hipLaunchKernelGGL((gd_gather_kernel<NEQ>), dim3((n_pos + 255) / 256), dim3(256), 0, c.stream, n_pos, \
                               c.d_gd_inv_ptr, c.d_gd_inv_idx, c.d_gd_elem, c.d_val, row_first, row_last, n);     \
            hipLaunchKernelGGL((gd_gather_residual_kernel<NEQ>), dim3((c.nv + 255) / 256), dim3(256), 0, c.stream, \
                               c.nv, c.d_gd_vinv_ptr, c.d_gd_vinv_idx, c.d_gd_elemF, c.d_F, row_first, n);        \
        } else {                                                                                                  \
            hipLaunchKernelGGL((gd_jacobian_rows_kernel<NEQ, 1, NRC, NQC>), gh, bh, lds_h, c.stream, c.d_gd, c.d_gd_fields, \
                               c.nv, (const int *)nullptr, n, c.d_cells, c.d_coords, c.d_ftags, c.d_cell_slots,   \
                               c.d_u, c.d_uold, c.d_uold1, c.dt, c.dt_old, c.d_val, c.d_F, mode, (double *)nullptr, exp_table); \
        }                                                                                                         \
    } while (0)
        switch (c.neq) {
            case 3: FEDM_GD_HAND_LAUNCH(3, 0, 0); break;
            case 4: FEDM_GD_HAND_LAUNCH(4, 0, 0); break;
            case 5: FEDM_GD_HAND_LAUNCH(5, 0, 0); break;
            case 6: FEDM_GD_HAND_LAUNCH(6, 0, 0); break;
        }
#undef FEDM_GD_HAND_LAUNCH
        return;
    }
    if (!jacobian && hand_mode >= 2) {
        // residual only: the same point functions without the derivative columns, all cells in one
        // launch; element residuals summed per vertex (no atomics: fixed order)
        const int n = c.nc;
        double *elemF = ((hand_mode == 3 || hand_mode == 5) && gd_elem_setup(c) == 0) ? c.d_gd_elemF : nullptr;
        size_t lds_h = sizeof(double) * ((size_t)(3 * SLICE + 1) / 2 + (size_t)SLICE * c.gd_n_fields * 3 +
                                         (size_t)SLICE * 3 * c.neq);
        const size_t lds_table = sizeof(double) * (size_t)c.gd.n_qp * c.neq * SLICE;
        const int exp_table = lds_h + lds_table <= 80 * 1024 ? 1 : 0;
        if (exp_table) lds_h += lds_table;
        lds_h += (size_t)(c.neq - 1) * FEDM_GD_MAX_REACTIONS * 12 + FEDM_GD_MAX_REACTIONS * 4 + 8;
        const dim3 gh((unsigned)((n + SLICE - 1) / SLICE), 1), bh(SLICE * (c.neq - 1));
        const int row_first = mode == 1 ? c.neq - 1 : 0;
#define FEDM_GD_RES_LAUNCH(NEQ, NRC, NQC)                                                                                \
    do {                                                                                                          \
        static bool lds_attr_set = false;                                                                         \
        if (lds_h > 64 * 1024 && !lds_attr_set) {                                                                 \
            hipFuncSetAttribute(reinterpret_cast<const void *>(&gd_jacobian_rows_kernel<NEQ, 2, NRC, NQC>),                 \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_h);                          \
            hipFuncSetAttribute(reinterpret_cast<const void *>(&gd_jacobian_rows_kernel<NEQ, 1, NRC, NQC>),                 \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_h);                          \
            hipFuncSetAttribute(reinterpret_cast<const void *>(&gd_jacobian_rows_kernel<NEQ, 0, NRC, NQC>),                 \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_h);                          \
            lds_attr_set = true;                                                                                  \
        }                                                                                                         \
        hipLaunchKernelGGL((gd_jacobian_rows_kernel<NEQ, 0, NRC, NQC>), gh, bh, lds_h, c.stream, c.d_gd, c.d_gd_fields,     \
                           c.nv, (const int *)nullptr, n, c.d_cells, c.d_coords, c.d_ftags, c.d_cell_slots,       \
                           c.d_u, c.d_uold, c.d_uold1, c.dt, c.dt_old, (double *)nullptr, c.d_F, mode, elemF, exp_table); \
        if (elemF)                                                                                                \
            hipLaunchKernelGGL((gd_gather_residual_kernel<NEQ>), dim3((c.nv + 255) / 256), dim3(256), 0, c.stream, \
                               c.nv, c.d_gd_vinv_ptr, c.d_gd_vinv_idx, elemF, c.d_F, row_first, n);               \
    } while (0)
        switch (c.neq) {
            case 3: FEDM_GD_RES_LAUNCH(3, 0, 0); break;
            case 4: FEDM_GD_RES_LAUNCH(4, 0, 0); break;
            case 5: FEDM_GD_RES_LAUNCH(5, 0, 0); break;
            case 6: FEDM_GD_RES_LAUNCH(6, 0, 0); break;
        }
#undef FEDM_GD_RES_LAUNCH
        return;
    }
    if (jacobian)
        hipMemsetAsync(c.d_val, 0, sizeof(double) * (size_t)c.pat.total_bc * SLICE * c.neq * c.neq, c.stream);
    for (int k = 0; k < ncol; ++k) {
        const int n = c.pat.colour_ptr[k + 1] - c.pat.colour_ptr[k];
        if (n == 0) continue;
        const size_t threads = jacobian ? (size_t)n * 3 * c.neq : (size_t)n;
        const dim3 g((unsigned)((threads + 127) / 128)), b(128);
#define FEDM_GD_LAUNCH(NEQ)                                                                        \
    hipLaunchKernelGGL((gd_assemble_kernel<NEQ>), g, b, 0, c.stream, c.d_gd, c.d_gd_fields, c.nv,   \
                       c.d_colour_cells + c.pat.colour_ptr[k], n, c.d_cells, c.d_coords, c.d_ftags, \
                       c.d_cell_slots, c.d_u, c.d_uold, c.d_uold1, c.dt, c.dt_old, c.d_val, c.d_F,  \
                       jacobian ? 1 : 0, mode)
        switch (c.neq) {
            case 3: FEDM_GD_LAUNCH(3); break;
            case 4: FEDM_GD_LAUNCH(4); break;
            case 5: FEDM_GD_LAUNCH(5); break;
            case 6: FEDM_GD_LAUNCH(6); break;
        }
#undef FEDM_GD_LAUNCH
    }
}

}  // namespace fedm
