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

namespace fedm {

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
        for (int j = 0; j < nr; ++j) {
            Dual rate = gd_rate_coefficient(fields, nv, ns, nr, j, c, phi, dme);
#pragma unroll
            for (int i = 0; i < ns; ++i)
                for (int e = 0; e < md->power[j][i]; ++e) rate = (i == 0) ? rate * md->N0 : rate * p.n[i];
#pragma unroll
            for (int i = 0; i < ns; ++i) f[i] = f[i] + (double)md->net[j][i] * rate;
            f_en = f_en - md->energy_loss[j] * rate;
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

void launch_assemble_gd(Ctx &c, bool jacobian, int mode) {
    hipMemsetAsync(c.d_F, 0, sizeof(double) * c.np, c.stream);
    if (jacobian)
        hipMemsetAsync(c.d_val, 0, sizeof(double) * (size_t)c.pat.total_bc * SLICE * c.neq * c.neq, c.stream);
    const int ncol = (int)c.pat.colour_ptr.size() - 1;
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
