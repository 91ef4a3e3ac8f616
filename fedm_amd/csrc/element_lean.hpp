// Lean cell evaluation for the patch assembly (F + J, models with a Poisson row, FIAT's 3-point
// triangle rule, no Expression sources): the same element tensors as element.hpp, arranged so
// that the live state is bounded by ONE equation row.  The species rows run in a loop that is
// not unrolled: the row's nodal values are re-read from the LDS staging area by run-time index,
// its transport coefficients are evaluated inside the row, and only the geometry, the field, the
// rate coefficients and exp(u) at the quadrature points are shared between rows.  That is what
// lets three workgroups share a CU instead of two (<= 168 VGPRs); see DESIGN.md 8.1.
#pragma once
#include "element.hpp"

namespace fedm {

template <int NS>
__device__ __forceinline__ double pick(const double (&v)[NS], int s) {
    double r = v[0];
#pragma unroll
    for (int i = 1; i < NS; ++i) r = (s == i) ? v[i] : r;
    return r;
}

// The packed indices of a cell (local vertex ids, block columns) are all that a thread keeps in
// registers between equation rows.
struct LeanCell {
    int wl, wj0, wj1, wj2;
};

// =============================================================================================
// Second generation of the row-at-a-time routine (assemble_lean2_kernel).  Differences:
//  * exp(u) at the quadrature points of FIAT's 3-point rule is a product of per-VERTEX factors:
//    u(q) = (u0+u1+u2)/6 + u_q/2, so with a_v = exp(u_v/6) staged once per patch vertex
//    n(q) = a0 a1 a2 * a_q^3 -- about 1.6 exponentials per thread (210 per patch) instead of six
//    per cell, and no per-thread LDS column for them;
//  * the field (E, 1/|E|, ln|E|) and -- for single-reaction models -- the rate coefficient and its
//    derivative are evaluated once per cell and kept in a per-thread LDS column between the rows
//    (sqrt, division, log and the rate's exponentials were re-evaluated in every row);
//  * JAC = false is the residual-only assembly on the same code (no accumulators, no row phases).
// =============================================================================================
template <int NR>
struct LeanStash {
    static constexpr bool K = NR == 1;                 // rate coefficients of one reaction fit the column
    static constexpr int N = 5 + (K ? 2 * NR : 0);     // doubles per thread: E (2), 1/|E|, ln|E|, 1/det [, k, k']
    static constexpr int IDET = 4, KV = 5;
};

template <int NS, int NR>
__device__ __forceinline__ LeanCell lean2_prologue(const fedm_model_desc *__restrict__ md, const PatchCell &pc,
                                                   const double *__restrict__ vx, const double *__restrict__ Ul,
                                                   double *__restrict__ cst, int stride) {
    constexpr int NEQ = NS + 1, IPHI = NS;
    LeanCell lc;
    lc.wl = pc.lv[0] | (pc.lv[1] << 8) | (pc.lv[2] << 16);
    lc.wj0 = pc.j[0] | (pc.j[1] << 8) | (pc.j[2] << 16) | (pc.j[3] << 24);
    lc.wj1 = pc.j[4] | (pc.j[5] << 8) | (pc.j[6] << 16) | (pc.j[7] << 24);
    lc.wj2 = pc.j[8];
    double x[3][2];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        x[a][0] = vx[2 * pc.lv[a]];
        x[a][1] = vx[2 * pc.lv[a] + 1];
    }
    CellGeom cg;
    cg.init(x, md->axisymmetric);
    double E[2] = {0.0, 0.0};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double p = Ul[pc.lv[a] * NEQ + IPHI];
        E[0] -= p * cg.G[a][0];
        E[1] -= p * cg.G[a][1];
    }
    // 1/|E| by reciprocal square root, |E| = E^2 / |E|, ln|E| = ln(E^2)/2: one transcendental chain
    // (rsqrt) instead of two (sqrt, division)
    const double E2 = E[0] * E[0] + E[1] * E[1];
    const double invEm = rsqrt(E2);
    const double Em = E2 * invEm;
    const double lnE = 0.5 * log(E2);
    cst[0 * stride] = E[0];
    cst[1 * stride] = E[1];
    cst[2 * stride] = invEm;
    cst[3 * stride] = lnE;
    cst[LeanStash<NR>::IDET * stride] = cg.inv_det;
    if constexpr (LeanStash<NR>::K) {
        double kv = 0.0, kd = 0.0;
        if (md->n_reactions > 0) termsum_eval(md->k[0], Em, invEm, lnE, kv, kd);
        cst[LeanStash<NR>::KV * stride] = kv;
        cst[(LeanStash<NR>::KV + 1) * stride] = kd;
    }
    return lc;
}

// geometry of the cell: gradients of the P1 basis and the three quadrature weights times 2 pi r
__device__ __forceinline__ void lean2_geometry(const fedm_model_desc *__restrict__ md, const LeanCell &lc,
                                               const double *__restrict__ vx, int (&lv)[3], double (&G)[3][2],
                                               double (&W)[3], double idet_known = 0.0) {
    const double two_pi = 6.283185307179586476925286766559;
#pragma unroll
    for (int a = 0; a < 3; ++a) lv[a] = (lc.wl >> (8 * a)) & 255;
    double x[3][2];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        x[a][0] = vx[2 * lv[a]];
        x[a][1] = vx[2 * lv[a] + 1];
    }
    CellGeom cg;
    cg.init(x, md->axisymmetric, idet_known);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        G[a][0] = cg.G[a][0];
        G[a][1] = cg.G[a][1];
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        double rq = 0.0;
#pragma unroll
        for (int a = 0; a < 3; ++a) rq += cg.rn[a] * (a == q ? 2.0 / 3.0 : 1.0 / 6.0);
        W[q] = (1.0 / 6.0) * cg.detJ * two_pi * rq;
    }
}

// cmask: bit (row * NEQ + col) marks a plane of the Jacobian that never changes -- the
// potential-potential plane 2 pi r grad(phi_a).grad(phi_b) (mesh and weights only) and the species
// planes that are structurally zero (no reaction couples the two species: d(electron row)/d(ion
// density) of the streamer model).  Once a full assembly has written them, later assemblies neither
// accumulate nor stream them out again (18 of the streamer's 90 LDS atomics per cell and two
// ninths of its matrix bytes).  The tests are wave-uniform.
// ROW >= 0: the equation row is known at compile time (the selects on the row index fold away, the
// model scalars of the row become constant-offset loads); ROW = -1: run-time row.
template <int NS, int NR, bool JAC, int ROW = -1>
__device__ __forceinline__ void lean2_row_core(const fedm_model_desc *__restrict__ md, int row_rt, const LeanCell &lc,
                                               const int (&lv)[3], const double (&G)[3][2], const double (&W)[3],
                                               const double *__restrict__ Ul,
                                               const double *__restrict__ Hl, const double *__restrict__ Al,
                                               const StepCoef sc, double *__restrict__ acc, double *__restrict__ Fl,
                                               const double *__restrict__ cst, int stride, uint32_t cmask = 0) {
    constexpr int NEQ = NS + 1, IPHI = NS;
    const int row = ROW >= 0 ? ROW : row_rt;
    const int wj0 = lc.wj0, wj1 = lc.wj1, wj2 = lc.wj2;
    const uint32_t rmask = cmask >> (row * NEQ);   // this row's planes
    const double E[2] = {cst[0 * stride], cst[1 * stride]};
    // a0 a1 a2 per species: exp((u0+u1+u2)/6)
    double P[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) P[i] = Al[lv[0] * NS + i] * Al[lv[1] * NS + i] * Al[lv[2] * NS + i];

    if (row == NS) {
        // Poisson row (fedm/functions.py:401): 2 pi r (grad Phi . grad v - sum_i Z_i e n_i / eps0 v)
        double m2[NS][6], m1h[3] = {0.0, 0.0, 0.0}, m01 = 0.0;
#pragma unroll
        for (int i = 0; i < NS; ++i)
#pragma unroll
            for (int k = 0; k < 6; ++k) m2[i][k] = 0.0;
        const double coe = md->charge_over_eps;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            double h = 0.0, g[NS];
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const double a = Al[lv[q] * NS + i];
                const double cz = md->Z[i] * (P[i] * (a * a * a)) * coe;
                h -= cz;
                g[i] = -cz;
            }
            const double Wq = W[q];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const double pa = Wq * (a == q ? 2.0 / 3.0 : 1.0 / 6.0);
                m1h[a] += pa * h;
                if constexpr (JAC) {
#pragma unroll
                    for (int b = a; b < 3; ++b) {
                        const double pp = pa * (b == q ? 2.0 / 3.0 : 1.0 / 6.0);
#pragma unroll
                        for (int i = 0; i < NS; ++i) m2[i][sym6(a, b)] += pp * g[i];
                    }
                }
            }
            m01 += Wq;
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int lane = lv[a];
            if (lane >= SLICE) continue;
            unsafeAtomicAdd(&Fl[lane * NEQ + IPHI], m1h[a] - (E[0] * G[a][0] + E[1] * G[a][1]) * m01);
            if constexpr (JAC) {
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    const int k = sym6(a, b);
                    const double ggk = G[a][0] * G[b][0] + G[a][1] * G[b][1];
                    const int e = a * 3 + b;
                    const int jab = ((e < 4 ? wj0 : e < 8 ? wj1 : wj2) >> (8 * (e & 3))) & 255;
                    double *dst = acc + (size_t)jab * NEQ * SLICE + lane;
#pragma unroll
                    for (int i = 0; i < NS; ++i) unsafeAtomicAdd(&dst[i * SLICE], m2[i][k]);
                    if (!((rmask >> IPHI) & 1u)) unsafeAtomicAdd(&dst[IPHI * SLICE], ggk * m01);
                }
            }
        }
        return;
    }

    const int s = row;
    const double invEm = cst[2 * stride], lnE = cst[3 * stride];
    const double Em = (E[0] * E[0] + E[1] * E[1]) * invEm;
    const int nreac = md->n_reactions;
    double kv[NR], kd[NR];
    if constexpr (LeanStash<NR>::K) {
        kv[0] = cst[LeanStash<NR>::KV * stride];
        kd[0] = cst[(LeanStash<NR>::KV + 1) * stride];
    } else {
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            kv[j] = kd[j] = 0.0;
            if (j < nreac) termsum_eval(md->k[j], Em, invEm, lnE, kv[j], kd[j]);
        }
    }
    double Us[3], Hs[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        Us[a] = Ul[lv[a] * NEQ + s];
        Hs[a] = Hl[lv[a] * NS + s];
    }
    const int eq = md->eq_type[s];
    const bool flux = eq != FEDM_EQ_REACTION;
    double Dv = 0.0, Dd = 0.0, muv = 0.0, mud = 0.0, vel[2] = {0.0, 0.0}, gradu[2] = {0.0, 0.0};
    bool fdrift = false;
    const double Z = md->Z[s];
    if (flux) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            gradu[0] += Us[a] * G[a][0];
            gradu[1] += Us[a] * G[a][1];
        }
        termsum_eval(md->D[s], Em, invEm, lnE, Dv, Dd);
        vel[0] = -Dv * gradu[0];
        vel[1] = -Dv * gradu[1];
        if (eq == FEDM_EQ_DRIFT_DIFFUSION_REACTION) {
            if (md->has_drift_w[s]) {
                vel[0] += md->drift_w[s][0];
                vel[1] += md->drift_w[s][1];
            } else {
                termsum_eval(md->mu[s], Em, invEm, lnE, muv, mud);
                vel[0] += Z * muv * E[0];
                vel[1] += Z * muv * E[1];
                fdrift = true;
            }
        }
    }
    double m2[NS][6], m1h[3] = {0.0, 0.0, 0.0}, m1n[3] = {0.0, 0.0, 0.0}, m1sp[3] = {0.0, 0.0, 0.0}, m0n = 0.0;
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int k = 0; k < 6; ++k) m2[i][k] = 0.0;
    const double usum6 = (Us[0] + Us[1] + Us[2]) * (1.0 / 6.0), hsum6 = (Hs[0] + Hs[1] + Hs[2]) * (1.0 / 6.0);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        __builtin_amdgcn_sched_barrier(0);
        double nqq[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const double a = Al[lv[q] * NS + i];
            nqq[i] = P[i] * (a * a * a);
        }
        const double ns_ = pick<NS>(nqq, s);
        const double u_part = sc.c_new * (usum6 + 0.5 * Us[q]) + (hsum6 + 0.5 * Hs[q]);
        double h = ns_ * u_part * sc.inv_dt, sp = 0.0, g[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) g[i] = (i == s) ? ns_ * (u_part + sc.c_new) * sc.inv_dt : 0.0;
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            if (j >= nreac) break;
            const double nu = (double)md->net[j][s];
            if (nu == 0.0) continue;
            double prod = 1.0;
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const int Pw = md->power[j][i];
                for (int e = 0; e < Pw; ++e) prod *= nqq[i];
            }
            h -= nu * kv[j] * prod;
            sp += nu * kd[j] * prod;
            if constexpr (JAC) {
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    const int Pw = md->power[j][i];
                    if (Pw) g[i] -= nu * kv[j] * (double)Pw * prod;
                }
            }
        }
        const double Wq = W[q];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double pa = Wq * (a == q ? 2.0 / 3.0 : 1.0 / 6.0);
            m1h[a] += pa * h;
            if constexpr (JAC) {
                m1n[a] += pa * ns_;
                m1sp[a] += pa * sp;
#pragma unroll
                for (int b = a; b < 3; ++b) {
                    const double pp = pa * (b == q ? 2.0 / 3.0 : 1.0 / 6.0);
#pragma unroll
                    for (int i = 0; i < NS; ++i)
                        if (!((rmask >> i) & 1u)) m2[i][sym6(a, b)] += pp * g[i];   // wave-uniform: planes kept
                }
            }
        }
        m0n += Wq * ns_;
    }
    double velG[3] = {0.0, 0.0, 0.0}, T[3] = {m1sp[0], m1sp[1], m1sp[2]}, DGm = 0.0, Kd = 0.0;
    if (flux) {
        DGm = Dv * m0n;
        const double zmud = fdrift ? Z * mud : 0.0;
        if (fdrift) Kd = Z * muv * m0n;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            velG[a] = vel[0] * G[a][0] + vel[1] * G[a][1];
            if constexpr (JAC) {
                const double Pa = gradu[0] * G[a][0] + gradu[1] * G[a][1];
                const double Qa = E[0] * G[a][0] + E[1] * G[a][1];
                T[a] += (zmud * Qa - Dd * Pa) * m0n;
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        __builtin_amdgcn_sched_barrier(0);
        const int lane = lv[a];
        if (lane >= SLICE) continue;  // row vertex owned by another patch
        unsafeAtomicAdd(&Fl[lane * NEQ + s], m1h[a] - velG[a] * m0n);
        if constexpr (JAC) {
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int k = sym6(a, b);
                const double ggk = G[a][0] * G[b][0] + G[a][1] * G[b][1];
                const double dEb = -(E[0] * G[b][0] + E[1] * G[b][1]) * invEm;
                const int e = a * 3 + b;
                const int jab = ((e < 4 ? wj0 : e < 8 ? wj1 : wj2) >> (8 * (e & 3))) & 255;
                double *dst = acc + (size_t)jab * NEQ * SLICE + lane;
#pragma unroll
                for (int i = 0; i < NS; ++i)
                    if (!((rmask >> i) & 1u))
                        unsafeAtomicAdd(&dst[i * SLICE], m2[i][k] + ((i == s) ? DGm * ggk - velG[a] * m1n[b] : 0.0));
                unsafeAtomicAdd(&dst[IPHI * SLICE], Kd * ggk - dEb * T[a]);
            }
        }
    }
}

// one equation row with the geometry recomputed (F + J: nothing but the packed indices lives across rows)
template <int NS, int NR, bool JAC, int ROW = -1>
__device__ __forceinline__ void lean2_row(const fedm_model_desc *__restrict__ md, int row, const LeanCell &lc,
                                          const double *__restrict__ vx, const double *__restrict__ Ul,
                                          const double *__restrict__ Hl, const double *__restrict__ Al,
                                          const StepCoef sc, double *__restrict__ acc, double *__restrict__ Fl,
                                          const double *__restrict__ cst, int stride, uint32_t cmask) {
    int lv[3];
    double G[3][2], W[3];
    lean2_geometry(md, lc, vx, lv, G, W, cst[LeanStash<NR>::IDET * stride]);
    lean2_row_core<NS, NR, JAC, ROW>(md, row, lc, lv, G, W, Ul, Hl, Al, sc, acc, Fl, cst, stride, cmask);
}

}  // namespace fedm
