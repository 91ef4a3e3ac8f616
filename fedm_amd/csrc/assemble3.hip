// Third generation of the LDS-patch assembly (Problem.F / Problem.J, fedm/functions.py:188-202) for the LFA
// models with a Poisson row and FIAT's 3-point rule: ONE pass over a patch's cells.
//
// The second generation (kernels.hip, assemble_lean2_kernel) bounded the live state by one equation row and
// paid for it: the rows were phases of the workgroup (two barriers and a stream-out each), every phase
// re-read the cell's nodal values from the staging area by run-time index, rebuilt its geometry and its
// densities at the quadrature points, and the run-time plane mask put a scalar test in front of every
// accumulation -- 1 950 vector and 800 scalar instructions per wave, a wave parked at a barrier or a wait
// for 45 % of its life.  Here a thread evaluates its cell once:
//   * geometry, field, rate coefficients, exp(u) at the quadrature points, grad(phi_a).grad(phi_b) and
//     E.grad(phi_a) are formed once and shared by the rows, which follow one another in straight-line code;
//   * the accumulators hold ALL rows of the slice, but only the planes that can change: the plane mask
//     (potential-potential: geometry only; species planes no reaction couples) is a template parameter, so a
//     kept plane costs neither LDS nor a test, and every accumulation is a ds_add_f64 with an immediate offset
//     from one of nine per-(a, b) addresses formed once per cell;
//   * the moments of the P1 basis at the three points are closed forms of the three weighted point values
//     (phi_a(q) = 1/6 + [a = q]/2): S/36 + 5/12 X_a on the diagonal, S/36 + (X_a + X_b)/12 off it;
//   * two barriers per workgroup (after staging, before the stream-out); the overlap of a workgroup's
//     stream-out with arithmetic comes from the three to four workgroups that share a CU.
// Same element tensors as element.hpp / element_lean.hpp (held against each other and against the oracle in
// tests/test_gpu_parity.py, tests/test_gpu_unstructured.py), same patch tables, same micro-coloured cell order.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "element_lean.hpp"
#include "fedm_internal.hpp"
#include "amg.hpp"
#include "species_planes.hpp"

namespace fedm {

// -DFEDM_LEAN3_PROBE=bits: an experiment build whose kernel leaves out a phase (WRONG results): 1 no stream-out
// stores, 2 the LDS accumulation as plain stores instead of atomics, 4 no cell arithmetic at all -- what each
// phase costs (tools/kernel_ab.py with FEDM_HIP_LIB).
#ifndef FEDM_LEAN3_PROBE
#define FEDM_LEAN3_PROBE 0
#endif
#if FEDM_LEAN3_PROBE & 32
#define termsum_eval(ts, E, iE, lE, v, d) do { v = (ts).c[0] * (E); d = (ts).c[0] * (iE); } while (0)
#endif
#if FEDM_LEAN3_PROBE & 2
#define LEAN3_ADD(ptr, v) (*(ptr) = (v))
#else
#define LEAN3_ADD(ptr, v) unsafeAtomicAdd(ptr, v)
#endif

// Planes of the (row, col) block that are accumulated and streamed out: those NOT in CMASK, numbered in
// row-major order.
template <int NS, uint32_t CMASK>
struct LivePlanes {
    static constexpr int NEQ = NS + 1, NEQ2 = NEQ * NEQ;
    static constexpr bool live(int r, int c) { return !((CMASK >> (r * NEQ + c)) & 1u); }
    static constexpr int index(int r, int c) {
        int n = 0;
        for (int k = 0; k < r * NEQ + c; ++k) n += ((CMASK >> k) & 1u) ? 0 : 1;
        return n;
    }
    static constexpr int N = index(NEQ - 1, NEQ - 1) + (live(NEQ - 1, NEQ - 1) ? 1 : 0);
    // (row * NEQ + col) of live plane p, four bits each
    static constexpr uint64_t packed() {
        uint64_t w = 0;
        int p = 0;
        for (int k = 0; k < NEQ2; ++k)
            if (!((CMASK >> k) & 1u)) w |= (uint64_t)k << (4 * p++);
        return w;
    }
};

// -DFEDM_PHASE_TIMING: wave 0's view of a workgroup's phases (100 MHz wall clock), summed over all workgroups:
// tools/phase_time3.py
#ifdef FEDM_PHASE_TIMING
__device__ unsigned long long g_phase3[8];
#define LEAN3_T(k) if (threadIdx.x == 0) { const unsigned long long now_ = wall_clock64(); atomicAdd(&g_phase3[k], now_ - t_prev_); t_prev_ = now_; }
extern "C" void fedm_debug_phase3(unsigned long long *out, int reset) {
    hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase3), sizeof(unsigned long long) * 8);
    if (reset) {
        unsigned long long z[8] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(g_phase3), z, sizeof(z));
    }
}
#else
#define LEAN3_T(k)
#endif

struct Lean3Params {
    int nv;
    const int *boff, *cell_ptr;
    const PatchCell *pcells;
    const int *halo_ptr, *halo;
    const double *coords, *u, *uold, *uold1;
    StepCoef sc;
    double *val, *F;
    int acc_doubles, max_verts, xcd;
    const int *patch_list;
    int n_patches;         // (persistent kernels: the grid is smaller than the number of patches)
    // the field split's planes from the accumulators (s16 = nullptr: not wanted): Ctx::planes_fused
    _Float16 *s16;
    float *val32;
    double *dinv;
    const uint32_t *diag_slot;
    int planes_upper;
    unsigned planes_zs;
};

__device__ __forceinline__ int lean3_xcd_contiguous(int b, int n) {
    const int per = n >> 3, full = per << 3;
    return b < full ? (b & 7) * per + (b >> 3) : b;
}

// The moments sum_q X_q phi_a(q) [phi_b(q)] of FIAT's 3-point rule from the three weighted point values.
__device__ __forceinline__ void p1_moments1(const double (&X)[3], double (&m)[3]) {
    const double s6 = (X[0] + X[1] + X[2]) * (1.0 / 6.0);
#pragma unroll
    for (int a = 0; a < 3; ++a) m[a] = fma(0.5, X[a], s6);
}
__device__ __forceinline__ void p1_moments2(const double (&X)[3], double (&m)[6]) {
    const double s36 = (X[0] + X[1] + X[2]) * (1.0 / 36.0);
#pragma unroll
    for (int a = 0; a < 3; ++a) m[a] = fma(5.0 / 12.0, X[a], s36);
    m[3] = fma(1.0 / 12.0, X[0] + X[1], s36);   // sym6(0, 1)
    m[4] = fma(1.0 / 12.0, X[0] + X[2], s36);   // sym6(0, 2)
    m[5] = fma(1.0 / 12.0, X[1] + X[2], s36);   // sym6(1, 2)
}

// ---------------------------------------------------------------------------------------------------------------
// The model compiled for these kernels (host: lean3_build_plan).  fedm_model_desc holds the deck's coefficient
// functions as general sums  sum_i c_i E^p_i exp(q_i E^r_i)  and the kernels of the earlier generations walked
// them term by term: a dependent scalar load from the descriptor, a wait and a branch per term and property, one
// exponential per term -- 13 of the F + J kernel's 74 us on the refined bench mesh (probe builds).  Here the
// functions arrive in the kernel arguments (scalar registers from the start, no descriptor loads at all) as
// products of a few ATOMS shared by all terms of the model:  E^p = E^m * exp(pf ln E)  with m integer and
// 0 <= pf < 1 (the streamer's E^0.74, E^-0.26, E^-2.26 share pf = 0.74), and  exp(q / E)  per distinct q -- three
// exponentials per cell for the streamer deck instead of five.  Models whose functions do not fit (r other than
// -1, more than two atoms of a kind, more than three terms) stay on the second generation.
// ---------------------------------------------------------------------------------------------------------------
constexpr int L3_TERMS = 3, L3_ATOMS = 2;
struct Lean3Slot {
    int n;                 // terms
    int code[L3_TERMS];    // bits 0-1: power atom (0 none, 1, 2), bits 2-3: exp atom, bits 4-8: m + 8
    double c[L3_TERMS], p[L3_TERMS], q[L3_TERMS];
};
template <int NS, int NR>
struct Lean3Plan {
    int nP, nQ, nreac, axisymmetric;
    double pf[L3_ATOMS], qv[L3_ATOMS];
    double coe;                              // e / eps0
    int eq_type[NS], has_drift_w[NS];
    double Z[NS], drift_w[NS][2];
    int power[NR][NS], net[NR][NS];
    Lean3Slot k[NR], D[NS], mu[NS];
};

struct Lean3Atoms {
    double P[L3_ATOMS], Q[L3_ATOMS], E, invE;
};

// ---------------------------------------------------------------------------------------------------------------
// The STRUCTURE of a model at compile time.  The reference hands every form to FFC, which generates code for exactly
// that form; the kernels above this line take the structure from the plan at run time instead -- which equation a
// species has, which reactions enter a row with which powers, how many terms a coefficient function has and which
// atoms each multiplies -- as wave-uniform scalar loads, waits and ~120 scalar branches per cell.  A signature type
// states the structure as constexpr functions; a kernel instantiated for it keeps only the NUMBERS (coefficients,
// exponents, charges) at run time: F + J 71 -> 61 us, residual only 30 -> 24 us on the refined bench mesh (back to
// back).  Lean3SigRuntime: any model the plan can hold.  A context takes a precompiled signature when its plan's
// structure equals it (lean3_sig_matches); FEDM_LEAN3_SIG=0 keeps the run-time structure;
// FEDM_LEAN3_PRINT_SIG=1 prints a plan's structure in the form of the table below.
// Slots: [0, NR) rate coefficients, NR + s diffusion of species s, NR + NS + s mobility of species s.
// ---------------------------------------------------------------------------------------------------------------
struct Lean3SigRuntime {
    static constexpr bool fixed = false;
    static constexpr int eq_type(int) { return 0; }
    static constexpr int has_drift_w(int) { return 0; }
    static constexpr int nreac() { return 0; }
    static constexpr int net(int, int) { return 0; }
    static constexpr int power(int, int) { return 0; }
    static constexpr int nP() { return 0; }
    static constexpr int nQ() { return 0; }
    static constexpr int slot_n(int) { return 0; }
    static constexpr int slot_code(int, int) { return 0; }
};
// tests/integrated_tests/streamer_discharge/file_input/benchmark_model (BASELINE configs[1], [3], [4]): ions with a
// source only, electrons drift-diffusion-reaction, one reaction e -> e + e + ion with k = alpha(E) mu_e(E) E,
//   mu_e = c E^-0.26, D_e = c E^0.22, alpha = (c1 + c2 E^-3) exp(q / E) + c3:  atoms E^0.22, E^0.74, exp(q / E)
struct Lean3SigBenchmark {
    static constexpr bool fixed = true;
    static constexpr int NS = 2, NR = 1;
    static constexpr int eq_type(int s) { return s == 0 ? FEDM_EQ_REACTION : FEDM_EQ_DRIFT_DIFFUSION_REACTION; }
    static constexpr int has_drift_w(int) { return 0; }
    static constexpr int nreac() { return 1; }
    static constexpr int net(int, int) { return 1; }
    static constexpr int power(int, int i) { return i == 1 ? 1 : 0; }
    static constexpr int nP() { return 2; }
    static constexpr int nQ() { return 1; }
    static constexpr int slot_n(int slot) { return slot == 0 ? 3 : (slot == 2 || slot == 4) ? 1 : 0; }
    // code = power atom | exp atom << 2 | (m + 8) << 4 of  c E^m P Q  (terms in ascending code)
    static constexpr int slot_code(int slot, int t) {
        return slot == 0   ? (t == 0 ? (2 | 1 << 2 | 5 << 4) : t == 1 ? (2 | 8 << 4) : (2 | 1 << 2 | 8 << 4))
               : slot == 2 ? (1 | 8 << 4)      // D_e = c E^0.22
               : slot == 4 ? (2 | 7 << 4)      // mu_e = c E^-1 E^0.74
                           : 0;
    }
};


// value and d/dE of one coefficient function (all tests are on scalar registers: wave-uniform)
template <class SIG, int SLOT>
__device__ __forceinline__ void lean3_coef(const Lean3Slot *__restrict__ slot, const Lean3Atoms &at, double &val, double &der) {
    const Lean3Slot sl = *slot;    // the whole record at once: two wide scalar loads and ONE wait, not one per term and property
    val = 0.0;
    der = 0.0;
#pragma unroll
    for (int t = 0; t < L3_TERMS; ++t) {
        if (t >= (SIG::fixed ? SIG::slot_n(SLOT) : sl.n)) break;
        const int code = SIG::fixed ? SIG::slot_code(SLOT, t) : sl.code[t];
        const int iP = code & 3, iQ = (code >> 2) & 3, m = ((code >> 4) & 31) - 8;
        double x = sl.c[t];
        if (iP) x *= (iP == 1 ? at.P[0] : at.P[1]);
        if (iQ) x *= (iQ == 1 ? at.Q[0] : at.Q[1]);
        const double f = m < 0 ? at.invE : at.E;
        for (int e = 0; e < (m < 0 ? -m : m); ++e) x *= f;
        val += x;
        der += x * (sl.p[t] - sl.q[t] * at.invE) * at.invE;
    }
}

// What the rows of a cell share.
template <int NS, int NR>
struct Lean3Cell {
    int lv[3];
    double G[3][2], W[3], m01;
    double E[2], invEm, lnE, Em;
    double Qa[3];          // E . grad(phi_a)
    double gg[6];          // grad(phi_a) . grad(phi_b), sym6 order
    double nq[NS][3];      // exp(u_i) at the quadrature points
    double kv[NR], kd[NR];
    double rq[NR][3];      // prod_i n_i^P_ji at the quadrature points (fedm/functions.py:835-843)
    Lean3Atoms at;
};

// Species row S (fedm/functions.py:350-368 with Flux :219-237 and the source of :777-843).
template <int NS, int NR, uint32_t CMASK, bool JAC, int S, class SIG>
__device__ __forceinline__ void lean3_species_row(const Lean3Plan<NS, NR> *__restrict__ md, const Lean3Cell<NS, NR> &c,
                                                  const double *__restrict__ Ul, const double *__restrict__ Hl,
                                                  const StepCoef sc, const uint32_t (&dst)[9], double *__restrict__ Fl,
                                                  char *__restrict__ lds_base) {
    constexpr int NEQ = NS + 1, IPHI = NS;
    using PL = LivePlanes<NS, CMASK>;
    double Us[3], Hs[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        Us[a] = Ul[c.lv[a] * NEQ + S];
        Hs[a] = Hl[c.lv[a] * NS + S];
    }
    const int eq = SIG::fixed ? SIG::eq_type(S) : md->eq_type[S];
    const bool flux = eq != FEDM_EQ_REACTION;
    double Dv = 0.0, Dd = 0.0, muv = 0.0, mud = 0.0, vel[2] = {0.0, 0.0}, gradu[2] = {0.0, 0.0};
    bool fdrift = false;
    const double Z = md->Z[S];
    if (flux) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            gradu[0] += Us[a] * c.G[a][0];
            gradu[1] += Us[a] * c.G[a][1];
        }
        lean3_coef<SIG, NR + S>(&md->D[S], c.at, Dv, Dd);
        vel[0] = -Dv * gradu[0];
        vel[1] = -Dv * gradu[1];
        if (eq == FEDM_EQ_DRIFT_DIFFUSION_REACTION) {
            if (SIG::fixed ? SIG::has_drift_w(S) : md->has_drift_w[S]) {
                vel[0] += md->drift_w[S][0];
                vel[1] += md->drift_w[S][1];
            } else {
                lean3_coef<SIG, NR + NS + S>(&md->mu[S], c.at, muv, mud);
                vel[0] += Z * muv * c.E[0];
                vel[1] += Z * muv * c.E[1];
                fdrift = true;
            }
        }
    }
    // weighted point values: H (residual integrand), N (density), SP (d source / d|E|), Xg[i] (d integrand / d u_i)
    const double usum6 = (Us[0] + Us[1] + Us[2]) * (1.0 / 6.0), hsum6 = (Hs[0] + Hs[1] + Hs[2]) * (1.0 / 6.0);
    const int nreac = SIG::fixed ? SIG::nreac() : md->nreac;
    double XH[3], XN[3], XS[3], XG[NS][3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const double ns_ = c.nq[S][q];
        const double u_part = sc.c_new * (usum6 + 0.5 * Us[q]) + (hsum6 + 0.5 * Hs[q]);
        double h = ns_ * u_part * sc.inv_dt, sp = 0.0, g[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) g[i] = (i == S) ? ns_ * (u_part + sc.c_new) * sc.inv_dt : 0.0;
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            if (j >= nreac) break;
            const double nu = (double)(SIG::fixed ? SIG::net(j, S) : md->net[j][S]);
            if (nu == 0.0) continue;
            const double prod = c.rq[j][q];
            const double nk = nu * c.kv[j] * prod;
            h -= nk;
            sp += nu * c.kd[j] * prod;
            if constexpr (JAC) {
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    const int Pw = SIG::fixed ? SIG::power(j, i) : md->power[j][i];
                    if (Pw) g[i] -= nk * (double)Pw;
                }
            }
        }
        const double Wq = c.W[q];
        XH[q] = Wq * h;
        XN[q] = Wq * ns_;
        XS[q] = Wq * sp;
#pragma unroll
        for (int i = 0; i < NS; ++i) XG[i][q] = Wq * g[i];
    }
    double m1h[3];
    p1_moments1(XH, m1h);
    const double m0n = XN[0] + XN[1] + XN[2];
    double velG[3] = {0.0, 0.0, 0.0};
    if (flux) {
#pragma unroll
        for (int a = 0; a < 3; ++a) velG[a] = vel[0] * c.G[a][0] + vel[1] * c.G[a][1];
    }
    double m1n[3], T[3], m2[NS][6], DGm = 0.0, Kd = 0.0;
    if constexpr (JAC) {
        p1_moments1(XN, m1n);
        p1_moments1(XS, T);
#pragma unroll
        for (int i = 0; i < NS; ++i)
            if (PL::live(S, i)) p1_moments2(XG[i], m2[i]);
        if (flux) {
            DGm = Dv * m0n;
            const double zmud = fdrift ? Z * mud : 0.0;
            if (fdrift) Kd = Z * muv * m0n;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const double Pa = gradu[0] * c.G[a][0] + gradu[1] * c.G[a][1];
                T[a] += (zmud * c.Qa[a] - Dd * Pa) * m0n;
            }
        }
    }
    if (FEDM_LEAN3_PROBE & 16) {    // no emission: the moments kept alive by one add
        double sum = m1h[0] + m1h[1] + m1h[2] + velG[0] + velG[1] + velG[2] + m0n;
        if constexpr (JAC) {
            sum += m1n[0] + m1n[1] + m1n[2] + T[0] + T[1] + T[2] + DGm + Kd;
#pragma unroll
            for (int i = 0; i < NS; ++i)
                if (PL::live(S, i)) sum += m2[i][0] + m2[i][1] + m2[i][2] + m2[i][3] + m2[i][4] + m2[i][5];
        }
        unsafeAtomicAdd(&Fl[c.lv[0] & 63], sum);
        return;
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int lane = c.lv[a];
        if (lane >= SLICE) continue;   // row vertex owned by another patch
        LEAN3_ADD(&Fl[lane * NEQ + S], m1h[a] - velG[a] * m0n);
        if constexpr (JAC) {
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int k = sym6(a, b);
                double *d = reinterpret_cast<double *>(lds_base + dst[a * 3 + b]);
#pragma unroll
                for (int i = 0; i < NS; ++i)
                    if (PL::live(S, i))
                        LEAN3_ADD(&d[PL::index(S, i) * SLICE],
                                        m2[i][k] + ((i == S) ? DGm * c.gg[k] - velG[a] * m1n[b] : 0.0));
                if constexpr (PL::live(S, IPHI))
                    LEAN3_ADD(&d[PL::index(S, IPHI) * SLICE], Kd * c.gg[k] + c.Qa[b] * c.invEm * T[a]);
            }
        }
    }
}

// Poisson row (fedm/functions.py:401): 2 pi r (grad Phi . grad v - sum_i Z_i e n_i / eps0 v)
template <int NS, int NR, uint32_t CMASK, bool JAC>
__device__ __forceinline__ void lean3_poisson_row(const Lean3Plan<NS, NR> *__restrict__ md, const Lean3Cell<NS, NR> &c,
                                                  const uint32_t (&dst)[9], double *__restrict__ Fl,
                                                  char *__restrict__ lds_base) {
    constexpr int NEQ = NS + 1, IPHI = NS;
    using PL = LivePlanes<NS, CMASK>;
    const double coe = md->coe;
    double XH[3] = {0.0, 0.0, 0.0}, XG[NS][3];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const double zc = -md->Z[i] * coe;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            XG[i][q] = c.W[q] * (zc * c.nq[i][q]);
            XH[q] += XG[i][q];
        }
    }
    double m1h[3], m2[NS][6];
    p1_moments1(XH, m1h);
    if constexpr (JAC) {
#pragma unroll
        for (int i = 0; i < NS; ++i)
            if (PL::live(IPHI, i)) p1_moments2(XG[i], m2[i]);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int lane = c.lv[a];
        if (lane >= SLICE) continue;
        LEAN3_ADD(&Fl[lane * NEQ + IPHI], m1h[a] - c.Qa[a] * c.m01);
        if constexpr (JAC) {
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int k = sym6(a, b);
                double *d = reinterpret_cast<double *>(lds_base + dst[a * 3 + b]);
#pragma unroll
                for (int i = 0; i < NS; ++i)
                    if (PL::live(IPHI, i)) LEAN3_ADD(&d[PL::index(IPHI, i) * SLICE], m2[i][k]);
                if constexpr (PL::live(IPHI, IPHI)) LEAN3_ADD(&d[PL::index(IPHI, IPHI) * SLICE], c.gg[k] * c.m01);
            }
        }
    }
}

template <int NS, int NR, uint32_t CMASK, bool JAC, int S, class SIG>
__device__ __forceinline__ void lean3_species_rows(const Lean3Plan<NS, NR> *__restrict__ md, const Lean3Cell<NS, NR> &c,
                                                   const double *__restrict__ Ul, const double *__restrict__ Hl,
                                                   const StepCoef sc, const uint32_t (&dst)[9], double *__restrict__ Fl,
                                                   char *__restrict__ lds_base) {
    if constexpr (S < NS) {
        lean3_species_row<NS, NR, CMASK, JAC, S, SIG>(md, c, Ul, Hl, sc, dst, Fl, lds_base);
        __builtin_amdgcn_sched_barrier(0);   // the rows one after the other: nothing of row S + 1 lives beside row S
        lean3_species_rows<NS, NR, CMASK, JAC, S + 1, SIG>(md, c, Ul, Hl, sc, dst, Fl, lds_base);
    }
}

// The local indices of a cell in its patch, as the kernels keep them (PatchCell without the cell number and the tags)
struct CellIdx {
    uint32_t lv;           // local vertex ids, a byte each
    uint32_t j0, j1, j2;   // local block columns of the nine (a, b) pairs, a byte each
};
__device__ __forceinline__ CellIdx cell_idx(const PatchCell &pc) {
    CellIdx ci;
    ci.lv = pc.lv[0] | (pc.lv[1] << 8) | (pc.lv[2] << 16);
    ci.j0 = pc.j[0] | (pc.j[1] << 8) | (pc.j[2] << 16) | ((uint32_t)pc.j[3] << 24);
    ci.j1 = pc.j[4] | (pc.j[5] << 8) | (pc.j[6] << 16) | ((uint32_t)pc.j[7] << 24);
    ci.j2 = pc.j[8];
    return ci;
}

// One cell: everything its rows share, then the rows one after the other (straight-line code).
template <int NS, int NR, uint32_t CMASK, bool JAC, class SIG>
__device__ __forceinline__ void lean3_cell(const Lean3Plan<NS, NR> *__restrict__ md, const CellIdx ci,
                                           const double *__restrict__ vx, const double *__restrict__ Ul,
                                           const double *__restrict__ Hl, const double *__restrict__ Al,
                                           const StepCoef sc, double *__restrict__ acc, double *__restrict__ Fl) {
    constexpr int NEQ = NS + 1, IPHI = NS;
    using PL = LivePlanes<NS, CMASK>;
    constexpr int NPL = PL::N;
        Lean3Cell<NS, NR> c;
    double x[3][2];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        c.lv[a] = (ci.lv >> (8 * a)) & 255u;
        x[a][0] = vx[2 * c.lv[a]];
        x[a][1] = vx[2 * c.lv[a] + 1];
    }
    CellGeom cg;
    cg.init(x, md->axisymmetric);
    const double two_pi = 6.283185307179586476925286766559;
    {
        // weights of the three points times 2 pi r: r(q) = (r0 + r1 + r2)/6 + r_q/2
        const double w6 = (1.0 / 6.0) * cg.detJ * two_pi;
        const double rs6 = (cg.rn[0] + cg.rn[1] + cg.rn[2]) * (1.0 / 6.0);
#pragma unroll
        for (int q = 0; q < 3; ++q) c.W[q] = w6 * fma(0.5, cg.rn[q], rs6);
        c.m01 = c.W[0] + c.W[1] + c.W[2];
    }
    c.E[0] = c.E[1] = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        c.G[a][0] = cg.G[a][0];
        c.G[a][1] = cg.G[a][1];
        const double ph = Ul[c.lv[a] * NEQ + IPHI];
        c.E[0] -= ph * cg.G[a][0];
        c.E[1] -= ph * cg.G[a][1];
    }
    {
        // 1/|E| by reciprocal square root, |E| = E^2 / |E|, ln|E| = ln(E^2)/2
        const double E2 = c.E[0] * c.E[0] + c.E[1] * c.E[1];
        c.invEm = rsqrt(E2);
        c.Em = E2 * c.invEm;
        c.lnE = 0.5 * log(E2);
        // the atoms of the model's coefficient functions (all of the cell's exponentials)
        c.at.E = c.Em;
        c.at.invE = c.invEm;
#pragma unroll
        for (int i = 0; i < L3_ATOMS; ++i) {
            c.at.P[i] = i < (SIG::fixed ? SIG::nP() : md->nP) ? exp(md->pf[i] * c.lnE) : 1.0;
            c.at.Q[i] = i < (SIG::fixed ? SIG::nQ() : md->nQ) ? exp(md->qv[i] * c.invEm) : 1.0;
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        c.Qa[a] = c.E[0] * c.G[a][0] + c.E[1] * c.G[a][1];
        if constexpr (JAC) {
#pragma unroll
            for (int b = a; b < 3; ++b) c.gg[sym6(a, b)] = c.G[a][0] * c.G[b][0] + c.G[a][1] * c.G[b][1];
        }
    }
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        // u(q) = (u0 + u1 + u2)/6 + u_q/2: exp of it from the per-vertex factors exp(u_v / 6)
        const double a0 = Al[c.lv[0] * NS + i], a1 = Al[c.lv[1] * NS + i], a2 = Al[c.lv[2] * NS + i];
        const double P = a0 * a1 * a2;
        c.nq[i][0] = P * (a0 * a0 * a0);
        c.nq[i][1] = P * (a1 * a1 * a1);
        c.nq[i][2] = P * (a2 * a2 * a2);
    }
    {
        static_assert(NR == 1 || !SIG::fixed, "a signature's rate-coefficient slot is written for one reaction");
        const int nreac = SIG::fixed ? SIG::nreac() : md->nreac;
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            c.kv[j] = c.kd[j] = 0.0;
            c.rq[j][0] = c.rq[j][1] = c.rq[j][2] = 1.0;
            if (j < nreac) {
                lean3_coef<SIG, 0>(&md->k[j], c.at, c.kv[j], c.kd[j]);   // (slot j; NR = 1 where this is instantiated)
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    const int Pw = SIG::fixed ? SIG::power(j, i) : md->power[j][i];
                    for (int e = 0; e < Pw; ++e) {
#pragma unroll
                        for (int q = 0; q < 3; ++q) c.rq[j][q] *= c.nq[i][q];
                    }
                }
            }
        }
    }
    // byte offsets of the nine (a, b) accumulators of this cell (row vertex a owned: lane < 64)
    uint32_t dst[9];
    char *lds_base = reinterpret_cast<char *>(acc);
    if constexpr (JAC) {
#pragma unroll
        for (int e = 0; e < 9; ++e) {
            const uint32_t jab = ((e < 4 ? ci.j0 : e < 8 ? ci.j1 : ci.j2) >> (8 * (e & 3))) & 255u;
            dst[e] = (jab * (NPL * SLICE) + (uint32_t)c.lv[e / 3]) * (uint32_t)sizeof(double);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (FEDM_LEAN3_PROBE & 8) {     // no rows: what the rows share, kept alive by one add
        double sum = c.m01 + c.invEm + c.lnE + c.kv[0] + c.kd[0] + c.rq[0][0] + c.rq[0][1] + c.rq[0][2];
#pragma unroll
        for (int i = 0; i < NS; ++i) sum += c.nq[i][0] + c.nq[i][1] + c.nq[i][2];
#pragma unroll
        for (int k = 0; k < 6; ++k) sum += c.gg[k];
#pragma unroll
        for (int a = 0; a < 3; ++a) sum += c.Qa[a] + c.W[a];
        unsafeAtomicAdd(&Fl[c.lv[0] & 63], sum);
        return;
    }
    lean3_species_rows<NS, NR, CMASK, JAC, 0, SIG>(md, c, Ul, Hl, sc, dst, Fl, lds_base);
    lean3_poisson_row<NS, NR, CMASK, JAC>(md, c, dst, Fl, lds_base);
}

// The field split's set-up (amg.hip, species_planes_kernel) from the patch's accumulators instead of from the matrix
// just written: D_uu^-1 of the slice's rows, S = D_uu^-1 J_uu in half precision, the coupling plane in single
// precision -- 115 MB of the Jacobian not read back behind every assembly.  A thread takes two neighbouring rows of
// a block column (its stores are 16, 8 and 16 bytes wide); the rows' own diagonal blocks come from the accumulators
// too.  Rows changed behind this kernel (boundary facets, Dirichlet values) are redone by
// species_planes_rows_kernel; the padding rows of the last slice are the identity rows they will become.
template <int NS, uint32_t CMASK, int THREADS>
__device__ __forceinline__ void lean3_species_planes(const Lean3Params &p, const double *acc, int S, int b0, int width,
                                                     uint2 diag) {
    using PL = LivePlanes<NS, CMASK>;
    constexpr int NEQ = NS + 1, NPL = PL::N, PAIRS = SLICE / 2;
    static_assert(THREADS % PAIRS == 0, "a thread keeps its pair of rows");
    const int pair = threadIdx.x % PAIRS;
    const unsigned zs = p.planes_zs;
    // a species plane of the block: the accumulator, or 0 where the plane is not formed (kept planes of the species
    // block are structurally zero ones: launch_assemble_lean3 checks)
    auto plane = [&](int bc, int r, int cidx, int row) {
        return PL::live(r, cidx) ? acc[((size_t)bc * NPL + PL::index(r, cidx)) * SLICE + row] : 0.0;
    };
    // ... of the thread's two rows in one 16-byte read
    auto plane2 = [&](int bc, int r, int cidx) {
        return PL::live(r, cidx) ? reinterpret_cast<const double2 *>(acc + ((size_t)bc * NPL + PL::index(r, cidx)) * SLICE)[pair]
                                 : make_double2(0.0, 0.0);
    };
    double d[2][NS][NS];
    int dbc[2];
    bool pad[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int row = 2 * pair + h;
        pad[h] = S * SLICE + row >= p.nv;
        dbc[h] = (int)((h == 0 ? diag.x : diag.y) >> 6) - b0;
        double A[NS][NS];
#pragma unroll
        for (int r = 0; r < NS; ++r)
#pragma unroll
            for (int cidx = 0; cidx < NS; ++cidx)
                A[r][cidx] = pad[h] ? (r == cidx ? 1.0 : 0.0) : plane(dbc[h], r, cidx, row);
        invert_species_block<NS>(A, d[h]);
    }
    for (int bc = threadIdx.x / PAIRS; bc < width; bc += THREADS / PAIRS) {
        _Float16 row16[2][NS * NS];
        float cpl[NS][2];
        double2 Jp[NS][NS], Cp[NS];
#pragma unroll
        for (int r = 0; r < NS; ++r)
#pragma unroll
            for (int cidx = 0; cidx < NS; ++cidx) Jp[r][cidx] = plane2(bc, r, cidx);
#pragma unroll
        for (int cidx = 0; cidx < NS; ++cidx) Cp[cidx] = p.planes_upper ? plane2(bc, cidx, NS) : plane2(bc, NS, cidx);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            double J[NS][NS];
#pragma unroll
            for (int r = 0; r < NS; ++r)
#pragma unroll
                for (int cidx = 0; cidx < NS; ++cidx)
                    J[r][cidx] = ((zs >> (r * NS + cidx)) & 1u) ? 0.0
                                 : pad[h]                          ? ((r == cidx && bc == dbc[h]) ? 1.0 : 0.0)
                                                                   : (h == 0 ? Jp[r][cidx].x : Jp[r][cidx].y);
            species_plane_entry<NS>(d[h], J, zs, row16[h]);
#pragma unroll
            for (int cidx = 0; cidx < NS; ++cidx) cpl[cidx][h] = pad[h] ? 0.f : (float)(h == 0 ? Cp[cidx].x : Cp[cidx].y);
        }
        // [(bc * 64 + row) * NS^2 + plane]: the two rows' entries are neighbours
        _Float16 *dst = p.s16 + ((size_t)(b0 + bc) * SLICE + 2 * pair) * (NS * NS);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < NS * NS; ++e) dst[h * NS * NS + e] = row16[h][e];
#pragma unroll
        for (int cidx = 0; cidx < NS; ++cidx)
            reinterpret_cast<float2 *>(p.val32 + ((size_t)(b0 + bc) * NS + cidx) * SLICE)[pair] = make_float2(cpl[cidx][0], cpl[cidx][1]);
        if (bc == 0) {
#pragma unroll
            for (int e = 0; e < NS * NS; ++e)
                reinterpret_cast<double2 *>(p.dinv + ((size_t)S * NS * NS + e) * SLICE)[pair] =
                    make_double2(d[0][e / NS][e % NS], d[1][e / NS][e % NS]);
        }
    }
}

template <int NS, int NR, int THREADS, uint32_t CMASK, bool JAC, class SIG>
__device__ __forceinline__ void assemble_lean3_body(const Lean3Plan<NS, NR> *__restrict__ plan, const Lean3Params &p) {
    constexpr int NEQ = NS + 1, NEQ2 = NEQ * NEQ, IPHI = NS;
    using PL = LivePlanes<NS, CMASK>;
    constexpr int NPL = PL::N;
    extern __shared__ __align__(16) double lds[];
    double *acc = lds;                          // [width][NPL][64]: the live planes of every block column (JAC)
    double *Fl = acc + p.acc_doubles;           // [64][NEQ]
    double *vx = Fl + SLICE * NEQ;              // [max_verts][2]
    double *Ul = vx + 2 * p.max_verts;          // [max_verts][NEQ]
    double *Hl = Ul + NEQ * p.max_verts;        // [max_verts][NS]
    double *Al = Hl + NS * p.max_verts;         // [max_verts][NS]: exp(u / 6)
#ifdef FEDM_PHASE_TIMING
    unsigned long long t_prev_ = wall_clock64();
#endif
#ifdef FEDM_LEAN3_STAGGER
    // experiment: the workgroups of the first round start at different times
    if (blockIdx.x < FEDM_LEAN3_STAGGER_BLOCKS) {
        const unsigned b = blockIdx.x;
        const unsigned k = FEDM_LEAN3_STAGGER == 1 ? (b >> 8) & 3u : FEDM_LEAN3_STAGGER == 2 ? b & 3u
                           : ((b * 2654435761u) >> 13) & 3u;
        for (unsigned i = 0; i < k; ++i) __builtin_amdgcn_s_sleep(FEDM_LEAN3_STAGGER_SLEEP);
    }
#endif
    const int blk = p.xcd ? lean3_xcd_contiguous(blockIdx.x, gridDim.x) : blockIdx.x;
    const int S = p.patch_list ? p.patch_list[blk] : blk;
    const int b0 = p.boff[S], width = p.boff[S + 1] - b0;
    const int c0 = p.cell_ptr[S], n_cells = p.cell_ptr[S + 1] - c0;
    // a cell per thread; the rare patch with more cells than threads (3 of 5 333 on the refined bench mesh)
    // gives its first threads a second one
    PatchCell pc = {};
    if ((int)threadIdx.x < n_cells) pc = p.pcells[c0 + threadIdx.x];
    // (the planes epilogue's diagonal slots of the thread's two rows: asked for now, used behind the second barrier)
    uint2 diag = make_uint2(0u, 0u);
    if constexpr (JAC) {
        if (p.s16) diag = reinterpret_cast<const uint2 *>(p.diag_slot + (size_t)S * SLICE)[threadIdx.x % (SLICE / 2)];
    }
    if constexpr (JAC) {
        double2 *acc2 = reinterpret_cast<double2 *>(acc);
        const int n2 = width * NPL * (SLICE / 2);
        for (int k = threadIdx.x; k < n2; k += THREADS) acc2[k] = make_double2(0.0, 0.0);
    }
    for (int k = threadIdx.x; k < SLICE * NEQ; k += THREADS) Fl[k] = 0.0;
    LEAN3_T(0)   // header, cell record request, zeroing
    const int h0 = p.halo_ptr[S], n_local = SLICE + p.halo_ptr[S + 1] - h0;
    for (int i = threadIdx.x; i < n_local; i += THREADS) {
        const int g = (i < SLICE) ? S * SLICE + i : p.halo[h0 + i - SLICE];
        if (g < p.nv) {
            vx[2 * i] = p.coords[2 * (size_t)g];
            vx[2 * i + 1] = p.coords[2 * (size_t)g + 1];
            double un[NEQ];
#pragma unroll
            for (int s = 0; s < NEQ; ++s) un[s] = p.u[(size_t)g * NEQ + s];
#pragma unroll
            for (int s = 0; s < NS; ++s)
                Hl[i * NS + s] = p.sc.c_old * p.uold[(size_t)g * NEQ + s] + p.sc.c_old1 * p.uold1[(size_t)g * NEQ + s];
#pragma unroll
            for (int s = 0; s < NEQ; ++s) Ul[i * NEQ + s] = un[s];
#pragma unroll
            for (int s = 0; s < NS; ++s) Al[i * NS + s] = exp(un[s] * (1.0 / 6.0));
        }
    }
    LEAN3_T(1)   // staging: halo id, vertex data, exponentials, LDS writes
    __syncthreads();
    LEAN3_T(2)   // barrier 1
    auto one_cell = [&](const PatchCell &pcx) { lean3_cell<NS, NR, CMASK, JAC, SIG>(plan, cell_idx(pcx), vx, Ul, Hl, Al, p.sc, acc, Fl); };
    // Straight-line code for the thread's cell; a second copy of it (never fetched by the other workgroups) for
    // the patches with more cells than threads.  As a loop the invariants the compiler hoists out of it cost 35
    // spilled registers.
    if (!(FEDM_LEAN3_PROBE & 4) && (int)threadIdx.x < n_cells) one_cell(pc);
    if ((int)threadIdx.x + THREADS < n_cells) {     // (n_cells <= 2 THREADS: lean3_applies)
        const PatchCell pc2 = p.pcells[c0 + threadIdx.x + THREADS];
        one_cell(pc2);
    }
    LEAN3_T(3)   // the cell
    __syncthreads();
    LEAN3_T(4)   // barrier 2
    if constexpr (JAC) {
        // stream-out: every live plane of every block column is 64 consecutive doubles in the matrix
        static_assert(THREADS % (SLICE / 2) == 0, "a thread keeps its 16-byte piece of the plane");
        constexpr uint64_t RC = PL::packed();
        constexpr int STEP = THREADS / (SLICE / 2);
        const int piece = threadIdx.x % (SLICE / 2);
        const int n_pl = width * NPL;
        const double2 *src = reinterpret_cast<const double2 *>(acc) + piece;
        for (int k = threadIdx.x / (SLICE / 2); k < n_pl; k += STEP) {
            const int bc = k / NPL, pl = k - bc * NPL;
            const int rc = (int)((RC >> (4 * pl)) & 15u);
            if (FEDM_LEAN3_PROBE & 1) {
                if (src[k * (SLICE / 2)].x == 1.2345e-300) p.val[0] = 1.0;   // (keeps the LDS read alive)
                continue;
            }
            reinterpret_cast<double2 *>(p.val + ((size_t)(b0 + bc) * NEQ2 + rc) * SLICE)[piece] = src[k * (SLICE / 2)];
        }
    }
    double *fdst = p.F + (size_t)S * SLICE * NEQ;
    for (int k = threadIdx.x; k < SLICE * NEQ; k += THREADS) fdst[k] = Fl[k];
    if constexpr (JAC) {
        if (p.s16) lean3_species_planes<NS, CMASK, THREADS>(p, acc, S, b0, width, diag);
    }
    LEAN3_T(5)   // stream-out (issue)
}

// ---------------------------------------------------------------------------------------------------------------
// The same assembly as a PERSISTENT, software-pipelined kernel.  Measured on the one-shot kernel above (probe
// builds, tools/kernel_ab.py): without the cell arithmetic it takes 30 us (the refined bench mesh; 6.4 TB/s:
// the memory system's limit), the arithmetic adds 45 us, and the whole takes 75 us -- the SUM.  A workgroup's
// life was: three dependent memory round trips (patch header -> halo vertex ids and cell records -> vertex data),
// the exponentials, a barrier, the arithmetic, a barrier, the stream-out; with three to four workgroups per CU the
// waves that compute at any moment are 1.3 per SIMD, and their dependent fp64 chains leave the vector pipes idle
// half of the time.  Here a workgroup stays and takes patch after patch:
//   * the NEXT patch's header is read two patches ahead (scalar loads), the id of the halo vertex a thread will
//     stage one patch ahead, at the start of the current patch's arithmetic -- one register that lives through it;
//   * behind the arithmetic's barrier the next patch's vertex data and cell records are REQUESTED FIRST, then the
//     current patch is streamed out (and its accumulators zeroed in the same pass) while they travel; the
//     exponentials and LDS writes of the staging follow;
//   * a patch therefore costs its arithmetic plus one stream-out instead of arithmetic plus three round trips
//     plus stream-out, and the only time a wave waits for memory is the prologue of its workgroup.
// The workgroups of an XCD share a contiguous range of patches (neighbouring patches stage the same halo vertices:
// one L2) and take them round robin.
// ---------------------------------------------------------------------------------------------------------------
struct PatchHdr {
    int S, b0, width, c0, nc, h0, nloc;
};

template <int NS, int NR, int THREADS, uint32_t CMASK, bool JAC, class SIG>
__device__ __forceinline__ void assemble_lean3p_body(const Lean3Plan<NS, NR> *__restrict__ md, double *__restrict__ gval,
                                                     double *__restrict__ gF, const Lean3Params &p) {
    constexpr int NEQ = NS + 1, NEQ2 = NEQ * NEQ;
    using PL = LivePlanes<NS, CMASK>;
    constexpr int NPL = PL::N;
    extern __shared__ __align__(16) double lds[];
    double *acc = lds;                          // [max width][NPL][64]
    double *Fl = acc + p.acc_doubles;           // [64][NEQ]
    double *vx = Fl + SLICE * NEQ;              // [max_verts][2]
    double *Ul = vx + 2 * p.max_verts;          // [max_verts][NEQ]
    double *Hl = Ul + NEQ * p.max_verts;        // [max_verts][NS]
    double *Al = Hl + NS * p.max_verts;         // [max_verts][NS]: exp(u / 6)
    const int tid = threadIdx.x;
    // this workgroup's patches: first, first + stride, ... below end
    int first, stride, end;
    if (p.xcd) {
        const int x = blockIdx.x & 7, per = (p.n_patches + 7) >> 3;
        first = x * per + (blockIdx.x >> 3);
        stride = gridDim.x >> 3;                // (the grid is a multiple of 8 workgroups: lean3_launch_one)
        end = min((x + 1) * per, p.n_patches);
    } else {
        first = blockIdx.x;
        stride = gridDim.x;
        end = p.n_patches;
    }
    if (first >= end) return;
    auto load_hdr = [&](int idx) {
        PatchHdr h;
        h.S = p.patch_list ? p.patch_list[idx] : idx;
        h.b0 = p.boff[h.S];
        h.width = p.boff[h.S + 1] - h.b0;
        h.c0 = p.cell_ptr[h.S];
        h.nc = p.cell_ptr[h.S + 1] - h.c0;
        h.h0 = p.halo_ptr[h.S];
        h.nloc = SLICE + p.halo_ptr[h.S + 1] - h.h0;
        return h;
    };
    // the thread's cell of a patch and the vertex it stages (nv: none)
    auto load_cell = [&](int t, const PatchHdr &h) {
        CellIdx ci = {0, 0, 0, 0};
        if (t < h.nc) ci = cell_idx(p.pcells[h.c0 + t]);
        return ci;
    };
    auto load_gid = [&](int t, const PatchHdr &h) {
        return t < h.nloc ? (t < SLICE ? h.S * SLICE + t : p.halo[h.h0 + t - SLICE]) : p.nv;
    };
    struct VData {
        double x0, x1, un[NEQ], uo[NS], uoo[NS];
    };
    auto load_data = [&](int gid) {
        VData d = {};
        if (gid < p.nv) {
            d.x0 = p.coords[2 * (size_t)gid];
            d.x1 = p.coords[2 * (size_t)gid + 1];
#pragma unroll
            for (int s = 0; s < NEQ; ++s) d.un[s] = p.u[(size_t)gid * NEQ + s];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                d.uo[s] = p.uold[(size_t)gid * NEQ + s];
                d.uoo[s] = p.uold1[(size_t)gid * NEQ + s];
            }
        }
        return d;
    };
    auto stage = [&](int t, int gid, const VData &d) {
        if (gid < p.nv) {
            vx[2 * t] = d.x0;
            vx[2 * t + 1] = d.x1;
#pragma unroll
            for (int s = 0; s < NS; ++s) Hl[t * NS + s] = p.sc.c_old * d.uo[s] + p.sc.c_old1 * d.uoo[s];
#pragma unroll
            for (int s = 0; s < NEQ; ++s) Ul[t * NEQ + s] = d.un[s];
#pragma unroll
            for (int s = 0; s < NS; ++s) Al[t * NS + s] = exp(d.un[s] * (1.0 / 6.0));
        }
    };
    // ---- prologue: the first patch the slow way ----
    PatchHdr h = load_hdr(first);
    CellIdx ci = load_cell(tid, h);
    const int gid = load_gid(tid, h);
    if constexpr (JAC) {
        double2 *acc2 = reinterpret_cast<double2 *>(acc);
        for (int k = tid; k < p.acc_doubles / 2; k += THREADS) acc2[k] = make_double2(0.0, 0.0);
    }
    for (int k = tid; k < SLICE * NEQ; k += THREADS) Fl[k] = 0.0;
    {
        const VData d = load_data(gid);
        stage(tid, gid, d);
    }
    int idx = first;
    bool has_next = idx + stride < end;
    PatchHdr hn = has_next ? load_hdr(idx + stride) : h;
    __syncthreads();
    while (true) {
        // the next patch's indices and the header of the one after it: in flight during this patch's arithmetic
        // (t: an opaque copy of the thread index per patch -- the addresses derived from it would otherwise be kept in
        // registers across the arithmetic as loop invariants)
        int t = tid;
        asm volatile("" : "+v"(t));
        const int gidn = has_next ? load_gid(t, hn) : p.nv;
        const bool has_next2 = idx + 2 * stride < end;
        const PatchHdr hn2 = has_next2 ? load_hdr(idx + 2 * stride) : hn;
        if (!(FEDM_LEAN3_PROBE & 4) && tid < h.nc) lean3_cell<NS, NR, CMASK, JAC, SIG>(md, ci, vx, Ul, Hl, Al, p.sc, acc, Fl);
        if (tid + THREADS < h.nc) {     // (n_cells <= 2 THREADS: lean3_applies)
            const CellIdx ci2 = cell_idx(p.pcells[h.c0 + tid + THREADS]);
            lean3_cell<NS, NR, CMASK, JAC, SIG>(md, ci2, vx, Ul, Hl, Al, p.sc, acc, Fl);
        }
        __syncthreads();
        asm volatile("" : "+v"(t));
        // the next patch's vertex data and cell records: issued before this patch is streamed out
        VData dn = {};
        CellIdx cin = {0, 0, 0, 0};
        if (has_next) {
            dn = load_data(gidn);
            cin = load_cell(t, hn);
        }
        if constexpr (JAC) {
            static_assert(THREADS % (SLICE / 2) == 0, "a thread keeps its 16-byte piece of the plane");
            constexpr uint64_t RC = PL::packed();
            constexpr int STEP = THREADS / (SLICE / 2);
            const int piece = t % (SLICE / 2);
            const int n_pl = h.width * NPL;
            double2 *src = reinterpret_cast<double2 *>(acc) + piece;
            for (int k = t / (SLICE / 2); k < n_pl; k += STEP) {
                const int bc = k / NPL, pl = k - bc * NPL;
                const int rc = (int)((RC >> (4 * pl)) & 15u);
                const double2 v = src[k * (SLICE / 2)];
                src[k * (SLICE / 2)] = make_double2(0.0, 0.0);
                if (FEDM_LEAN3_PROBE & 1) {
                    if (v.x == 1.2345e-300) gval[0] = 1.0;
                    continue;
                }
                reinterpret_cast<double2 *>(gval + ((size_t)(h.b0 + bc) * NEQ2 + rc) * SLICE)[piece] = v;
            }
        }
        {
            double *fdst = gF + (size_t)h.S * SLICE * NEQ;
            for (int k = t; k < SLICE * NEQ; k += THREADS) {
                fdst[k] = Fl[k];
                Fl[k] = 0.0;
            }
        }
        if (!has_next) break;
        stage(t, gidn, dn);
        __syncthreads();
        h = hn;
        hn = hn2;
        ci = cin;
        idx += stride;
        has_next = has_next2;
    }
}

template <int NS, int NR, int THREADS, uint32_t CMASK, class SIG>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(3, 3))) void assemble_lean3p_kernel(
    const Lean3Plan<NS, NR> *__restrict__ plan, double *__restrict__ val, double *__restrict__ F, const Lean3Params p) {
    assemble_lean3p_body<NS, NR, THREADS, CMASK, true, SIG>(plan, val, F, p);
}

template <int NS, int NR, int THREADS, class SIG>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void residual_lean3p_kernel(
    const Lean3Plan<NS, NR> *__restrict__ plan, double *__restrict__ val, double *__restrict__ F, const Lean3Params p) {
    assemble_lean3p_body<NS, NR, THREADS, 0u, false, SIG>(plan, val, F, p);
}

#ifndef FEDM_LEAN3_WAVES
#define FEDM_LEAN3_WAVES 3
#endif
template <int NS, int NR, int THREADS, uint32_t CMASK, class SIG>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(FEDM_LEAN3_WAVES, FEDM_LEAN3_WAVES))) void assemble_lean3_kernel(
    const Lean3Plan<NS, NR> *__restrict__ plan, const Lean3Params p) {
    assemble_lean3_body<NS, NR, THREADS, CMASK, true, SIG>(plan, p);
}

#ifndef FEDM_RES3_WAVES
#define FEDM_RES3_WAVES 4
#endif
template <int NS, int NR, int THREADS, class SIG>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(FEDM_RES3_WAVES, FEDM_RES3_WAVES))) void residual_lean3_kernel(
    const Lean3Plan<NS, NR> *__restrict__ plan, const Lean3Params p) {
    assemble_lean3_body<NS, NR, THREADS, 0u, false, SIG>(plan, p);
}

// fedm_model_desc -> Lean3Plan.  false: the model's coefficient functions do not fit the plan's form.
template <int NS, int NR>
static bool lean3_build_plan(const fedm_model_desc &m, Lean3Plan<NS, NR> &pl) {
    if (m.n_species != NS || m.n_reactions > NR || !m.poisson) return false;
    pl = Lean3Plan<NS, NR>{};
    pl.nreac = m.n_reactions;
    pl.axisymmetric = m.axisymmetric;
    pl.coe = m.charge_over_eps;
    for (int s = 0; s < NS; ++s) {
        pl.eq_type[s] = m.eq_type[s];
        pl.has_drift_w[s] = m.has_drift_w[s];
        pl.Z[s] = m.Z[s];
        pl.drift_w[s][0] = m.drift_w[s][0];
        pl.drift_w[s][1] = m.drift_w[s][1];
    }
    for (int j = 0; j < NR; ++j)
        for (int s = 0; s < NS; ++s) {
            pl.power[j][s] = j < m.n_reactions ? m.power[j][s] : 0;
            pl.net[j][s] = j < m.n_reactions ? m.net[j][s] : 0;
        }
    auto atom = [](double *list, int &n, double v) {     // index + 1 of v in the list (added when new); 0: no room
        for (int i = 0; i < n; ++i)
            if (std::fabs(list[i] - v) <= 1e-13 * std::max(1.0, std::fabs(v))) return i + 1;
        if (n >= L3_ATOMS) return 0;
        list[n] = v;
        return ++n;
    };
    auto slot = [&](const fedm_termsum &ts, Lean3Slot &sl) {
        sl.n = 0;
        for (int i = 0; i < ts.n_terms; ++i) {
            if (ts.c[i] == 0.0) continue;
            if (sl.n >= L3_TERMS) return false;
            const double pw = ts.p[i], q = ts.q[i];
            if (q != 0.0 && ts.r[i] != -1.0) return false;                 // exp(q / E) only
            const double fl = std::floor(pw + 1e-13), fr = pw - fl;        // E^p = E^m exp(fr ln E)
            const int mpow = (int)fl;
            if (mpow < -8 || mpow > 8) return false;
            int iP = 0, iQ = 0;
            if (fr > 1e-13 && !(iP = atom(pl.pf, pl.nP, fr))) return false;
            if (q != 0.0 && !(iQ = atom(pl.qv, pl.nQ, q))) return false;
            sl.code[sl.n] = iP | (iQ << 2) | ((mpow + 8) << 4);
            sl.c[sl.n] = ts.c[i];
            sl.p[sl.n] = pw;
            sl.q[sl.n] = q;
            ++sl.n;
        }
        return true;
    };
    auto all_slots = [&]() {
        for (int j = 0; j < m.n_reactions; ++j)
            if (!slot(m.k[j], pl.k[j])) return false;
        for (int s = 0; s < NS; ++s) {
            // (only what the rows evaluate: no flux, no coefficient)
            if (m.eq_type[s] == FEDM_EQ_REACTION) continue;
            if (!slot(m.D[s], pl.D[s])) return false;
            if (m.eq_type[s] == FEDM_EQ_DRIFT_DIFFUSION_REACTION && !m.has_drift_w[s] && !slot(m.mu[s], pl.mu[s])) return false;
        }
        return true;
    };
    if (!all_slots()) return false;
    // Canonical form, so that the STRUCTURE of a plan (what a signature states) does not depend on the order in which
    // a deck or a script happens to list terms: the atoms in ascending order, then every slot's terms by code.
    std::sort(pl.pf, pl.pf + pl.nP);
    std::sort(pl.qv, pl.qv + pl.nQ);
    if (!all_slots()) return false;      // (every atom is in the lists now: the codes take their sorted positions)
    auto sort_slot = [](Lean3Slot &sl) {
        for (int a = 0; a < sl.n; ++a)
            for (int b = a + 1; b < sl.n; ++b)
                if (sl.code[b] < sl.code[a]) {
                    std::swap(sl.code[a], sl.code[b]);
                    std::swap(sl.c[a], sl.c[b]);
                    std::swap(sl.p[a], sl.p[b]);
                    std::swap(sl.q[a], sl.q[b]);
                }
    };
    for (int j = 0; j < NR; ++j) sort_slot(pl.k[j]);
    for (int s = 0; s < NS; ++s) {
        sort_slot(pl.D[s]);
        sort_slot(pl.mu[s]);
    }
    return true;
}

// the slots in a signature's numbering
template <int NS, int NR>
static const Lean3Slot &lean3_slot(const Lean3Plan<NS, NR> &pl, int slot) {
    return slot < NR ? pl.k[slot] : slot < NR + NS ? pl.D[slot - NR] : pl.mu[slot - NR - NS];
}

// Does the plan have exactly the structure the signature states?
template <class SIG, int NS, int NR>
static bool lean3_sig_matches(const Lean3Plan<NS, NR> &pl) {
    if (!SIG::fixed) return true;
    if (SIG::NS != NS || SIG::NR != NR) return false;
    if (pl.nreac != SIG::nreac() || pl.nP != SIG::nP() || pl.nQ != SIG::nQ()) return false;
    for (int s = 0; s < NS; ++s)
        if (pl.eq_type[s] != SIG::eq_type(s) || (pl.has_drift_w[s] != 0) != (SIG::has_drift_w(s) != 0)) return false;
    for (int j = 0; j < pl.nreac; ++j)
        for (int s = 0; s < NS; ++s)
            if (pl.power[j][s] != SIG::power(j, s) || pl.net[j][s] != SIG::net(j, s)) return false;
    for (int k = 0; k < NR + 2 * NS; ++k) {
        const Lean3Slot &sl = lean3_slot(pl, k);
        if (sl.n != SIG::slot_n(k)) return false;
        for (int t = 0; t < sl.n; ++t)
            if (sl.code[t] != SIG::slot_code(k, t)) return false;
    }
    return true;
}

// FEDM_LEAN3_PRINT_SIG=1: the structure of this context's plan, as the signature tables state it
template <int NS, int NR>
static void lean3_print_signature(const Lean3Plan<NS, NR> &pl, bool precompiled) {
    std::fprintf(stderr, "[fedm] one-pass assembly: model structure %s\n  eq_type {", precompiled ? "precompiled (Lean3SigBenchmark)" : "taken at run time");
    for (int s = 0; s < NS; ++s) std::fprintf(stderr, "%d%s", pl.eq_type[s], s + 1 < NS ? ", " : "} has_drift_w {");
    for (int s = 0; s < NS; ++s) std::fprintf(stderr, "%d%s", pl.has_drift_w[s], s + 1 < NS ? ", " : "}");
    std::fprintf(stderr, " nreac %d nP %d nQ %d\n", pl.nreac, pl.nP, pl.nQ);
    for (int j = 0; j < pl.nreac; ++j) {
        std::fprintf(stderr, "  reaction %d: power {", j);
        for (int s = 0; s < NS; ++s) std::fprintf(stderr, "%d%s", pl.power[j][s], s + 1 < NS ? ", " : "} net {");
        for (int s = 0; s < NS; ++s) std::fprintf(stderr, "%d%s", pl.net[j][s], s + 1 < NS ? ", " : "}\n");
    }
    for (int k = 0; k < NR + 2 * NS; ++k) {
        const Lean3Slot &sl = lean3_slot(pl, k);
        std::fprintf(stderr, "  slot %d: n %d, codes (power atom | exp atom << 2 | (m + 8) << 4):", k, sl.n);
        for (int t = 0; t < sl.n; ++t)
            std::fprintf(stderr, " %d | %d << 2 | %d << 4", sl.code[t] & 3, (sl.code[t] >> 2) & 3, (sl.code[t] >> 4) & 31);
        std::fprintf(stderr, "\n");
    }
}

// LDS of one workgroup: live planes of its block columns + residual + staged vertex data
static size_t lean3_lds_bytes(int neq, int ns, int width, int verts, int n_live_planes, bool jacobian) {
    const size_t acc = jacobian ? (size_t)width * n_live_planes * SLICE : 0;
    return sizeof(double) * (acc + SLICE * neq + 2 * (size_t)verts + (size_t)(neq + 2 * ns) * verts);
}

// Two classes of patches.  Dynamic LDS is one number per launch, and a launch sized by the widest slice and the
// patch with the most staged vertices (9 block columns and 141 vertices on the refined bench mesh: 44 KB, three
// workgroups per CU) wastes it on the many that need less (4 722 of 5 333 slices there have 7 block columns:
// 35 KB, four per CU).  The patches that fit a quarter of the CU's LDS with the planes kept go in one launch,
// the rest in a second one sized for them -- when the split is worth a second launch.
struct Lean3Classes {
    int *d_list[2] = {nullptr, nullptr};
    int n[2] = {0, 0}, width[2] = {0, 0}, verts[2] = {0, 0};
    bool split = false;
};

static Lean3Classes *lean3_classes(Ctx &c, int n_live_planes) {
    if (c.lean3_classes) return static_cast<Lean3Classes *>(c.lean3_classes);
    Lean3Classes *k = new Lean3Classes();
    c.lean3_classes = k;
    // (measured on the refined bench mesh, tools/kernel_ab.py: 79-81 us with the split against 74-75 us for one
    // launch sized for everything -- the second launch's start-up and the first one's tail cost more than the
    // fourth workgroup per CU gains: opt-in, FEDM_LEAN3_CLASSES=1)
    static const bool off = [] {
        const char *e = std::getenv("FEDM_LEAN3_CLASSES");
        return !(e && e[0] == '1');
    }();
    const Pattern &pat = c.pat;
    const size_t budget = 160 * 1024 / 4;
    std::vector<int> lists[2];
    for (int S = 0; S < pat.n_slices; ++S) {
        const int w = pat.slice_boff[S + 1] - pat.slice_boff[S];
        const int v = SLICE + pat.patch_halo_ptr[S + 1] - pat.patch_halo_ptr[S];
        const int cls = lean3_lds_bytes(c.neq, c.ns, w, v, n_live_planes, true) <= budget ? 0 : 1;
        lists[cls].push_back(S);
        k->width[cls] = std::max(k->width[cls], w);
        k->verts[cls] = std::max(k->verts[cls], v);
    }
    // worth it: the launch sized for everything does not fit four workgroups a CU, most patches do
    k->split = !off && !lists[1].empty() && lists[0].size() >= 3 * lists[1].size();
    if (k->split) {
        for (int cls = 0; cls < 2; ++cls) {
            k->n[cls] = (int)lists[cls].size();
            if (hipMalloc((void **)&k->d_list[cls], sizeof(int) * lists[cls].size()) != hipSuccess ||
                hipMemcpy(k->d_list[cls], lists[cls].data(), sizeof(int) * lists[cls].size(), hipMemcpyHostToDevice) != hipSuccess) {
                hipGetLastError();
                k->split = false;
                break;
            }
        }
    }
    return k;
}

void lean3_release(Ctx &c) {
    if (c.d_lean3_plan) hipFree(c.d_lean3_plan);
    c.d_lean3_plan = nullptr;
    if (!c.lean3_classes) return;
    Lean3Classes *k = static_cast<Lean3Classes *>(c.lean3_classes);
    for (int cls = 0; cls < 2; ++cls)
        if (k->d_list[cls]) hipFree(k->d_list[cls]);
    delete k;
    c.lean3_classes = nullptr;
}

// In-run timing (fedm_profile): when this launch is the whole assembly, the pair of events prof_begin has reserved
// is handed to the dispatch itself (hipExtLaunchKernel: the events then carry the kernel's own begin and end, what
// rocprofv3 reports) instead of bracketing it with two hipEventRecord, whose marker packets add 3-5 us of
// command-processor time to an 80 us kernel.  FEDM_PROF_EXT=0: bracketing events.
template <class K, class... Args>
static void lean3_dispatch(Ctx &c, bool whole, K kernel, int grid, int threads, size_t lds, Args... args) {
    static const bool ext = [] {
        const char *e = std::getenv("FEDM_PROF_EXT");
        return !(e && e[0] == '0');
    }();
    Prof &pf = c.prof;
    if (ext && whole && pf.on && pf.recording && !c.capturing && pf.used + 2 <= (int)pf.ev.size()) {
        hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(threads), (uint32_t)lds, c.stream, pf.ev[pf.used], pf.ev[pf.used + 1], 0,
                              args...);
        pf.used += 2;            // the record is complete: prof_end has nothing to add
        pf.recording = false;
        return;
    }
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(threads), lds, c.stream, args...);
}

// the compiled model in device memory, uploaded at first use (fedm_ctx_destroy frees it: lean3_release), and the
// signature its structure selects (Ctx::lean3_sig: 0 = structure at run time, 1 = Lean3SigBenchmark)
template <int NS, int NR>
static bool lean3_ensure_plan(Ctx &c) {
    if (c.d_lean3_plan) return true;
    Lean3Plan<NS, NR> host_plan;
    if (!lean3_build_plan<NS, NR>(c.model, host_plan)) return false;
    if (hipMalloc(&c.d_lean3_plan, sizeof(host_plan)) != hipSuccess ||
        hipMemcpy(c.d_lean3_plan, &host_plan, sizeof(host_plan), hipMemcpyHostToDevice) != hipSuccess) {
        hipGetLastError();
        c.d_lean3_plan = nullptr;
        return false;
    }
    const char *e = std::getenv("FEDM_LEAN3_SIG");
    c.lean3_sig = (!(e && e[0] == '0') && lean3_sig_matches<Lean3SigBenchmark>(host_plan)) ? 1 : 0;
    const char *pr = std::getenv("FEDM_LEAN3_PRINT_SIG");
    if (pr && pr[0] == '1') lean3_print_signature(host_plan, c.lean3_sig == 1);
    return true;
}

template <int NS, int NR, uint32_t CMASK, class SIG>
static bool lean3_launch_one(Ctx &c, bool jacobian, const int *list, int n, int width, int verts, bool whole = true) {
    using PL = LivePlanes<NS, CMASK>;
    if (n <= 0) return true;
    if (!lean3_ensure_plan<NS, NR>(c)) return false;
    const Lean3Plan<NS, NR> *plan = static_cast<const Lean3Plan<NS, NR> *>(c.d_lean3_plan);
    Lean3Params p;
    p.nv = c.nv;
    p.boff = c.d_slice_boff;
    p.cell_ptr = c.d_patch_cell_ptr;
    p.pcells = c.d_patch_cells;
    p.halo_ptr = c.d_patch_halo_ptr;
    p.halo = c.d_patch_halo;
    p.coords = c.d_coords;
    p.u = c.d_u;
    p.uold = c.d_uold;
    p.uold1 = c.d_uold1;
    p.sc = step_coef(c.dt, c.dt_old);
    p.val = c.d_val;
    p.F = c.d_F;
    p.acc_doubles = jacobian ? width * PL::N * SLICE : 0;
    p.max_verts = verts;
    p.xcd = c.xcd_remap ? 1 : 0;
    p.patch_list = list;
    p.s16 = nullptr;
    p.val32 = nullptr;
    p.dinv = nullptr;
    p.diag_slot = nullptr;
    p.planes_upper = 0;
    p.planes_zs = 0u;
    const size_t lds = lean3_lds_bytes(c.neq, c.ns, width, verts, PL::N, jacobian);
    if (lds > 160 * 1024) return false;
    constexpr int T = 192;
    p.n_patches = n;
    // one workgroup per patch (default) or the persistent, software-pipelined kernels (FEDM_LEAN3_PERSISTENT=1:
    // measured equal on the refined mesh, 79.4 against 79.5 us in the bench, and slower on the tensor-product mesh,
    // 88.8 against 66.5 us, whose 5 203 patches leave 1 024 resident workgroups with six or five patches each)
    // (read per launch, not once per process: the tests compare the two forms in one process)
    const char *env_persistent = std::getenv("FEDM_LEAN3_PERSISTENT"), *env_wgs = std::getenv("FEDM_LEAN3_WGS_PER_CU");
    const bool persistent = env_persistent && env_persistent[0] == '1';
    const int wgs_per_cu_env = env_wgs ? std::atoi(env_wgs) : 0;
    if (persistent && verts <= T) {
        // as many workgroups as the chip holds at once (a multiple of 8: the XCDs' shares), each taking its patches
        // one after the other
        static int n_cu = 0;
        if (n_cu == 0 && (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, c.device) != hipSuccess || n_cu <= 0)) {
            hipGetLastError();
            n_cu = 256;
        }
        auto grid_for = [&](const void *kernel, size_t &granted) {
            if (lds > 64 * 1024 && lds > granted) {
                hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                granted = lds;
            }
            int per_cu = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, T, lds) != hipSuccess || per_cu <= 0) {
                hipGetLastError();
                per_cu = 1;
            }
            if (wgs_per_cu_env > 0) per_cu = wgs_per_cu_env;
            int g = std::min(n, n_cu * per_cu);
            if (p.xcd) g = std::max(8, g / 8 * 8);
            return g;
        };
        if (jacobian) {
            static size_t granted = 0;
            const int g = grid_for(reinterpret_cast<const void *>(&assemble_lean3p_kernel<NS, NR, T, CMASK, SIG>), granted);
            lean3_dispatch(c, whole, assemble_lean3p_kernel<NS, NR, T, CMASK, SIG>, g, T, lds, plan, p.val, p.F, p);
        } else {
            static size_t granted = 0;
            const int g = grid_for(reinterpret_cast<const void *>(&residual_lean3p_kernel<NS, NR, T, SIG>), granted);
            lean3_dispatch(c, whole, residual_lean3p_kernel<NS, NR, T, SIG>, g, T, lds, plan, p.val, p.F, p);
        }
        return true;
    }
    if (jacobian) {
        static size_t granted = 0;   // per instantiation: beyond the 64 KiB default the dynamic LDS is opt-in
        if (lds > 64 * 1024 && lds > granted) {
            hipFuncSetAttribute(reinterpret_cast<const void *>(&assemble_lean3_kernel<NS, NR, T, CMASK, SIG>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            granted = lds;
        }
        // the field split's planes from the accumulators (Ctx::planes_fused): the whole mesh in this one launch, the
        // planes' arrays there (from the second assembly on), kept species planes structurally zero
        const bool fuse = whole && !list && c.planes_fuse_ok && c.d_s16 && c.d_val32 && c.d_dinv && c.amg && c.poisson &&
                          !c.comm && ((CMASK & ~(1u << 8)) & ~c.zero_plane_mask) == 0u;
        if (fuse) {
            p.s16 = c.d_s16;
            p.val32 = c.d_val32;
            p.dinv = c.d_dinv;
            p.diag_slot = c.d_diag_slot;
            p.planes_upper = fieldsplit_upper(c) ? 1 : 0;
            p.planes_zs = fieldsplit_zero_species_planes(c);
        }
        lean3_dispatch(c, whole, assemble_lean3_kernel<NS, NR, T, CMASK, SIG>, n, T, lds, plan, p);
        c.planes_fused = fuse;
    } else {
        lean3_dispatch(c, whole, residual_lean3_kernel<NS, NR, T, SIG>, n, T, lds, plan, p);
    }
    return true;
}

template <int NS, int NR, uint32_t CMASK, class SIG>
static bool lean3_launch_sig(Ctx &c, bool jacobian, const int *list, int n) {
    using PL = LivePlanes<NS, CMASK>;
    // the split applies to the Jacobian assembly of a whole mesh with the planes kept (the steady state of a run);
    // the listed launches of the several-GPU path and the first full assembly take one launch sized for everything
    if (jacobian && !list && CMASK != 0u) {
        Lean3Classes *k = lean3_classes(c, PL::N);
        if (k->split) {
            // the few large ones first: their tail overlaps nothing, so it should be short
            return lean3_launch_one<NS, NR, CMASK, SIG>(c, true, k->d_list[1], k->n[1], k->width[1], k->verts[1], false) &&
                   lean3_launch_one<NS, NR, CMASK, SIG>(c, true, k->d_list[0], k->n[0], k->width[0], k->verts[0], false);
        }
    }
    return lean3_launch_one<NS, NR, CMASK, SIG>(c, jacobian, list, n, c.pat.max_patch_width, c.pat.max_patch_verts, list == nullptr);
}

// ... with the model's structure compiled in where a signature for it exists
template <int NS, int NR, uint32_t CMASK>
static bool lean3_launch(Ctx &c, bool jacobian, const int *list, int n) {
    if (!lean3_ensure_plan<NS, NR>(c)) return false;
    if (c.lean3_sig == 1) return lean3_launch_sig<NS, NR, CMASK, Lean3SigBenchmark>(c, jacobian, list, n);
    return lean3_launch_sig<NS, NR, CMASK, Lean3SigRuntime>(c, jacobian, list, n);
}

// The models this generation is instantiated for: two species and one reaction (the streamer family) with the
// planes the model keeps -- potential-potential alone, or together with a structurally zero species plane.
// Everything else stays on the second generation (kernels.hip).
bool lean3_applies(const Ctx &c) {
    // (FEDM_ASSEMBLY_LEAN below 3 keeps the earlier generations: Ctx::assembly_lean, read when the context is created)
    if (c.assembly_lean < 3 || c.ns != 2 || !c.poisson || c.model.n_reactions > 1 || c.pat.max_patch_cells > 2 * 192) return false;
    Lean3Plan<2, 1> plan;
    return lean3_build_plan<2, 1>(c.model, plan);
}

// Which precompiled structure signature this context's model selects (0: none -- the structure is read at run time)
int lean3_signature(const Ctx &c) {
    Lean3Plan<2, 1> plan;
    if (c.ns != 2 || !lean3_build_plan<2, 1>(c.model, plan)) return 0;
    const char *e = std::getenv("FEDM_LEAN3_SIG");
    return (!(e && e[0] == '0') && lean3_sig_matches<Lean3SigBenchmark>(plan)) ? 1 : 0;
}

// cmask: the planes that are kept for this launch (0 on a context's first full assembly: everything is written).
// false: the launch does not fit (LDS beyond the device's limit: the caller takes the second generation).
bool launch_assemble_lean3(Ctx &c, bool jacobian, const int *list, int n, uint32_t cmask) {
    if (n <= 0) return true;
    if (!jacobian) return lean3_launch<2, 1, 0u>(c, false, list, n);
    constexpr uint32_t PHIPHI = 1u << 8;
    switch (cmask) {
        case 0u: return lean3_launch<2, 1, 0u>(c, true, list, n);
        case PHIPHI: return lean3_launch<2, 1, PHIPHI>(c, true, list, n);
        case PHIPHI | (1u << 3): return lean3_launch<2, 1, PHIPHI | (1u << 3)>(c, true, list, n);   // + d(row 1)/d(u_0)
        case PHIPHI | (1u << 1): return lean3_launch<2, 1, PHIPHI | (1u << 1)>(c, true, list, n);   // + d(row 0)/d(u_1)
        // any other set of kept planes: the instantiation that keeps the potential-potential plane alone (the
        // other kept planes are recomputed and rewritten with the values they already hold), or none
        default: return (cmask & PHIPHI) ? lean3_launch<2, 1, PHIPHI>(c, true, list, n) : lean3_launch<2, 1, 0u>(c, true, list, n);
    }
}

}  // namespace fedm
