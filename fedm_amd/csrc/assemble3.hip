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

#include <cstdlib>

#include "element_lean.hpp"
#include "fedm_internal.hpp"

namespace fedm {

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

struct Lean3Params {
    const fedm_model_desc *md;
    int nv;
    const int *boff, *cell_ptr;
    const PatchCell *pcells;
    const int *halo_ptr, *halo;
    const double *coords, *u, *uold, *uold1;
    StepCoef sc;
    double *val, *F;
    int acc_doubles, max_verts, xcd;
    const int *patch_list;
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
};

// Species row S (fedm/functions.py:350-368 with Flux :219-237 and the source of :777-843).
template <int NS, int NR, uint32_t CMASK, bool JAC, int S>
__device__ __forceinline__ void lean3_species_row(const fedm_model_desc *__restrict__ md, const Lean3Cell<NS, NR> &c,
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
    const int eq = md->eq_type[S];
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
        termsum_eval(md->D[S], c.Em, c.invEm, c.lnE, Dv, Dd);
        vel[0] = -Dv * gradu[0];
        vel[1] = -Dv * gradu[1];
        if (eq == FEDM_EQ_DRIFT_DIFFUSION_REACTION) {
            if (md->has_drift_w[S]) {
                vel[0] += md->drift_w[S][0];
                vel[1] += md->drift_w[S][1];
            } else {
                termsum_eval(md->mu[S], c.Em, c.invEm, c.lnE, muv, mud);
                vel[0] += Z * muv * c.E[0];
                vel[1] += Z * muv * c.E[1];
                fdrift = true;
            }
        }
    }
    // weighted point values: H (residual integrand), N (density), SP (d source / d|E|), Xg[i] (d integrand / d u_i)
    const double usum6 = (Us[0] + Us[1] + Us[2]) * (1.0 / 6.0), hsum6 = (Hs[0] + Hs[1] + Hs[2]) * (1.0 / 6.0);
    const int nreac = md->n_reactions;
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
            const double nu = (double)md->net[j][S];
            if (nu == 0.0) continue;
            double prod = 1.0;
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const int Pw = md->power[j][i];
                for (int e = 0; e < Pw; ++e) prod *= c.nq[i][q];
            }
            const double nk = nu * c.kv[j] * prod;
            h -= nk;
            sp += nu * c.kd[j] * prod;
            if constexpr (JAC) {
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    const int Pw = md->power[j][i];
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
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int lane = c.lv[a];
        if (lane >= SLICE) continue;   // row vertex owned by another patch
        unsafeAtomicAdd(&Fl[lane * NEQ + S], m1h[a] - velG[a] * m0n);
        if constexpr (JAC) {
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int k = sym6(a, b);
                double *d = reinterpret_cast<double *>(lds_base + dst[a * 3 + b]);
#pragma unroll
                for (int i = 0; i < NS; ++i)
                    if (PL::live(S, i))
                        unsafeAtomicAdd(&d[PL::index(S, i) * SLICE],
                                        m2[i][k] + ((i == S) ? DGm * c.gg[k] - velG[a] * m1n[b] : 0.0));
                if constexpr (PL::live(S, IPHI))
                    unsafeAtomicAdd(&d[PL::index(S, IPHI) * SLICE], Kd * c.gg[k] + c.Qa[b] * c.invEm * T[a]);
            }
        }
    }
}

// Poisson row (fedm/functions.py:401): 2 pi r (grad Phi . grad v - sum_i Z_i e n_i / eps0 v)
template <int NS, int NR, uint32_t CMASK, bool JAC>
__device__ __forceinline__ void lean3_poisson_row(const fedm_model_desc *__restrict__ md, const Lean3Cell<NS, NR> &c,
                                                  const uint32_t (&dst)[9], double *__restrict__ Fl,
                                                  char *__restrict__ lds_base) {
    constexpr int NEQ = NS + 1, IPHI = NS;
    using PL = LivePlanes<NS, CMASK>;
    const double coe = md->charge_over_eps;
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
        unsafeAtomicAdd(&Fl[lane * NEQ + IPHI], m1h[a] - c.Qa[a] * c.m01);
        if constexpr (JAC) {
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int k = sym6(a, b);
                double *d = reinterpret_cast<double *>(lds_base + dst[a * 3 + b]);
#pragma unroll
                for (int i = 0; i < NS; ++i)
                    if (PL::live(IPHI, i)) unsafeAtomicAdd(&d[PL::index(IPHI, i) * SLICE], m2[i][k]);
                if constexpr (PL::live(IPHI, IPHI)) unsafeAtomicAdd(&d[PL::index(IPHI, IPHI) * SLICE], c.gg[k] * c.m01);
            }
        }
    }
}

template <int NS, int NR, uint32_t CMASK, bool JAC, int S>
__device__ __forceinline__ void lean3_species_rows(const fedm_model_desc *__restrict__ md, const Lean3Cell<NS, NR> &c,
                                                   const double *__restrict__ Ul, const double *__restrict__ Hl,
                                                   const StepCoef sc, const uint32_t (&dst)[9], double *__restrict__ Fl,
                                                   char *__restrict__ lds_base) {
    if constexpr (S < NS) {
        lean3_species_row<NS, NR, CMASK, JAC, S>(md, c, Ul, Hl, sc, dst, Fl, lds_base);
        __builtin_amdgcn_sched_barrier(0);   // the rows one after the other: nothing of row S + 1 lives beside row S
        lean3_species_rows<NS, NR, CMASK, JAC, S + 1>(md, c, Ul, Hl, sc, dst, Fl, lds_base);
    }
}

template <int NS, int NR, int THREADS, uint32_t CMASK, bool JAC>
__device__ __forceinline__ void assemble_lean3_body(const Lean3Params &p) {
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
    const int blk = p.xcd ? lean3_xcd_contiguous(blockIdx.x, gridDim.x) : blockIdx.x;
    const int S = p.patch_list ? p.patch_list[blk] : blk;
    const int b0 = p.boff[S], width = p.boff[S + 1] - b0;
    const int c0 = p.cell_ptr[S], n_cells = p.cell_ptr[S + 1] - c0;
    const bool active = (int)threadIdx.x < n_cells;   // one cell per thread (n_cells <= THREADS)
    PatchCell pc = {};
    if (active) pc = p.pcells[c0 + threadIdx.x];
    if constexpr (JAC) {
        double2 *acc2 = reinterpret_cast<double2 *>(acc);
        const int n2 = width * NPL * (SLICE / 2);
        for (int k = threadIdx.x; k < n2; k += THREADS) acc2[k] = make_double2(0.0, 0.0);
    }
    for (int k = threadIdx.x; k < SLICE * NEQ; k += THREADS) Fl[k] = 0.0;
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
    __syncthreads();
    if (active) {
        const fedm_model_desc *__restrict__ md = p.md;
        Lean3Cell<NS, NR> c;
        double x[3][2];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            c.lv[a] = pc.lv[a];
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
            const int nreac = md->n_reactions;
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                c.kv[j] = c.kd[j] = 0.0;
                if (j < nreac) termsum_eval(md->k[j], c.Em, c.invEm, c.lnE, c.kv[j], c.kd[j]);
            }
        }
        // byte offsets of the nine (a, b) accumulators of this cell (row vertex a owned: lane < 64)
        uint32_t dst[9];
        char *lds_base = reinterpret_cast<char *>(acc);
        if constexpr (JAC) {
#pragma unroll
            for (int e = 0; e < 9; ++e)
                dst[e] = ((uint32_t)pc.j[e] * (NPL * SLICE) + (uint32_t)pc.lv[e / 3]) * (uint32_t)sizeof(double);
        }
        __builtin_amdgcn_sched_barrier(0);
        lean3_species_rows<NS, NR, CMASK, JAC, 0>(md, c, Ul, Hl, p.sc, dst, Fl, lds_base);
        lean3_poisson_row<NS, NR, CMASK, JAC>(md, c, dst, Fl, lds_base);
    }
    __syncthreads();
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
            reinterpret_cast<double2 *>(p.val + ((size_t)(b0 + bc) * NEQ2 + rc) * SLICE)[piece] = src[k * (SLICE / 2)];
        }
    }
    double *fdst = p.F + (size_t)S * SLICE * NEQ;
    for (int k = threadIdx.x; k < SLICE * NEQ; k += THREADS) fdst[k] = Fl[k];
}

#ifndef FEDM_LEAN3_WAVES
#define FEDM_LEAN3_WAVES 3
#endif
template <int NS, int NR, int THREADS, uint32_t CMASK>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(FEDM_LEAN3_WAVES, FEDM_LEAN3_WAVES))) void assemble_lean3_kernel(
    const Lean3Params p) {
    assemble_lean3_body<NS, NR, THREADS, CMASK, true>(p);
}

template <int NS, int NR, int THREADS>
__global__ __launch_bounds__(THREADS) void residual_lean3_kernel(const Lean3Params p) {
    assemble_lean3_body<NS, NR, THREADS, 0u, false>(p);
}

// LDS of one workgroup: live planes of the widest slice + residual + staged vertex data
static size_t lean3_lds_bytes(const Ctx &c, int n_live_planes, bool jacobian) {
    const int neq = c.neq, mv = c.pat.max_patch_verts;
    const size_t acc = jacobian ? (size_t)c.pat.max_patch_width * n_live_planes * SLICE : 0;
    return sizeof(double) * (acc + SLICE * neq + 2 * (size_t)mv + (size_t)(neq + 2 * c.ns) * mv);
}

template <int NS, int NR, uint32_t CMASK>
static void lean3_launch(Ctx &c, bool jacobian, const int *list, int n) {
    using PL = LivePlanes<NS, CMASK>;
    Lean3Params p;
    p.md = c.d_model;
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
    p.acc_doubles = jacobian ? c.pat.max_patch_width * PL::N * SLICE : 0;
    p.max_verts = c.pat.max_patch_verts;
    p.xcd = c.xcd_remap ? 1 : 0;
    p.patch_list = list;
    const size_t lds = lean3_lds_bytes(c, PL::N, jacobian);
    const bool wide = c.pat.max_patch_cells > 192;
    if (jacobian) {
        if (wide) hipLaunchKernelGGL((assemble_lean3_kernel<NS, NR, 256, CMASK>), dim3(n), dim3(256), lds, c.stream, p);
        else hipLaunchKernelGGL((assemble_lean3_kernel<NS, NR, 192, CMASK>), dim3(n), dim3(192), lds, c.stream, p);
    } else {
        if (wide) hipLaunchKernelGGL((residual_lean3_kernel<NS, NR, 256>), dim3(n), dim3(256), lds, c.stream, p);
        else hipLaunchKernelGGL((residual_lean3_kernel<NS, NR, 192>), dim3(n), dim3(192), lds, c.stream, p);
    }
}

// The models this generation is instantiated for: two species and one reaction (the streamer family) with the
// planes the model keeps -- potential-potential alone, or together with a structurally zero species plane.
// Everything else stays on the second generation (kernels.hip).
bool lean3_applies(const Ctx &c) {
    static const bool off = [] {
        const char *e = std::getenv("FEDM_ASSEMBLY_LEAN");
        return e && std::atoi(e) < 3;
    }();
    if (off || c.ns != 2 || !c.poisson || c.model.n_reactions > 1 || c.pat.max_patch_cells > 256) return false;
    return true;
}

// cmask: the planes that are kept for this launch (0 on a context's first full assembly: everything is written).
// false: no instantiation for this mask (the caller takes the second generation).
bool launch_assemble_lean3(Ctx &c, bool jacobian, const int *list, int n, uint32_t cmask) {
    if (n <= 0) return true;
    if (!jacobian) {
        lean3_launch<2, 1, 0u>(c, false, list, n);
        return true;
    }
    constexpr uint32_t PHIPHI = 1u << 8;
    switch (cmask) {
        case 0u: lean3_launch<2, 1, 0u>(c, true, list, n); return true;
        case PHIPHI: lean3_launch<2, 1, PHIPHI>(c, true, list, n); return true;
        case PHIPHI | (1u << 3): lean3_launch<2, 1, PHIPHI | (1u << 3)>(c, true, list, n); return true;   // + d(row 1)/d(u_0)
        case PHIPHI | (1u << 1): lean3_launch<2, 1, PHIPHI | (1u << 1)>(c, true, list, n); return true;   // + d(row 0)/d(u_1)
        default: return false;
    }
}

}  // namespace fedm
