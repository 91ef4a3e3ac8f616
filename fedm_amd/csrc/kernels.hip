// HIP kernels (gfx950) of the FEDM hot path: coloured element assembly into the sliced
// block-ELL Jacobian, Dirichlet rows, block-Jacobi inverse, SpMV and the vector kernels
// that GMRES / Newton need.  All of it is fp64 and HBM-bound; no MFMA.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>

#include "comm.hpp"
#include "element.hpp"
#include "element_lean.hpp"
#include "fedm_internal.hpp"

namespace fedm {

// The ghost entries of the state are refreshed lazily: the Newton loop only marks them stale
// (Ctx::halo_pending) and the next assembly either overlaps the exchange with its interior
// patches or, on every other path, performs it here first.
static void flush_pending_halo(Ctx &c) {
    if (!c.halo_pending) return;
    comm_halo(c, c.d_u);
    c.halo_pending = false;
}

// =============================================================================================
// Assembly, variant 0: one thread per cell, one launch per colour (cells of a colour share
// no vertex, so the read-modify-write of matrix blocks and residual entries is conflict-free
// and the summation order is fixed -> bitwise reproducible).
// Problem.F / Problem.J, fedm/functions.py:188-202
// =============================================================================================
template <int NS, bool PO, int NR, int CACHE, bool LIN>
__global__ __launch_bounds__(256) void assemble_colour_kernel(
    const fedm_model_desc *__restrict__ md, const int *__restrict__ cell_list, int n_cells,
    const int *__restrict__ cells, const double *__restrict__ coords,
    const uint32_t *__restrict__ cell_slots, const double *__restrict__ u,
    const double *__restrict__ uold, const double *__restrict__ uold1, StepCoef sc,
    const double *ext0, const double *ext1, const double *ext2, const double *ext3,
    double *__restrict__ val, double *__restrict__ F, int jacobian, int mode) {
    constexpr int NEQ = NS + (PO ? 1 : 0);
    constexpr int NEQ2 = NEQ * NEQ;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_cells) return;
    const int c = cell_list[t];
    int v[3];
    double x[3][2], Uc[3][NEQ], Hc[3][NS];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        v[a] = cells[3 * c + a];
        x[a][0] = coords[2 * v[a]];
        x[a][1] = coords[2 * v[a] + 1];
#pragma unroll
        for (int s = 0; s < NEQ; ++s) Uc[a][s] = u[(size_t)v[a] * NEQ + s];
#pragma unroll
        for (int s = 0; s < NS; ++s)
            Hc[a][s] = sc.c_old * uold[(size_t)v[a] * NEQ + s] + sc.c_old1 * uold1[(size_t)v[a] * NEQ + s];
    }
    const double *extp[4] = {ext0, ext1, ext2, ext3};
    const double *ext[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s)
        ext[s] = (extp[s] && md->ext_nodes[s]) ? extp[s] + (size_t)c * md->ext_nodes[s] : nullptr;

    Element<NS, PO, NR, CACHE, LIN> el;
    el.setup(md, x, Uc, Hc, sc, mode);
    uint32_t slot[9];
    if (jacobian) {
#pragma unroll
        for (int k = 0; k < 9; ++k) slot[k] = cell_slots[(size_t)c * 9 + k];
    }
#pragma unroll
    for (int row = 0; row < NEQ; ++row) {
        if (mode == 1 && PO && row != NEQ - 1) continue;  // Poisson-only: species rows are identity
        el.row_moments(md, row, Uc, Hc, sc, ext);
            el.row_prepare(md, row);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            F[(size_t)v[a] * NEQ + row] += el.residual(row, a);
            if (!jacobian) continue;
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                double B[NEQ];
                el.block_row(md, row, a, b, B);
                const uint32_t sl = slot[a * 3 + b];
                double *dst = val + ((size_t)(sl >> 6) * NEQ2 + row * NEQ) * SLICE + (sl & 63);
#pragma unroll
                for (int i = 0; i < NEQ; ++i) dst[(size_t)i * SLICE] += B[i];
            }
        }
    }
}

template <int NS, bool PO, int NR, int CACHE, bool LIN>
static void assemble_colour_t(Ctx &c, bool jacobian, int mode) {
    constexpr int NEQ = NS + (PO ? 1 : 0);
    flush_pending_halo(c);
    hipMemsetAsync(c.d_F, 0, sizeof(double) * c.np, c.stream);
    if (jacobian)
        hipMemsetAsync(c.d_val, 0, sizeof(double) * (size_t)c.pat.total_bc * SLICE * NEQ * NEQ, c.stream);
    const StepCoef sc = step_coef(c.dt, c.dt_old);
    const int ncol = (int)c.pat.colour_ptr.size() - 1;
    for (int k = 0; k < ncol; ++k) {
        const int n = c.pat.colour_ptr[k + 1] - c.pat.colour_ptr[k];
        if (n == 0) continue;
        hipLaunchKernelGGL((assemble_colour_kernel<NS, PO, NR, CACHE, LIN>), dim3((n + 255) / 256), dim3(256), 0,
                           c.stream, c.d_model, c.d_colour_cells + c.pat.colour_ptr[k], n,
                           c.d_cells, c.d_coords, c.d_cell_slots, c.d_u, c.d_uold, c.d_uold1, sc,
                           c.d_ext[0], c.d_ext[1], c.d_ext[2], c.d_ext[3], c.d_val, c.d_F,
                           jacobian ? 1 : 0, mode);
    }
}

// =============================================================================================
// Assembly, variant 1 (default): LDS patches.  One workgroup owns one matrix slice (64 vertex
// rows).  It stages the patch's vertex data (owned + halo vertices: coordinates, u and the
// folded BDF history) in LDS, evaluates every cell that touches an owned vertex (cells on
// patch borders are evaluated by each patch they touch), accumulates the owned rows' blocks
// and residual entries in LDS with ds_add_f64, and finally streams the finished slice out
// with fully coalesced stores: every matrix value is written exactly once -- no
// read-modify-write in HBM and no zero-fill pass.
// =============================================================================================

#ifdef FEDM_PHASE_TIMING
__device__ unsigned long long g_phase[8];
#define FEDM_T(k) if (threadIdx.x == 0) { const unsigned long long now_ = wall_clock64(); atomicAdd(&g_phase[k], now_ - t_prev_); t_prev_ = now_; }
extern "C" void fedm_debug_phase(unsigned long long *out, int reset) {
    hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), sizeof(unsigned long long) * 8);
    if (reset) {
        unsigned long long z[8] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z));
    }
}
#else
#define FEDM_T(k)
#endif

// The kernel sits at 237-255 VGPRs; amdgpu_waves_per_eu pins it to two waves per SIMD (one
// wave per SIMD is 1.5x slower) should a compiler change push it over 256.
// THREADS: workgroup size = the patch's cell count rounded up (192 for Z-ordered meshes: two
// 3-wave workgroups per CU at 2 waves/SIMD keep 6 waves busy; 320 covers 1-D strips)
// JAC = false is the residual-only assembly (final Newton check): without the Jacobian code it
// needs about half the registers and no accumulators, so it is compiled as a kernel of its own
// that the compiler may run at a higher occupancy.
template <int NS, bool PO, int NR, int CACHE, int THREADS, bool JAC, bool LIN>
__device__ __forceinline__ void assemble_patch_body(
    const fedm_model_desc *__restrict__ md, int nv, const int *__restrict__ boff,
    const int *__restrict__ cell_ptr, const PatchCell *__restrict__ pcells,
    const int *__restrict__ halo_ptr, const int *__restrict__ halo,
    const double *__restrict__ coords, const double *__restrict__ u,
    const double *__restrict__ uold, const double *__restrict__ uold1, StepCoef sc,
    const double *ext0, const double *ext1, const double *ext2, const double *ext3,
    double *__restrict__ val, double *__restrict__ F, int mode, int acc_doubles,
    int max_verts) {
    constexpr bool jacobian = JAC;
    constexpr int NEQ = NS + (PO ? 1 : 0);
    constexpr int NEQ2 = NEQ * NEQ;
    extern __shared__ __align__(16) double lds[];
    double *acc = lds;                          // [width][NEQ2][64]
    double *Fl = acc + acc_doubles;             // [64][NEQ]
    double *vx = Fl + SLICE * NEQ;              // [max_verts][2]
    double *Ul = vx + 2 * max_verts;            // [max_verts][NEQ]
    double *Hl = Ul + NEQ * max_verts;          // [max_verts][NS]

#ifdef FEDM_PHASE_TIMING
    unsigned long long t_prev_ = wall_clock64();
#endif
    const int S = blockIdx.x;
    const int b0 = boff[S], width = boff[S + 1] - b0;
    const int n_acc = jacobian ? width * NEQ2 * SLICE : 0;  // a multiple of 64: 16-byte LDS / HBM accesses
    {
        double2 *acc2 = reinterpret_cast<double2 *>(acc);
        for (int k = threadIdx.x; k < n_acc / 2; k += blockDim.x) acc2[k] = make_double2(0.0, 0.0);
    }
    for (int k = threadIdx.x; k < SLICE * NEQ; k += blockDim.x) Fl[k] = 0.0;
    FEDM_T(0)
    const int h0 = halo_ptr[S], n_local = SLICE + halo_ptr[S + 1] - h0;
    for (int i = threadIdx.x; i < n_local; i += blockDim.x) {
        const int g = (i < SLICE) ? S * SLICE + i : halo[h0 + i - SLICE];
        if (g < nv) {
            vx[2 * i] = coords[2 * (size_t)g];
            vx[2 * i + 1] = coords[2 * (size_t)g + 1];
#pragma unroll
            for (int s = 0; s < NEQ; ++s) Ul[i * NEQ + s] = u[(size_t)g * NEQ + s];
#pragma unroll
            for (int s = 0; s < NS; ++s)
                Hl[i * NS + s] = sc.c_old * uold[(size_t)g * NEQ + s] + sc.c_old1 * uold1[(size_t)g * NEQ + s];
        }
    }
    FEDM_T(1)
    __syncthreads();
    FEDM_T(2)

    const int c0 = cell_ptr[S], n_cells = cell_ptr[S + 1] - c0;
    for (int i = threadIdx.x; i < n_cells; i += blockDim.x) {
        const PatchCell pc = pcells[c0 + i];
        double x[3][2], Uc[3][NEQ], Hc[3][NS];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int l = pc.lv[a];
            x[a][0] = vx[2 * l];
            x[a][1] = vx[2 * l + 1];
#pragma unroll
            for (int s = 0; s < NEQ; ++s) Uc[a][s] = Ul[l * NEQ + s];
#pragma unroll
            for (int s = 0; s < NS; ++s) Hc[a][s] = Hl[l * NS + s];
        }
        const double *extp[4] = {ext0, ext1, ext2, ext3};
        const double *ext[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s)
            ext[s] = (extp[s] && md->ext_nodes[s]) ? extp[s] + (size_t)pc.cell * md->ext_nodes[s] : nullptr;

        FEDM_T(3)
        Element<NS, PO, NR, CACHE, LIN> el;
        el.setup(md, x, Uc, Hc, sc, mode);
        FEDM_T(4)
#pragma unroll
        for (int row = 0; row < NEQ; ++row) {
            if (mode == 1 && PO && row != NEQ - 1) continue;
            el.row_moments(md, row, Uc, Hc, sc, ext);
            el.row_prepare(md, row);
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const int lane = pc.lv[a];
                if (lane >= SLICE) continue;  // row vertex owned by another patch
                unsafeAtomicAdd(&Fl[lane * NEQ + row], el.residual(row, a));
                if (!jacobian) continue;
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    double B[NEQ];
                    el.block_row(md, row, a, b, B);
                    double *dst = acc + ((size_t)pc.j[a * 3 + b] * NEQ2 + row * NEQ) * SLICE + lane;
#pragma unroll
                    for (int i = 0; i < NEQ; ++i) unsafeAtomicAdd(&dst[i * SLICE], B[i]);
                }
            }
        }
    }
    FEDM_T(5)
    __syncthreads();
    FEDM_T(6)
    {
        double2 *vdst2 = reinterpret_cast<double2 *>(val + (size_t)b0 * NEQ2 * SLICE);
        const double2 *acc2 = reinterpret_cast<const double2 *>(acc);
        for (int k = threadIdx.x; k < n_acc / 2; k += blockDim.x) vdst2[k] = acc2[k];
    }
    double *fdst = F + (size_t)S * SLICE * NEQ;
    for (int k = threadIdx.x; k < SLICE * NEQ; k += blockDim.x) fdst[k] = Fl[k];
    FEDM_T(7)
}

#define FEDM_PATCH_PARAMS                                                                          \
    const fedm_model_desc *__restrict__ md, int nv, const int *__restrict__ boff,                  \
        const int *__restrict__ cell_ptr, const PatchCell *__restrict__ pcells,                    \
        const int *__restrict__ halo_ptr, const int *__restrict__ halo,                            \
        const double *__restrict__ coords, const double *__restrict__ u,                           \
        const double *__restrict__ uold, const double *__restrict__ uold1, StepCoef sc,            \
        const double *ext0, const double *ext1, const double *ext2, const double *ext3,            \
        double *__restrict__ val, double *__restrict__ F, int mode, int acc_doubles, int max_verts
#define FEDM_PATCH_ARGS                                                                            \
    md, nv, boff, cell_ptr, pcells, halo_ptr, halo, coords, u, uold, uold1, sc, ext0, ext1, ext2,  \
        ext3, val, F, mode, acc_doubles, max_verts

template <int NS, bool PO, int NR, int CACHE, int THREADS, bool LIN>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void assemble_patch_kernel(
    FEDM_PATCH_PARAMS) {
    assemble_patch_body<NS, PO, NR, CACHE, THREADS, true, LIN>(FEDM_PATCH_ARGS);
}

template <int NS, bool PO, int NR, int CACHE, int THREADS, bool LIN>
__global__ __launch_bounds__(THREADS) void residual_patch_kernel(FEDM_PATCH_PARAMS) {
    assemble_patch_body<NS, PO, NR, CACHE, THREADS, false, LIN>(FEDM_PATCH_ARGS);
}

// workgroup barrier that orders LDS accesses only: global stores issued before it stay in flight
__device__ __forceinline__ void lds_only_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Workgroups b, b + 8, b + 16, ... are observed to share an XCD (round-robin dispatch; a speed
// heuristic, never relied upon for correctness): give every XCD a contiguous range of patches so
// that the halo vertices two neighbouring patches both stage are served by ONE L2.
__device__ __forceinline__ int xcd_contiguous(int b, int n) {
    const int per = n >> 3, full = per << 3;
    return b < full ? (b & 7) * per + (b >> 3) : b;
}

// Second generation of the row-phase kernel (element_lean.hpp, lean2_*): per-vertex exponentials,
// cell constants kept in an LDS column between the rows; JAC = false is the residual-only assembly.
template <int NS, int NR, int THREADS, bool JAC>
__device__ __forceinline__ void assemble_lean2_body(FEDM_PATCH_PARAMS, int xcd, const int *__restrict__ patch_list,
                                                    uint32_t cmask) {
    constexpr int NEQ = NS + 1, NEQ2 = NEQ * NEQ;
    constexpr int NST = LeanStash<NR>::N;
    extern __shared__ __align__(16) double lds[];
    double *acc = lds;                          // [width][NEQ][64]: one row of every block (JAC)
    double *Fl = acc + acc_doubles;             // [64][NEQ]
    double *vx = Fl + SLICE * NEQ;              // [max_verts][2]
    double *Ul = vx + 2 * max_verts;            // [max_verts][NEQ]
    double *Hl = Ul + NEQ * max_verts;          // [max_verts][NS]
    double *Al = Hl + NS * max_verts;           // [max_verts][NS]: exp(u / 6)
    double *cst = Al + NS * max_verts;          // [NST][THREADS]
#ifdef FEDM_PHASE_TIMING
    unsigned long long t_prev_ = wall_clock64();
#endif
    // patch_list: the launch covers those patches only (interior / boundary halves across GPUs)
    const int blk = xcd ? xcd_contiguous(blockIdx.x, gridDim.x) : blockIdx.x;
    const int S = patch_list ? patch_list[blk] : blk;
    const int b0 = boff[S], width = boff[S + 1] - b0;
    const int n_acc = JAC ? width * NEQ * SLICE : 0;
    const int c0 = cell_ptr[S], n_cells = cell_ptr[S + 1] - c0;
    const bool active = (int)threadIdx.x < n_cells;   // one cell per thread (n_cells <= THREADS)
    PatchCell pc_own = {};
    if (active) pc_own = pcells[c0 + threadIdx.x];
    if constexpr (JAC) {
        double2 *acc2 = reinterpret_cast<double2 *>(acc);
        for (int k = threadIdx.x; k < n_acc / 2; k += THREADS) acc2[k] = make_double2(0.0, 0.0);
    }
    for (int k = threadIdx.x; k < SLICE * NEQ; k += THREADS) Fl[k] = 0.0;
    const int h0 = halo_ptr[S], n_local = SLICE + halo_ptr[S + 1] - h0;
    for (int i = threadIdx.x; i < n_local; i += THREADS) {
        const int g = (i < SLICE) ? S * SLICE + i : halo[h0 + i - SLICE];
        if (g < nv) {
            vx[2 * i] = coords[2 * (size_t)g];
            vx[2 * i + 1] = coords[2 * (size_t)g + 1];
            double un[NEQ];
#pragma unroll
            for (int s = 0; s < NEQ; ++s) un[s] = u[(size_t)g * NEQ + s];
#pragma unroll
            for (int s = 0; s < NS; ++s)
                Hl[i * NS + s] = sc.c_old * uold[(size_t)g * NEQ + s] + sc.c_old1 * uold1[(size_t)g * NEQ + s];
#pragma unroll
            for (int s = 0; s < NEQ; ++s) Ul[i * NEQ + s] = un[s];
#pragma unroll
            for (int s = 0; s < NS; ++s) Al[i * NS + s] = exp(un[s] * (1.0 / 6.0));
        }
    }
    FEDM_T(0)   // zero + stage (issue) + vertex exponentials
    __syncthreads();
    FEDM_T(1)   // barrier: the staged loads arrive
    LeanCell lc = {0, 0, 0, 0};
    if (active) lc = lean2_prologue<NS, NR>(md, pc_own, vx, Ul, cst + threadIdx.x, THREADS);
    FEDM_T(2)   // prologue: cell record, field, rate coefficient
    if constexpr (!JAC) {
        // residual only: no accumulators, no row phases -- the rows share the cell's geometry
        if (active) {
            int lv[3];
            double G[3][2], W[3];
            lean2_geometry(md, lc, vx, lv, G, W, cst[LeanStash<NR>::IDET * THREADS + threadIdx.x]);
#pragma unroll
            for (int row = 0; row < NEQ; ++row)
                lean2_row_core<NS, NR, false>(md, row, lc, lv, G, W, Ul, Hl, Al, sc, acc, Fl, cst + threadIdx.x, THREADS);
        }
    }
#pragma unroll 1
    for (int row = 0; JAC && row < NEQ; ++row) {
        asm volatile("" : "+v"(lc.wl), "+v"(lc.wj0), "+v"(lc.wj1), "+v"(lc.wj2));  // nothing hoisted out of the row
        if (active) {
#ifndef FEDM_LEAN2_ROW_GENERIC
            // one body per equation row: the row index is a compile-time constant inside each (the
            // selects on it fold away: 161 -> 144 VGPRs, -3 %; -DFEDM_LEAN2_ROW_GENERIC: one shared body)
#define FEDM_ROW_CASE(R)                                                                                   \
    case R:                                                                                                \
        if constexpr (NEQ > R)                                                                             \
            lean2_row<NS, NR, JAC, R>(md, row, lc, vx, Ul, Hl, Al, sc, acc, Fl, cst + threadIdx.x, THREADS, cmask); \
        break;
            switch (row) {
                FEDM_ROW_CASE(0)
                FEDM_ROW_CASE(1)
                FEDM_ROW_CASE(2)
                FEDM_ROW_CASE(3)
            }
#undef FEDM_ROW_CASE
#else
            lean2_row<NS, NR, JAC>(md, row, lc, vx, Ul, Hl, Al, sc, acc, Fl, cst + threadIdx.x, THREADS, cmask);
#endif
        }
        FEDM_T(3)   // the row (wave 0's view)
        if constexpr (JAC) {
            __syncthreads();
            FEDM_T(4)   // barrier: the other waves finish the row
            constexpr int PER = NEQ * SLICE / 2;   // 16-byte pieces per block column
            // planes that never change (cmask) are neither read out nor zeroed: 32 pieces per plane
            const uint32_t rmask = cmask >> (row * NEQ);
            if constexpr (THREADS % PER == 0) {
                // a thread keeps its place within the block column and strides over the columns: all
                // its LDS reads are issued before the first store waits for one of them
                int tid = threadIdx.x;
                asm volatile("" : "+v"(tid));   // keeps the addresses below out of the row loop's live state
                const int rem = tid % PER;
                constexpr int STEP = THREADS / PER;
                double2 *srcs = reinterpret_cast<double2 *>(acc) + rem;
                if (!((rmask >> (rem / (SLICE / 2))) & 1u)) {
                    // two block columns per pass: both LDS reads are in flight before the first store
                    int bc = tid / PER;
                    for (; bc + STEP < width; bc += 2 * STEP) {
                        const double2 a = srcs[bc * PER], b = srcs[(bc + STEP) * PER];
                        reinterpret_cast<double2 *>(val + ((size_t)(b0 + bc) * NEQ2 + row * NEQ) * SLICE)[rem] = a;
                        reinterpret_cast<double2 *>(val + ((size_t)(b0 + bc + STEP) * NEQ2 + row * NEQ) * SLICE)[rem] = b;
                        srcs[bc * PER] = make_double2(0.0, 0.0);
                        srcs[(bc + STEP) * PER] = make_double2(0.0, 0.0);
                    }
                    if (bc < width) {
                        reinterpret_cast<double2 *>(val + ((size_t)(b0 + bc) * NEQ2 + row * NEQ) * SLICE)[rem] = srcs[bc * PER];
                        srcs[bc * PER] = make_double2(0.0, 0.0);
                    }
                }
            } else {
                for (int k = threadIdx.x; k < n_acc / 2; k += THREADS) {
                    const int bc = k / PER, rem = k - bc * PER;
                    if ((rmask >> (rem / (SLICE / 2))) & 1u) continue;
                    double2 *dst = reinterpret_cast<double2 *>(val + ((size_t)(b0 + bc) * NEQ2 + row * NEQ) * SLICE);
                    double2 *src = reinterpret_cast<double2 *>(acc) + k;
                    dst[rem] = *src;
                    *src = make_double2(0.0, 0.0);
                }
            }
            FEDM_T(5)   // stream-out + zeroing (issue)
            lds_only_barrier();   // accumulators zero again; the stores above stay in flight
            FEDM_T(6)
        }
    }
    if constexpr (!JAC) __syncthreads();
    double *fdst = F + (size_t)S * SLICE * NEQ;
    for (int k = threadIdx.x; k < SLICE * NEQ; k += THREADS) fdst[k] = Fl[k];
    FEDM_T(7)
}

#ifndef FEDM_LEAN2_WAVES
#define FEDM_LEAN2_WAVES 3
#endif
template <int NS, int NR, int THREADS>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(FEDM_LEAN2_WAVES, FEDM_LEAN2_WAVES))) void assemble_lean2_kernel(
    FEDM_PATCH_PARAMS, int xcd, const int *__restrict__ patch_list, uint32_t cmask) {
    assemble_lean2_body<NS, NR, THREADS, true>(FEDM_PATCH_ARGS, xcd, patch_list, cmask);
}

template <int NS, int NR, int THREADS>
__global__ __launch_bounds__(THREADS) void residual_lean2_kernel(FEDM_PATCH_PARAMS, int xcd,
                                                                 const int *__restrict__ patch_list, uint32_t cmask) {
    assemble_lean2_body<NS, NR, THREADS, false>(FEDM_PATCH_ARGS, xcd, patch_list, cmask);
}
#undef FEDM_PATCH_PARAMS
#undef FEDM_PATCH_ARGS

size_t patch_lds_bytes(const Ctx &c, bool jacobian) {
    const int neq = c.neq, mv = c.pat.max_patch_verts;
    const size_t acc = jacobian ? (size_t)c.pat.max_patch_width * neq * neq * SLICE : 0;
    return sizeof(double) * (acc + SLICE * neq + 2 * mv + (size_t)(neq + c.ns) * mv);
}

template <int NS, bool PO, int NR, int CACHE, bool LIN>
static void assemble_patch_t(Ctx &c, bool jacobian, int mode) {
    constexpr int NEQ = NS + (PO ? 1 : 0);
    const StepCoef sc = step_coef(c.dt, c.dt_old);
    const int acc_doubles = jacobian ? c.pat.max_patch_width * NEQ * NEQ * SLICE : 0;
#define FEDM_PATCH_LAUNCH(KERNEL, T)                                                              \
    hipLaunchKernelGGL((KERNEL<NS, PO, NR, CACHE, T, LIN>), dim3(c.pat.n_slices), dim3(T),         \
                       patch_lds_bytes(c, jacobian), c.stream, c.d_model, c.nv, c.d_slice_boff,    \
                       c.d_patch_cell_ptr, c.d_patch_cells, c.d_patch_halo_ptr, c.d_patch_halo,    \
                       c.d_coords, c.d_u, c.d_uold, c.d_uold1, sc, c.d_ext[0], c.d_ext[1],         \
                       c.d_ext[2], c.d_ext[3], c.d_val, c.d_F, mode, acc_doubles,                  \
                       c.pat.max_patch_verts)
    if constexpr (PO && CACHE == 2 && NS >= 1 && !LIN) {
        bool ext = false;
        for (int s_ = 0; s_ < NS; ++s_) ext = ext || c.model.ext_nodes[s_] > 0;
        // (the third generation takes patches of up to 384 cells -- a second cell for some threads --, the second
        // one cell per thread of at most 256)
        const bool gen3_ok = c.assembly_lean >= 3 && lean3_applies(c);
        if (mode == 0 && !ext && c.assembly_lean >= 2 && (c.pat.max_patch_cells <= 256 || gen3_ok)) {
            // one cell per thread: 192 threads where every patch has at most 192 cells (tensor-product
            // meshes: 160; compact patches of an unstructured mesh: 170-190), else 256
            const int T = c.pat.max_patch_cells <= 192 ? 192 : 256;
            const int acc_row = jacobian ? c.pat.max_patch_width * NEQ * SLICE : 0;
            const size_t lds_bytes = sizeof(double) * ((size_t)acc_row + SLICE * NEQ + 2 * c.pat.max_patch_verts +
                                                       (size_t)(NEQ + 2 * NS) * c.pat.max_patch_verts +
                                                       (size_t)LeanStash<NR>::N * T);
#define FEDM_LEAN2_LAUNCH_T(KERNEL, LIST, N, TT)                                                            \
    hipLaunchKernelGGL((KERNEL<NS, NR, TT>), dim3(N), dim3(TT), lds_bytes, c.stream, c.d_model,             \
                       c.nv, c.d_slice_boff, c.d_patch_cell_ptr, c.d_patch_cells, c.d_patch_halo_ptr,       \
                       c.d_patch_halo, c.d_coords, c.d_u, c.d_uold, c.d_uold1, sc, c.d_ext[0], c.d_ext[1],  \
                       c.d_ext[2], c.d_ext[3], c.d_val, c.d_F, mode, acc_row, c.pat.max_patch_verts,        \
                       c.xcd_remap ? 1 : 0, LIST, cmask)
#define FEDM_LEAN2_LAUNCH(KERNEL, LIST, N)                                                                  \
    do {                                                                                                    \
        if (T == 192) FEDM_LEAN2_LAUNCH_T(KERNEL, LIST, N, 192);                                            \
        else FEDM_LEAN2_LAUNCH_T(KERNEL, LIST, N, 256);                                                     \
    } while (0)
#define FEDM_LEAN2_BOTH(LIST, N)                                                                            \
    do {                                                                                                    \
        if ((N) <= 0) break;                                                                                \
        if (jacobian) {                                                                              \
            FEDM_LEAN2_LAUNCH(assemble_lean2_kernel, LIST, N);                                              \
        } else {                                                                                            \
            FEDM_LEAN2_LAUNCH(residual_lean2_kernel, LIST, N);                                              \
        }                                                                                                   \
    } while (0)
            // the constant potential-potential plane: written by the first full assembly, kept afterwards
            const uint32_t cmask = (jacobian && c.skip_const_planes && c.const_planes_valid) ? c.const_plane_mask : 0u;
            // third generation (assemble3.hip: one pass over the cells, compile-time plane mask) where it is
            // instantiated; FEDM_ASSEMBLY_LEAN=2 keeps the row-phase kernels below
            const bool gen3 = gen3_ok;
#define FEDM_LEAN3_OR(LIST, N) (gen3 && launch_assemble_lean3(c, jacobian, LIST, N, cmask))
            if (c.halo_pending && c.comm && c.comm->d_patch_interior) {
                // the ghost values of the new state travel on the communication stream while the
                // patches that stage no ghost vertex are assembled (north_star: "ghost exchange
                // overlapped with interior assembly"); the patches that do follow the exchange
                Comm &cm = *c.comm;
                comm_halo_begin(c);
                if (!FEDM_LEAN3_OR(cm.d_patch_interior, cm.n_patch_interior))
                    FEDM_LEAN2_BOTH(cm.d_patch_interior, cm.n_patch_interior);
                comm_halo_exchange(c, c.d_u);
                if (!FEDM_LEAN3_OR(cm.d_patch_boundary, cm.n_patch_boundary))
                    FEDM_LEAN2_BOTH(cm.d_patch_boundary, cm.n_patch_boundary);
                c.halo_pending = false;
            } else {
                flush_pending_halo(c);
                if (!FEDM_LEAN3_OR((const int *)nullptr, c.pat.n_slices))
                    FEDM_LEAN2_BOTH((const int *)nullptr, c.pat.n_slices);
            }
#undef FEDM_LEAN3_OR
#undef FEDM_LEAN2_BOTH
#undef FEDM_LEAN2_LAUNCH
#undef FEDM_LEAN2_LAUNCH_T
            if (jacobian) c.const_planes_valid = true;
            return;
        }
    }
    flush_pending_halo(c);
    if (jacobian) {
        if (c.pat.max_patch_cells <= 192) FEDM_PATCH_LAUNCH(assemble_patch_kernel, 192);
        else FEDM_PATCH_LAUNCH(assemble_patch_kernel, 320);
    } else {
        if (c.pat.max_patch_cells <= 192) FEDM_PATCH_LAUNCH(residual_patch_kernel, 192);
        else FEDM_PATCH_LAUNCH(residual_patch_kernel, 320);
    }
#undef FEDM_PATCH_LAUNCH
}

// =============================================================================================
// Neumann boundary facets (fedm/functions.py:523-524): one thread per tagged facet, one launch
// per facet colour (facets of a colour share no vertex), plain adds on top of the volume
// assembly -> fixed summation order.
// =============================================================================================
template <int NS, bool ATOMIC>
__global__ void boundary_kernel(const fedm_model_desc *__restrict__ md, int n_facets,
                                const int *__restrict__ facets /* [n][3] = cell, local facet, tag */,
                                const int *__restrict__ cells, const double *__restrict__ coords,
                                const uint32_t *__restrict__ cell_slots,
                                const double *__restrict__ u, double *__restrict__ val,
                                double *__restrict__ F, int jacobian) {
    constexpr int NEQ = NS + 1, NEQ2 = NEQ * NEQ;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_facets) return;
    const int c = facets[3 * t], fi = facets[3 * t + 1], tag = facets[3 * t + 2];
    int v[3];
    double x[3][2], Uc[3][NEQ];
    for (int a = 0; a < 3; ++a) {
        v[a] = cells[3 * c + a];
        x[a][0] = coords[2 * v[a]];
        x[a][1] = coords[2 * v[a] + 1];
        for (int s = 0; s < NEQ; ++s) Uc[a][s] = u[(size_t)v[a] * NEQ + s];
    }
    auto addR = [&](int a, int s, double value) {
        double *p = &F[(size_t)v[a] * NEQ + s];
        if (ATOMIC) unsafeAtomicAdd(p, value);
        else *p += value;
    };
    auto addJ = [&](int a, int b, int sr, int scol, double value) {
        const uint32_t slot = cell_slots[(size_t)c * 9 + a * 3 + b];
        double *p = &val[((size_t)(slot >> 6) * NEQ2 + sr * NEQ + scol) * SLICE + (slot & 63)];
        if (ATOMIC) unsafeAtomicAdd(p, value);
        else *p += value;
    };
    boundary_facet<NS>(md, x, Uc, fi, tag, jacobian != 0, addR, addJ);
}

template <bool ATOMIC>
static void launch_boundary_range(Ctx &c, bool jacobian, int f0, int n) {
    const dim3 g((n + 127) / 128), b(128);
    const int *fl = c.d_bfacets + 3 * f0;
    switch (c.ns) {
        case 1: hipLaunchKernelGGL((boundary_kernel<1, ATOMIC>), g, b, 0, c.stream, c.d_model, n, fl, c.d_cells, c.d_coords, c.d_cell_slots, c.d_u, c.d_val, c.d_F, jacobian ? 1 : 0); break;
        case 2: hipLaunchKernelGGL((boundary_kernel<2, ATOMIC>), g, b, 0, c.stream, c.d_model, n, fl, c.d_cells, c.d_coords, c.d_cell_slots, c.d_u, c.d_val, c.d_F, jacobian ? 1 : 0); break;
        case 3: hipLaunchKernelGGL((boundary_kernel<3, ATOMIC>), g, b, 0, c.stream, c.d_model, n, fl, c.d_cells, c.d_coords, c.d_cell_slots, c.d_u, c.d_val, c.d_F, jacobian ? 1 : 0); break;
        case 4: hipLaunchKernelGGL((boundary_kernel<4, ATOMIC>), g, b, 0, c.stream, c.d_model, n, fl, c.d_cells, c.d_coords, c.d_cell_slots, c.d_u, c.d_val, c.d_F, jacobian ? 1 : 0); break;
    }
}

static void launch_boundary(Ctx &c, bool jacobian) {
    if (!c.poisson || c.n_bfacets == 0) return;
    if (c.assembly_kind == 1) {
        // the patch assembly already sums in a run-dependent order (LDS atomics): all facets in
        // one launch with fp64 atomics instead of one launch per colour
        launch_boundary_range<true>(c, jacobian, 0, c.n_bfacets);
        return;
    }
    const int ncol = (int)c.bfacet_colour_ptr.size() - 1;
    for (int k = 0; k < ncol; ++k) {
        const int f0 = c.bfacet_colour_ptr[k], n = c.bfacet_colour_ptr[k + 1] - f0;
        if (n > 0) launch_boundary_range<false>(c, jacobian, f0, n);
    }
}

template <int NS, bool PO, int NR, int CACHE, bool LIN = false>
static void assemble_variant(Ctx &c, bool jacobian, int mode) {
    if (c.assembly_kind == 1) assemble_patch_t<NS, PO, NR, CACHE, LIN>(c, jacobian, mode);
    else assemble_colour_t<NS, PO, NR, CACHE, LIN>(c, jacobian, mode);
}

template <int NS, bool PO>
static void assemble_dispatch(Ctx &c, bool jacobian, int mode) {
    constexpr int NEQ = NS + (PO ? 1 : 0);
    const bool few = c.model.n_reactions <= 1;
    // cache exp(u) at the quadrature points when the tensors are emitted in several row passes
    const bool cache = NEQ > 1 && c.model.n_qp <= 3;
    // FIAT's degree-2 rule (points (1/6,1/6), (1/6,2/3), (2/3,1/6), weights 1/6) as constants
    const fedm_model_desc &m = c.model;
    const double sixth = 1.0 / 6.0, two3 = 2.0 / 3.0;
    const bool stdq = cache && m.n_qp == 3 && m.qp_x[0] == sixth && m.qp_x[1] == sixth && m.qp_x[2] == two3 &&
                      m.qp_y[0] == sixth && m.qp_y[1] == two3 && m.qp_y[2] == sixth && m.qp_w[0] == sixth &&
                      m.qp_w[1] == sixth && m.qp_w[2] == sixth;
    // the non-logarithmic representation runs on the generic (uncached, any reaction count) element only
    if (m.linear_representation) assemble_variant<NS, PO, FEDM_MAX_REACTIONS, 0, true>(c, jacobian, mode);
    else if (few && stdq) assemble_variant<NS, PO, 1, (NEQ > 1) ? 2 : 0>(c, jacobian, mode);
    else if (few && cache) assemble_variant<NS, PO, 1, (NEQ > 1) ? 1 : 0>(c, jacobian, mode);
    else if (few) assemble_variant<NS, PO, 1, 0>(c, jacobian, mode);
    else if (cache) assemble_variant<NS, PO, FEDM_MAX_REACTIONS, (NEQ > 1) ? 1 : 0>(c, jacobian, mode);
    else assemble_variant<NS, PO, FEDM_MAX_REACTIONS, 0>(c, jacobian, mode);
}

void launch_assemble(Ctx &c, bool jacobian, int mode) {
    if (c.model_kind == 1) {
        flush_pending_halo(c);
        prof_begin(c, jacobian ? 0 : 2);
        launch_assemble_gd(c, jacobian, mode);
        prof_end(c);
        return;
    }
    if (jacobian) c.planes_fused = false;   // (set by the one-pass kernel when it forms the field split's planes itself)
    prof_begin(c, jacobian ? 0 : 2);  // the volume kernel only (all colours in variant 0)
    if (c.ns == 1 && !c.poisson) assemble_dispatch<1, false>(c, jacobian, mode);
    else if (c.ns == 1 && c.poisson) assemble_dispatch<1, true>(c, jacobian, mode);
    else if (c.ns == 2 && c.poisson) assemble_dispatch<2, true>(c, jacobian, mode);
    else if (c.ns == 2 && !c.poisson) assemble_dispatch<2, false>(c, jacobian, mode);
    else if (c.ns == 3 && c.poisson) assemble_dispatch<3, true>(c, jacobian, mode);
    else if (c.ns == 4 && c.poisson) assemble_dispatch<4, true>(c, jacobian, mode);
    prof_end(c);
    if (mode == 0) {
        static const bool fuse_off = [] {
            const char *e = std::getenv("FEDM_FUSED_BOUNDARY");
            return e && e[0] == '0';
        }();
        // one GPU, patch assembly (the facets in one launch, with atomics), no row shared with a Dirichlet dof: the
        // facets go into launch_finalize's launch
        if (!fuse_off && !c.comm && c.n_owned == c.nv && !c.d_identity && c.assembly_kind == 1 && c.poisson &&
            c.n_bfacets > 0 && c.boundary_rows_disjoint && c.ns >= 1 && c.ns <= 4)
            c.boundary_pending = jacobian ? 2 : 1;
        else
            launch_boundary(c, jacobian);
    }
}

// =============================================================================================
// Dirichlet rows (bc.apply(b, x): b_i = x_i - g_i; bc.apply(A): identity row), padding
// vertices and -- in Poisson-only mode -- frozen species rows.
// =============================================================================================
__device__ __forceinline__ void identity_row(double *val, const int *boff, const int *colidx,
                                             int neq, int vtx, int cr) {
    const int slice = vtx >> 6, lane = vtx & 63;
    const int neq2 = neq * neq;
    for (int bc = boff[slice]; bc < boff[slice + 1]; ++bc) {
        const int col = colidx[(size_t)bc * SLICE + lane];
        for (int cc = 0; cc < neq; ++cc)
            val[((size_t)bc * neq2 + cr * neq + cc) * SLICE + lane] = (col == vtx && cc == cr) ? 1.0 : 0.0;
    }
    // padded duplicates of the diagonal column (col == vtx beyond row_len) must stay zero:
    // they are only ever produced for j >= row_len where the first match already got the 1.
}

// Dirichlet rows (F = u - value, unit row) and, in the same launch when no species is frozen, the
// identity rows of the vertices [n_owned, nvp) (ghosts and padding): one kernel boundary less per
// assembly.  n_id = 0: Dirichlet rows only.
__global__ void dirichlet_kernel(int n_dir, const int *__restrict__ dofs,
                                 const double *__restrict__ vals, const double *__restrict__ u,
                                 double *__restrict__ F, double *__restrict__ val,
                                 const int *__restrict__ boff, const int *__restrict__ colidx,
                                 const uint32_t *__restrict__ diag_slot, int neq, int jacobian,
                                 int id_first, int n_id, const int *__restrict__ id_list, int n_list) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int neq2 = neq * neq;
    if (t >= n_dir) {
        if (t - n_dir >= n_id) return;
        // identity rows: the range [id_first, ...) or, with deep halos, the listed ghost vertices (the
        // outermost layer) followed by the padding range
        const int k = t - n_dir;
        const int vtx = id_list ? (k < n_list ? id_list[k] : id_first + (k - n_list)) : id_first + k;
        const int slice = vtx >> 6, lane = vtx & 63;
        for (int cr = 0; cr < neq; ++cr) {
            F[(size_t)vtx * neq + cr] = 0.0;
            if (!jacobian) continue;
            for (int bc = boff[slice]; bc < boff[slice + 1]; ++bc)
                for (int cc = 0; cc < neq; ++cc)
                    val[((size_t)bc * neq2 + cr * neq + cc) * SLICE + lane] = 0.0;
            const uint32_t ds = diag_slot[vtx];
            val[((size_t)(ds >> 6) * neq2 + cr * neq + cr) * SLICE + (ds & 63)] = 1.0;
        }
        return;
    }
    const int dof = dofs[t];
    const int vtx = dof / neq, cr = dof % neq;
    // a ghost's row (one layer: identity with F = 0, written above; deep halos: an assembled row whose
    // Dirichlet condition applies as on its owner)
    if (!id_list && n_id > 0 && vtx >= id_first) return;
    F[dof] = u[dof] - vals[t];
    if (!jacobian) return;
    const int slice = vtx >> 6, lane = vtx & 63;
    for (int bc = boff[slice]; bc < boff[slice + 1]; ++bc)
        for (int cc = 0; cc < neq; ++cc)
            val[((size_t)bc * neq2 + cr * neq + cc) * SLICE + lane] = 0.0;
    const uint32_t ds = diag_slot[vtx];
    val[((size_t)(ds >> 6) * neq2 + cr * neq + cr) * SLICE + (ds & 63)] = 1.0;
}

// rows that are identity by construction: padding vertices (all components) and, in
// Poisson-only mode, every species component of every vertex
__global__ void identity_rows_kernel(int nv, int nvp, int neq, int ns_frozen,
                                     double *__restrict__ F, double *__restrict__ val,
                                     const int *__restrict__ boff,
                                     const uint32_t *__restrict__ diag_slot, int jacobian) {
    const int vtx = blockIdx.x * blockDim.x + threadIdx.x;
    if (vtx >= nvp) return;
    const int neq2 = neq * neq;
    const int slice = vtx >> 6, lane = vtx & 63;
    const int n_rows = (vtx >= nv) ? neq : ns_frozen;
    for (int cr = 0; cr < n_rows; ++cr) {
        F[(size_t)vtx * neq + cr] = 0.0;
        if (!jacobian) continue;
        for (int bc = boff[slice]; bc < boff[slice + 1]; ++bc)
            for (int cc = 0; cc < neq; ++cc)
                val[((size_t)bc * neq2 + cr * neq + cc) * SLICE + lane] = 0.0;
        const uint32_t ds = diag_slot[vtx];
        val[((size_t)(ds >> 6) * neq2 + cr * neq + cr) * SLICE + (ds & 63)] = 1.0;
    }
}

// boundary_kernel<NS, true> and dirichlet_kernel in one launch (128-thread blocks: the first ones take the facets, the
// rest the Dirichlet and padding rows) -- allowed when no row belongs to both (Ctx::boundary_rows_disjoint)
template <int NS>
__global__ __launch_bounds__(128) void boundary_dirichlet_kernel(
    const fedm_model_desc *__restrict__ md, int n_facets, int facet_blocks, const int *__restrict__ facets,
    const int *__restrict__ cells, const double *__restrict__ coords, const uint32_t *__restrict__ cell_slots,
    const double *__restrict__ u, double *__restrict__ val, double *__restrict__ F, int jacobian, int n_dir,
    const int *__restrict__ dofs, const double *__restrict__ vals, const int *__restrict__ boff,
    const uint32_t *__restrict__ diag_slot, int id_first, int n_id) {
    constexpr int NEQ = NS + 1, NEQ2 = NEQ * NEQ;
    if ((int)blockIdx.x < facet_blocks) {
        const int t = blockIdx.x * blockDim.x + threadIdx.x;
        if (t >= n_facets) return;
        const int c = facets[3 * t], fi = facets[3 * t + 1], tag = facets[3 * t + 2];
        int v[3];
        double x[3][2], Uc[3][NEQ];
        for (int a = 0; a < 3; ++a) {
            v[a] = cells[3 * c + a];
            x[a][0] = coords[2 * v[a]];
            x[a][1] = coords[2 * v[a] + 1];
            for (int s = 0; s < NEQ; ++s) Uc[a][s] = u[(size_t)v[a] * NEQ + s];
        }
        auto addR = [&](int a, int s, double value) { unsafeAtomicAdd(&F[(size_t)v[a] * NEQ + s], value); };
        auto addJ = [&](int a, int b, int sr, int scol, double value) {
            const uint32_t slot = cell_slots[(size_t)c * 9 + a * 3 + b];
            unsafeAtomicAdd(&val[((size_t)(slot >> 6) * NEQ2 + sr * NEQ + scol) * SLICE + (slot & 63)], value);
        };
        boundary_facet<NS>(md, x, Uc, fi, tag, jacobian != 0, addR, addJ);
        return;
    }
    // the rows of dirichlet_kernel (one GPU: Dirichlet dofs, then the padding vertices [id_first, id_first + n_id))
    const int t = ((int)blockIdx.x - facet_blocks) * blockDim.x + threadIdx.x;
    if (t >= n_dir + n_id) return;
    if (t >= n_dir) {
        const int vtx = id_first + (t - n_dir), slice = vtx >> 6, lane = vtx & 63;
        for (int cr = 0; cr < NEQ; ++cr) {
            F[(size_t)vtx * NEQ + cr] = 0.0;
            if (!jacobian) continue;
            for (int bc = boff[slice]; bc < boff[slice + 1]; ++bc)
                for (int cc = 0; cc < NEQ; ++cc) val[((size_t)bc * NEQ2 + cr * NEQ + cc) * SLICE + lane] = 0.0;
            const uint32_t ds = diag_slot[vtx];
            val[((size_t)(ds >> 6) * NEQ2 + cr * NEQ + cr) * SLICE + (ds & 63)] = 1.0;
        }
        return;
    }
    const int dof = dofs[t], vtx = dof / NEQ, cr = dof % NEQ;
    F[dof] = u[dof] - vals[t];
    if (!jacobian) return;
    const int slice = vtx >> 6, lane = vtx & 63;
    for (int bc = boff[slice]; bc < boff[slice + 1]; ++bc)
        for (int cc = 0; cc < NEQ; ++cc) val[((size_t)bc * NEQ2 + cr * NEQ + cc) * SLICE + lane] = 0.0;
    const uint32_t ds = diag_slot[vtx];
    val[((size_t)(ds >> 6) * NEQ2 + cr * NEQ + cr) * SLICE + (ds & 63)] = 1.0;
}

void launch_finalize(Ctx &c, bool jacobian, int mode) {
    if (c.boundary_pending) {
        const bool jac = c.boundary_pending == 2;
        c.boundary_pending = 0;
        if (mode == 0 && jac == jacobian) {
            const int fb = (c.n_bfacets + 127) / 128, n_id = c.nvp - c.nv;
            const int rb = (c.n_dir + n_id + 127) / 128;
#define FEDM_BD(NS_)                                                                                              \
    hipLaunchKernelGGL(boundary_dirichlet_kernel<NS_>, dim3(fb + rb), dim3(128), 0, c.stream, c.d_model, c.n_bfacets, fb, \
                       c.d_bfacets, c.d_cells, c.d_coords, c.d_cell_slots, c.d_u, c.d_val, c.d_F, jacobian ? 1 : 0, c.n_dir, \
                       c.d_dir_dofs, c.d_dir_vals, c.d_slice_boff, c.d_diag_slot, c.nv, n_id)
            if (c.ns == 1) FEDM_BD(1);
            else if (c.ns == 2) FEDM_BD(2);
            else if (c.ns == 3) FEDM_BD(3);
            else FEDM_BD(4);
#undef FEDM_BD
            return;
        }
        launch_boundary(c, jac);   // (not the pair this was deferred for: the facets first, then the rows as usual)
    }
    const int ns_frozen = (mode == 1) ? c.ns : 0;
    const bool deep = c.d_identity != nullptr;
    // identity rows: padding + ghost vertices (deep halos: padding + the listed outermost ghost layer)
    int n_id = deep ? c.n_identity + (c.nvp - c.nv) : c.nvp - c.n_owned;
    if (ns_frozen > 0) {           // frozen species: every vertex has identity rows
        // (vertices beyond the first argument: all rows -- the ghosts of a one-layer halo and the padding;
        // deep halos: the padding only, the inner ghost layers keep their potential rows and the
        // outermost layer is listed)
        hipLaunchKernelGGL(identity_rows_kernel, dim3((c.nvp + 255) / 256), dim3(256), 0, c.stream,
                           deep ? c.nv : c.n_owned, c.nvp, c.neq, ns_frozen, c.d_F, c.d_val, c.d_slice_boff,
                           c.d_diag_slot, jacobian ? 1 : 0);
        n_id = deep ? c.n_identity : 0;
    }
    // (Dirichlet dofs of ghost vertices are left to the identity branch: the row sets are disjoint)
    if (c.n_dir + n_id > 0)
        hipLaunchKernelGGL(dirichlet_kernel, dim3((c.n_dir + n_id + 255) / 256), dim3(256), 0, c.stream,
                           c.n_dir, c.d_dir_dofs, c.d_dir_vals, c.d_u, c.d_F, c.d_val,
                           c.d_slice_boff, c.d_colidx, c.d_diag_slot, c.neq, jacobian ? 1 : 0,
                           deep ? c.nv : c.n_owned, n_id, c.d_identity, c.n_identity);
}

__global__ void set_dirichlet_state_kernel(int n_dir, const int *__restrict__ dofs,
                                           const double *__restrict__ vals, double *__restrict__ u) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_dir) u[dofs[t]] = vals[t];
}

void launch_set_dirichlet_state(Ctx &c) {
    if (c.n_dir > 0)
        hipLaunchKernelGGL(set_dirichlet_state_kernel, dim3((c.n_dir + 255) / 256), dim3(256), 0,
                           c.stream, c.n_dir, c.d_dir_dofs, c.d_dir_vals, c.d_u);
}

// =============================================================================================
// Point-block Jacobi: inverse of every vertex's n_eq x n_eq diagonal block (Gauss-Jordan with
// partial pivoting), stored sliced so that lanes are contiguous.
// =============================================================================================
template <int NEQ>
__global__ void block_inverse_kernel(int nvp, const double *__restrict__ val,
                                     const uint32_t *__restrict__ diag_slot,
                                     double *__restrict__ dinv) {
    constexpr int NEQ2 = NEQ * NEQ;
    const int vtx = blockIdx.x * blockDim.x + threadIdx.x;
    if (vtx >= nvp) return;
    const uint32_t ds = diag_slot[vtx];
    double A[NEQ][NEQ], I[NEQ][NEQ];
#pragma unroll
    for (int r = 0; r < NEQ; ++r)
#pragma unroll
        for (int cidx = 0; cidx < NEQ; ++cidx) {
            A[r][cidx] = val[((size_t)(ds >> 6) * NEQ2 + r * NEQ + cidx) * SLICE + (ds & 63)];
            I[r][cidx] = (r == cidx) ? 1.0 : 0.0;
        }
#pragma unroll
    for (int k = 0; k < NEQ; ++k) {
        // partial pivoting without dynamic register indexing: swap rows by predication
        int piv = k;
        double best = fabs(A[k][k]);
#pragma unroll
        for (int r = k + 1; r < NEQ; ++r)
            if (fabs(A[r][k]) > best) {
                best = fabs(A[r][k]);
                piv = r;
            }
#pragma unroll
        for (int r = k + 1; r < NEQ; ++r)
            if (piv == r) {
#pragma unroll
                for (int cidx = 0; cidx < NEQ; ++cidx) {
                    double t = A[k][cidx];
                    A[k][cidx] = A[r][cidx];
                    A[r][cidx] = t;
                    t = I[k][cidx];
                    I[k][cidx] = I[r][cidx];
                    I[r][cidx] = t;
                }
            }
        const double inv = 1.0 / A[k][k];
#pragma unroll
        for (int cidx = 0; cidx < NEQ; ++cidx) {
            A[k][cidx] *= inv;
            I[k][cidx] *= inv;
        }
#pragma unroll
        for (int r = 0; r < NEQ; ++r) {
            if (r == k) continue;
            const double f = A[r][k];
#pragma unroll
            for (int cidx = 0; cidx < NEQ; ++cidx) {
                A[r][cidx] -= f * A[k][cidx];
                I[r][cidx] -= f * I[k][cidx];
            }
        }
    }
    const int slice = vtx >> 6, lane = vtx & 63;
#pragma unroll
    for (int e = 0; e < NEQ2; ++e) dinv[((size_t)slice * NEQ2 + e) * SLICE + lane] = I[e / NEQ][e % NEQ];
}

void launch_block_inverse(Ctx &c) {
    const dim3 g((c.nvp + 255) / 256), b(256);
    switch (c.neq) {
        case 1: hipLaunchKernelGGL(block_inverse_kernel<1>, g, b, 0, c.stream, c.nvp, c.d_val, c.d_diag_slot, c.d_dinv); break;
        case 2: hipLaunchKernelGGL(block_inverse_kernel<2>, g, b, 0, c.stream, c.nvp, c.d_val, c.d_diag_slot, c.d_dinv); break;
        case 3: hipLaunchKernelGGL(block_inverse_kernel<3>, g, b, 0, c.stream, c.nvp, c.d_val, c.d_diag_slot, c.d_dinv); break;
        case 4: hipLaunchKernelGGL(block_inverse_kernel<4>, g, b, 0, c.stream, c.nvp, c.d_val, c.d_diag_slot, c.d_dinv); break;
        case 5: hipLaunchKernelGGL(block_inverse_kernel<5>, g, b, 0, c.stream, c.nvp, c.d_val, c.d_diag_slot, c.d_dinv); break;
        case 6: hipLaunchKernelGGL(block_inverse_kernel<6>, g, b, 0, c.stream, c.nvp, c.d_val, c.d_diag_slot, c.d_dinv); break;
    }
}

// =============================================================================================
// SpMV on the sliced block-ELL matrix: one wavefront per slice, one lane per vertex.
// Matrix values and column indices stream in coalesced (lanes contiguous); x is gathered per
// neighbour (n_eq contiguous doubles).  Optional fused block-Jacobi scaling y = Dinv (A x).
// =============================================================================================
// ZMASK: bit (r * NEQ + c) marks a value plane that is structurally zero for this model (no reaction
// couples the two species: d(electron row)/d(ion density) of the streamer model) -- it is neither
// loaded nor multiplied: one ninth of the streamer matrix's bytes.
template <int NEQ, bool FS, unsigned ZMASK = 0u>
__global__ __launch_bounds__(256) void spmv_kernel(int n_slices, int n_owned,
                                                   const int *__restrict__ boff,
                                                   const int *__restrict__ colidx,
                                                   const double *__restrict__ val,
                                                   const double *__restrict__ x,
                                                   double *__restrict__ y,
                                                   const double *__restrict__ dinv,
                                                   double *__restrict__ fs_z, double *__restrict__ fs_b0,
                                                   double fs_scale, const int *__restrict__ slice_list,
                                                   int fs_compact32 = 0, int xcd = 0) {
    constexpr int NEQ2 = NEQ * NEQ;
    // xcd: consecutive slices (neighbours in the Z-curve, sharing most of their x entries) on one XCD
    const int blk = (xcd & 1) ? xcd_contiguous(blockIdx.x, gridDim.x) : blockIdx.x;
    const int wave_id = blk * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (wave_id >= n_slices) return;  // n_slices: number of slices this launch covers
    const int slice = slice_list ? slice_list[wave_id] : wave_id;
    const int b0 = boff[slice], b1 = boff[slice + 1];
    // epilogue operands (block inverse) requested before the gather loop
    constexpr int ND = FS ? (NEQ - 1) * (NEQ - 1) : NEQ2;
    double dv[ND > 0 ? ND : 1];
    if (FS || dinv) {
        const double *dp = dinv + (size_t)slice * ND * SLICE + lane;
#pragma unroll
        for (int e = 0; e < ND; ++e) dv[e] = dp[(size_t)e * SLICE];
    }
    double acc[NEQ];
#pragma unroll
    for (int r = 0; r < NEQ; ++r) acc[r] = 0.0;
    // xcd bit 1: the matrix values by non-temporal loads (launch_spmv: matrices beyond half the Infinity Cache).  Read
    // once per product, they then neither go through the cache nor push the vectors, the preconditioner's planes and
    // the multigrid out of it: the 4 M-DOF product 136 -> 123 us (69 -> 76 % of the HBM peak), the 1 M-DOF time step
    // -2.7 % and the developed streamer's -3.6 % although the product by itself, back to back, slows from 28 to 35 us
    // there (its 177 MB would have stayed in the cache if nothing else ran).
    auto products = [&](auto nt_c) {
        constexpr bool NT = decltype(nt_c)::value;
        for (int bc = b0; bc < b1; ++bc) {
            const int col = colidx[(size_t)bc * SLICE + lane];
            double xj[NEQ];
#pragma unroll
            for (int cc = 0; cc < NEQ; ++cc) xj[cc] = x[(size_t)col * NEQ + cc];
            const double *vp = val + (size_t)bc * NEQ2 * SLICE + lane;
#pragma unroll
            for (int r = 0; r < NEQ; ++r)
#pragma unroll
                for (int cc = 0; cc < NEQ; ++cc)
                    if (!((ZMASK >> (r * NEQ + cc)) & 1u)) {
                        const double *ap = &vp[(size_t)(r * NEQ + cc) * SLICE];
                        acc[r] += (NT ? __builtin_nontemporal_load(ap) : *ap) * xj[cc];
                    }
        }
    };
    if (xcd & 2) products(std::true_type{});
    else products(std::false_type{});
    const size_t vtx = (size_t)slice * SLICE + lane;
    if ((int)vtx >= n_owned) {  // ghost / padding rows belong to someone else (or to nobody)
#pragma unroll
        for (int r = 0; r < NEQ; ++r) acc[r] = 0.0;
    }
    if (FS) {
        // first stage of the field-split preconditioner in the epilogue (amg.hip):
        // t = A x is kept, z_u = fs_scale * Duu^-1 t_u starts the species sweeps, b0 = t_phi
        constexpr int NS = NEQ - 1;
#pragma unroll
        for (int r = 0; r < NEQ; ++r) y[vtx * NEQ + r] = acc[r];
#pragma unroll
        for (int r = 0; r < NS; ++r) {
            double z = 0.0;
#pragma unroll
            for (int cc = 0; cc < NS; ++cc) z += dv[r * NS + cc] * acc[cc];
            // (with sweeps to follow the first species iterate is a compact single-precision vector)
            if (fs_compact32) reinterpret_cast<float *>(fs_z)[vtx * NS + r] = (float)(fs_scale * z);
            else fs_z[vtx * NEQ + r] = fs_scale * z;
        }
        if (!fs_compact32) fs_z[vtx * NEQ + NS] = 0.0;  // whole lines are written; the V-cycle result lands here later
        fs_b0[vtx] = acc[NS];
    } else if (dinv) {
#pragma unroll
        for (int r = 0; r < NEQ; ++r) {
            double z = 0.0;
#pragma unroll
            for (int cc = 0; cc < NEQ; ++cc) z += dv[r * NEQ + cc] * acc[cc];
            y[vtx * NEQ + r] = z;
        }
    } else {
#pragma unroll
        for (int r = 0; r < NEQ; ++r) y[vtx * NEQ + r] = acc[r];
    }
}

// bit 1 of the products' `xcd` argument: the Jacobian's values by non-temporal loads when the matrix is larger than half
// the Infinity Cache (256 MiB) -- see spmv_kernel; FEDM_SPMV_NT=0 / 1 forces it off / on
static int spmv_nontemporal(const Ctx &c) {
    static const int forced = [] {
        const char *e = std::getenv("FEDM_SPMV_NT");
        return e ? (e[0] == '0' ? 0 : 1) : -1;
    }();
    if (forced >= 0) return forced ? 2 : 0;
    const double bytes = (double)c.pat.total_bc * SLICE * c.neq * c.neq * sizeof(double);
    return bytes > 128.0 * 1024.0 * 1024.0 ? 2 : 0;
}

// slice_list != nullptr: only those n_list matrix slices (interior / boundary halves across GPUs)
void launch_spmv(Ctx &c, const double *x, double *y, bool scale_dinv, const int *slice_list, int n_list) {
    const int n = slice_list ? n_list : c.pat.n_slices;
    if (n == 0) return;
    const dim3 g((n + 3) / 4), b(256);
    const double *dinv = scale_dinv ? c.d_dinv : nullptr;
#define FEDM_SPMV_Z(NEQ, Z)                                                                           \
    hipLaunchKernelGGL((spmv_kernel<NEQ, false, Z>), g, b, 0, c.stream, n, c.n_owned, c.d_slice_boff, \
                       c.d_colidx, c.d_val, x, y, dinv, (double *)nullptr, (double *)nullptr, 0.0,         \
                       slice_list, 0, ((c.xcd_remap && !slice_list) ? 1 : 0) | spmv_nontemporal(c))
#define FEDM_SPMV(NEQ) FEDM_SPMV_Z(NEQ, 0u)
    switch (c.neq) {
        case 1: FEDM_SPMV(1); break;
        case 2: FEDM_SPMV(2); break;
        case 3:  // two species + potential: the species-species planes (0,1) / (1,0) may be zero
            switch (c.zero_plane_mask & 10u) {
                case 2u: FEDM_SPMV_Z(3, 2u); break;
                case 8u: FEDM_SPMV_Z(3, 8u); break;
                case 10u: FEDM_SPMV_Z(3, 10u); break;
                default: FEDM_SPMV(3); break;
            }
            break;
        case 4: FEDM_SPMV(4); break;
        case 5: FEDM_SPMV(5); break;
        case 6: FEDM_SPMV(6); break;
    }
#undef FEDM_SPMV
#undef FEDM_SPMV_Z
}

// t = A x together with the first field-split stage (c.d_dinv holds the species-block inverses);
// slice_list != nullptr: only those n_list matrix slices (interior / boundary halves across GPUs)
void launch_spmv_fieldsplit(Ctx &c, const double *x, double *t, double *z, double *b0, double scale,
                            const int *slice_list, int n_list, bool compact32) {
    const int n = slice_list ? n_list : c.pat.n_slices;
    if (n == 0) return;
    const dim3 g((n + 3) / 4), b(256);
#define FEDM_SPMV_Z(NEQ, Z)                                                                          \
    hipLaunchKernelGGL((spmv_kernel<NEQ, true, Z>), g, b, 0, c.stream, n, c.n_owned, c.d_slice_boff, \
                       c.d_colidx, c.d_val, x, t, c.d_dinv, z, b0, scale, slice_list, compact32 ? 1 : 0,      \
                       ((c.xcd_remap && !slice_list) ? 1 : 0) | spmv_nontemporal(c))
#define FEDM_SPMV(NEQ) FEDM_SPMV_Z(NEQ, 0u)
    switch (c.neq) {
        case 2: FEDM_SPMV(2); break;
        case 3:
            switch (c.zero_plane_mask & 10u) {
                case 2u: FEDM_SPMV_Z(3, 2u); break;
                case 8u: FEDM_SPMV_Z(3, 8u); break;
                case 10u: FEDM_SPMV_Z(3, 10u); break;
                default: FEDM_SPMV(3); break;
            }
            break;
        case 4: FEDM_SPMV(4); break;
        case 5: FEDM_SPMV(5); break;
        case 6: FEDM_SPMV(6); break;
    }
#undef FEDM_SPMV
#undef FEDM_SPMV_Z
}

// y = alpha * Dinv x
template <int NEQ>
__global__ void apply_dinv_kernel(int nvp, const double *__restrict__ dinv,
                                  const double *__restrict__ x, double *__restrict__ y, double alpha) {
    constexpr int NEQ2 = NEQ * NEQ;
    const int vtx = blockIdx.x * blockDim.x + threadIdx.x;
    if (vtx >= nvp) return;
    const int slice = vtx >> 6, lane = vtx & 63;
    const double *dp = dinv + (size_t)slice * NEQ2 * SLICE + lane;
    double xv[NEQ];
#pragma unroll
    for (int cc = 0; cc < NEQ; ++cc) xv[cc] = x[(size_t)vtx * NEQ + cc];
#pragma unroll
    for (int r = 0; r < NEQ; ++r) {
        double z = 0.0;
#pragma unroll
        for (int cc = 0; cc < NEQ; ++cc) z += dp[(size_t)(r * NEQ + cc) * SLICE] * xv[cc];
        y[(size_t)vtx * NEQ + r] = alpha * z;
    }
}

void launch_apply_dinv(Ctx &c, const double *x, double *y, double alpha) {
    const dim3 g((c.nvp + 255) / 256), b(256);
    switch (c.neq) {
        case 1: hipLaunchKernelGGL(apply_dinv_kernel<1>, g, b, 0, c.stream, c.nvp, c.d_dinv, x, y, alpha); break;
        case 2: hipLaunchKernelGGL(apply_dinv_kernel<2>, g, b, 0, c.stream, c.nvp, c.d_dinv, x, y, alpha); break;
        case 3: hipLaunchKernelGGL(apply_dinv_kernel<3>, g, b, 0, c.stream, c.nvp, c.d_dinv, x, y, alpha); break;
        case 4: hipLaunchKernelGGL(apply_dinv_kernel<4>, g, b, 0, c.stream, c.nvp, c.d_dinv, x, y, alpha); break;
        case 5: hipLaunchKernelGGL(apply_dinv_kernel<5>, g, b, 0, c.stream, c.nvp, c.d_dinv, x, y, alpha); break;
        case 6: hipLaunchKernelGGL(apply_dinv_kernel<6>, g, b, 0, c.stream, c.nvp, c.d_dinv, x, y, alpha); break;
    }
}

// =============================================================================================
// Reductions: deterministic two-stage (per-block partials in a fixed grid, then one block).
// =============================================================================================
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

template <int K>
__device__ __forceinline__ void block_reduce_store(double (&acc)[K], double *partials, int kbase) {
    __shared__ double sm[4][K];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < K; ++i) {
        const double s = wave_sum(acc[i]);
        if (lane == 0) sm[wave][i] = s;
    }
    __syncthreads();
    if (threadIdx.x < K) {
        const double s = sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x];
        partials[PARTIAL_AT(blockIdx.x, kbase + threadIdx.x)] = s;
    }
}

struct PtrPack8 {
    const double *p[8];
};

// partials[block][kbase + i] = sum over the block's grid-stride range of xs[i] * y
template <int K>
__global__ __launch_bounds__(256) void dots_kernel(PtrPack8 xs, const double *__restrict__ y,
                                                   size_t n, double *__restrict__ partials, int kbase) {
    double acc[K];
#pragma unroll
    for (int i = 0; i < K; ++i) acc[i] = 0.0;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n;
         idx += (size_t)gridDim.x * blockDim.x) {
        const double yv = y[idx];
#pragma unroll
        for (int i = 0; i < K; ++i) acc[i] += xs.p[i][idx] * yv;
    }
    block_reduce_store<K>(acc, partials, kbase);
}

__global__ void reduce_partials_kernel(const double *__restrict__ partials, int nblocks, int k,
                                       double *__restrict__ out) {
    // one wave per output; fixed summation order
    const int i = blockIdx.x;
    if (i >= k) return;
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 64) s += partials[PARTIAL_AT(b, i)];
    s = wave_sum(s);
    if (threadIdx.x == 0) out[i] = s;
}

// GMRES orthogonalisation step in ONE reduction: the dots kernel has produced h_i = v_i . w
// (i < k-1) and ww = w . w (slot k-1); after the (all-)reduction one thread derives
// |w - sum h_i v_i|^2 = ww - sum h_i^2 (Pythagoras; V orthonormal).
// out[k-1] <- that squared norm, out[RED_K-2] <- ww, out[RED_K-1] <- scale for the update
// (1/norm, or 1 when cancellation is too strong to trust the formula -> host refines).
// Results go to the host through a mailbox in host-mapped pinned memory: values, a system-scope
// fence, then the sequence tag the host polls (wait_red) -- no copy kernel, no stream
// synchronisation, and the host can queue the next iteration while this one finishes.
// The tag is a launch counter kept in device memory (a replayed graph has fixed arguments).
__device__ __forceinline__ void publish(const double *__restrict__ red, int k, double *mail,
                                        unsigned long long *seq) {
    // one wave; MAIL_SLOTS mailbox slots, chosen by the tag's low bits: the host may still be reading
    // publication n when n+1 and n+2 (Krylov steps launched ahead) arrive
    const unsigned long long tag = *seq + 1;
    double *slot = mail + (tag & (MAIL_SLOTS - 1)) * (RED_K + 1);
    for (int i = threadIdx.x; i < k; i += 64) slot[i] = red[i];
    __threadfence_system();
    if (threadIdx.x == 0) {
        *seq = tag;
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(slot + RED_K), tag, __ATOMIC_RELEASE,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__global__ __launch_bounds__(64) void publish_kernel(const double *__restrict__ red, int k, double *mail,
                                                     unsigned long long *seq) {
    publish(red, k, mail, seq);
}

__global__ __launch_bounds__(64) void cgs_finish_kernel(int k, double *__restrict__ out, double *mail,
                                                        unsigned long long *seq) {
    if (threadIdx.x == 0) {
        const double ww = out[k - 1];
        double hh = 0.0;
        for (int i = 0; i < k - 1; ++i) hh += out[i] * out[i];
        const double hn2 = ww - hh;
        out[RED_K - 2] = ww;
        out[k - 1] = hn2;
        out[RED_K - 1] = (hn2 > 1e-8 * ww && hn2 > 0.0) ? 1.0 / sqrt(hn2) : 1.0;
    }
    __syncthreads();
    publish(out, RED_K, mail, seq);
}

// Single-GPU Krylov step, two kernels instead of five:
//  dots_scatter_kernel  finishes the preconditioner (the potential component of y is taken from
//                       the V-cycle result x0 and stored into y: replaces fs_scatter_kernel) and
//                       forms the per-block partial dot products like dots_kernel;
//  reduce_finish_kernel reduces the partials of all k slots in the fixed order of
//                       reduce_partials_kernel, applies the cgs_finish_kernel formulae and
//                       publishes to the host mailbox.
// Every kernel boundary costs ~4-5 us on this GPU, more than the work of these small kernels.
// (A single kernel with a last-block-done ticket was tried: the agent-scope fences it needs
// across the 8 XCDs' L2s cost 60 us.)
template <int K>
__global__ __launch_bounds__(256) void dots_scatter_kernel(PtrPack8 xs, double *__restrict__ y, size_t n,
                                                           double *__restrict__ partials, int kbase,
                                                           int self_index, const double *__restrict__ x0,
                                                           int neq) {
    double acc[K];
#pragma unroll
    for (int i = 0; i < K; ++i) acc[i] = 0.0;
    if (x0) {
        // one vertex per thread: its potential component comes from x0 and is stored into y
        const size_t nvert = n / (size_t)neq;
        for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvert;
             v += (size_t)gridDim.x * blockDim.x) {
            const size_t base = v * neq;
            const double phi = x0[v];
            y[base + neq - 1] = phi;
            for (int cidx = 0; cidx < neq; ++cidx) {
                const double yv = (cidx == neq - 1) ? phi : y[base + cidx];
#pragma unroll
                for (int i = 0; i < K; ++i) acc[i] += (i == self_index ? yv : xs.p[i][base + cidx]) * yv;
            }
        }
    } else {
        for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n;
             idx += (size_t)gridDim.x * blockDim.x) {
            const double yv = y[idx];
#pragma unroll
            for (int i = 0; i < K; ++i) acc[i] += (i == self_index ? yv : xs.p[i][idx]) * yv;
        }
    }
    block_reduce_store<K>(acc, partials, kbase);
}

__global__ __launch_bounds__(1024) void reduce_finish_kernel(const double *__restrict__ partials, int nblocks,
                                                            int k, double *__restrict__ out, double *mail,
                                                            unsigned long long *seq) {
    __shared__ double fin[RED_K];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x < RED_K) fin[threadIdx.x] = (threadIdx.x == RED_SPARE) ? out[RED_SPARE] : 0.0;
    __syncthreads();
    // 16 waves, one slot each per pass: a Krylov step's j + 2 slots finish in one or two passes
    for (int i = wave; i < k; i += 16) {
        double sum = 0.0;
        for (int b = lane; b < nblocks; b += 64) sum += partials[PARTIAL_AT(b, i)];
        sum = wave_sum(sum);
        if (lane == 0) fin[i] = sum;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double ww = fin[k - 1];
        double hh = 0.0;
        for (int i = 0; i < k - 1; ++i) hh += fin[i] * fin[i];
        const double hn2 = ww - hh;
        fin[RED_K - 2] = ww;
        fin[k - 1] = hn2;
        fin[RED_K - 1] = (hn2 > 1e-8 * ww && hn2 > 0.0) ? 1.0 / sqrt(hn2) : 1.0;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const unsigned long long tag = *seq + 1;
        double *slot = mail + (tag & (MAIL_SLOTS - 1)) * (RED_K + 1);
        for (int i = threadIdx.x; i < RED_K; i += 64) {
            out[i] = fin[i];
            slot[i] = fin[i];
        }
        __threadfence_system();
        if (threadIdx.x == 0) {
            *seq = tag;
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(slot + RED_K), tag, __ATOMIC_RELEASE,
                               __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// Krylov steps with at most eight reduction slots (the first seven steps of a solve: all there are early in a
// streamer run, most of them later): the Jacobian product w = J z and the step's dot products v_i . w, w . w in ONE kernel -- the wave
// that has formed a slice's rows of w multiplies them with the same rows of the basis vectors before it stores them
// (w is not read back: 8 MB and a 9 us kernel less per step).  One partial per workgroup and slot,
// partials[slot * n_blocks + block]; spmv_dots_finish_kernel reduces them in a fixed order.
template <int NEQ, unsigned ZMASK, int K>
__global__ __launch_bounds__(256) void spmv_dots_kernel(int n_slices, int n_owned, const int *__restrict__ boff,
                                                        const int *__restrict__ colidx,
                                                        const double *__restrict__ val,
                                                        const double *__restrict__ x, double *__restrict__ y,
                                                        PtrPack8 xs, double *__restrict__ partials, int xcd) {
    constexpr int NEQ2 = NEQ * NEQ;
    const int blk = (xcd & 1) ? xcd_contiguous(blockIdx.x, gridDim.x) : blockIdx.x;
    const int wave_id = blk * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const bool live = wave_id < n_slices;          // (no early return: the workgroup reduces together)
    const int slice = live ? wave_id : 0;
    const int b0 = boff[slice], b1 = live ? boff[slice + 1] : b0;
    double acc[NEQ];
#pragma unroll
    for (int r = 0; r < NEQ; ++r) acc[r] = 0.0;
    // xcd bit 1: the matrix values by non-temporal loads (launch_spmv: matrices beyond half the Infinity Cache).  Read
    // once per product, they then neither go through the cache nor push the vectors, the preconditioner's planes and
    // the multigrid out of it: the 4 M-DOF product 136 -> 123 us (69 -> 76 % of the HBM peak), the 1 M-DOF time step
    // -2.7 % and the developed streamer's -3.6 % although the product by itself, back to back, slows from 28 to 35 us
    // there (its 177 MB would have stayed in the cache if nothing else ran).
    auto products = [&](auto nt_c) {
        constexpr bool NT = decltype(nt_c)::value;
        for (int bc = b0; bc < b1; ++bc) {
            const int col = colidx[(size_t)bc * SLICE + lane];
            double xj[NEQ];
#pragma unroll
            for (int cc = 0; cc < NEQ; ++cc) xj[cc] = x[(size_t)col * NEQ + cc];
            const double *vp = val + (size_t)bc * NEQ2 * SLICE + lane;
#pragma unroll
            for (int r = 0; r < NEQ; ++r)
#pragma unroll
                for (int cc = 0; cc < NEQ; ++cc)
                    if (!((ZMASK >> (r * NEQ + cc)) & 1u)) {
                        const double *ap = &vp[(size_t)(r * NEQ + cc) * SLICE];
                        acc[r] += (NT ? __builtin_nontemporal_load(ap) : *ap) * xj[cc];
                    }
        }
    };
    if (xcd & 2) products(std::true_type{});
    else products(std::false_type{});
    const size_t vtx = (size_t)slice * SLICE + lane;
    const bool owned = live && (int)vtx < n_owned;
    double d[K];
#pragma unroll
    for (int i = 0; i < K; ++i) d[i] = 0.0;
    if (owned) {
#pragma unroll
        for (int i = 0; i < K - 1; ++i)
#pragma unroll
            for (int r = 0; r < NEQ; ++r) d[i] += xs.p[i][vtx * NEQ + r] * acc[r];
#pragma unroll
        for (int r = 0; r < NEQ; ++r) d[K - 1] += acc[r] * acc[r];
    }
    if (live) {
#pragma unroll
        for (int r = 0; r < NEQ; ++r) y[vtx * NEQ + r] = owned ? acc[r] : 0.0;
    }
    __shared__ double sm[4][K];
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < K; ++i) {
        const double t = wave_sum(d[i]);
        if (lane == 0) sm[wave][i] = t;
    }
    __syncthreads();
    if (threadIdx.x < K)
        partials[(size_t)threadIdx.x * gridDim.x + blockIdx.x] =
            sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x];
}

// reduce_finish_kernel for the partials of spmv_dots_kernel (one per workgroup of the product: thousands, not
// RED_BLOCKS): the 16 waves share the k <= 8 slots, wave w sums the blocks of chunk w / k of slot w % k, the chunks
// are added in their order; then the formulae and the publication of cgs_finish_kernel.
__global__ __launch_bounds__(1024) void spmv_dots_finish_kernel(const double *__restrict__ partials, int nblocks,
                                                               int k, double *__restrict__ out, double *mail,
                                                               unsigned long long *seq, int finish) {
    __shared__ double fin[RED_K];
    __shared__ double part[16];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x < RED_K) fin[threadIdx.x] = (threadIdx.x == RED_SPARE) ? out[RED_SPARE] : 0.0;
    const int chunks = 16 / k, slot = wave % k, chunk = wave / k;
    double sum = 0.0;
    if (chunk < chunks)
        for (int b = chunk * 64 + lane; b < nblocks; b += chunks * 64) sum += partials[(size_t)slot * nblocks + b];
    sum = wave_sum(sum);
    if (lane == 0) part[wave] = sum;
    __syncthreads();
    if (threadIdx.x < k) {
        double t = 0.0;
        for (int ch = 0; ch < chunks; ++ch) t += part[ch * k + threadIdx.x];
        fin[threadIdx.x] = t;
        if (!finish) out[threadIdx.x] = t;   // several GPUs: the local sums, for the all-reduce that follows
    }
    if (!finish) return;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double ww = fin[k - 1];
        double hh = 0.0;
        for (int i = 0; i < k - 1; ++i) hh += fin[i] * fin[i];
        const double hn2 = ww - hh;
        fin[RED_K - 2] = ww;
        fin[k - 1] = hn2;
        fin[RED_K - 1] = (hn2 > 1e-8 * ww && hn2 > 0.0) ? 1.0 / sqrt(hn2) : 1.0;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const unsigned long long tag = *seq + 1;
        double *slot_ = mail + (tag & (MAIL_SLOTS - 1)) * (RED_K + 1);
        for (int i = threadIdx.x; i < RED_K; i += 64) {
            out[i] = fin[i];
            slot_[i] = fin[i];
        }
        __threadfence_system();
        if (threadIdx.x == 0) {
            *seq = tag;
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(slot_ + RED_K), tag, __ATOMIC_RELEASE,
                               __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

template <int K, bool FINAL>
__global__ void cgs_update_kernel(size_t n, const double *__restrict__ coef, int base, PtrPack8 xs,
                                  double *__restrict__ y) {
    double cf[K];
#pragma unroll
    for (int k = 0; k < K; ++k) cf[k] = coef[base + k];
    const double scale = FINAL ? coef[RED_K - 1] : 1.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        double s = y[i];
#pragma unroll
        for (int k = 0; k < K; ++k) s -= cf[k] * xs.p[k][i];
        y[i] = FINAL ? s * scale : s;
    }
}

static dim3 vec_grid(const Ctx &c);

static int red_grid(const Ctx &c) {
    const size_t blocks = ((size_t)c.n_dot + 255) / 256;
    return (int)(blocks < (size_t)RED_BLOCKS ? blocks : (size_t)RED_BLOCKS);
}

void launch_dots(Ctx &c, const double *const *xs, const double *y, int k, bool finish) {
    const int grid = red_grid(c);
    int done = 0;
    while (done < k) {
        const int kk = (k - done) >= 8 ? 8 : (k - done);
        PtrPack8 pk;
        for (int i = 0; i < 8; ++i) pk.p[i] = xs[done + (i < kk ? i : 0)];
        switch (kk) {
            case 1: hipLaunchKernelGGL(dots_kernel<1>, dim3(grid), dim3(256), 0, c.stream, pk, y, (size_t)c.n_dot, c.d_partials, done); break;
            case 2: hipLaunchKernelGGL(dots_kernel<2>, dim3(grid), dim3(256), 0, c.stream, pk, y, (size_t)c.n_dot, c.d_partials, done); break;
            case 3: hipLaunchKernelGGL(dots_kernel<3>, dim3(grid), dim3(256), 0, c.stream, pk, y, (size_t)c.n_dot, c.d_partials, done); break;
            case 4: hipLaunchKernelGGL(dots_kernel<4>, dim3(grid), dim3(256), 0, c.stream, pk, y, (size_t)c.n_dot, c.d_partials, done); break;
            case 5: hipLaunchKernelGGL(dots_kernel<5>, dim3(grid), dim3(256), 0, c.stream, pk, y, (size_t)c.n_dot, c.d_partials, done); break;
            case 6: hipLaunchKernelGGL(dots_kernel<6>, dim3(grid), dim3(256), 0, c.stream, pk, y, (size_t)c.n_dot, c.d_partials, done); break;
            case 7: hipLaunchKernelGGL(dots_kernel<7>, dim3(grid), dim3(256), 0, c.stream, pk, y, (size_t)c.n_dot, c.d_partials, done); break;
            default: hipLaunchKernelGGL(dots_kernel<8>, dim3(grid), dim3(256), 0, c.stream, pk, y, (size_t)c.n_dot, c.d_partials, done); break;
        }
        done += kk;
    }
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(k), dim3(64), 0, c.stream, c.d_partials, grid, k, c.d_red);
    comm_allreduce(c, c.d_red, k);
    if (finish) {
        hipLaunchKernelGGL(cgs_finish_kernel, dim3(1), dim3(64), 0, c.stream, k, c.d_red, c.h_mail,
                           c.d_mail_seq);
        if (!c.capturing) ++c.mail_seq;  // a captured launch counts when its graph is launched
    }
}

// single-GPU Krylov step: dots (+ the scatter of the V-cycle result x0 into the potential
// component of y), then reduction + finish + publication
void launch_cgs_finish(Ctx &c, int k) {
    hipLaunchKernelGGL(cgs_finish_kernel, dim3(1), dim3(64), 0, c.stream, k, c.d_red, c.h_mail, c.d_mail_seq);
    if (!c.capturing) ++c.mail_seq;
}

// finish = false (several GPUs): only the local sums, d_red[0..k), for the all-reduce that follows
void launch_dots_fused(Ctx &c, const double *const *xs, double *y, int k, const double *x0, bool finish) {
    const int grid = red_grid(c);
    int done = 0;
    while (done < k) {
        const int kk = (k - done) >= 8 ? 8 : (k - done);
        PtrPack8 pk;
        int self = -1;
        for (int i = 0; i < 8; ++i) {
            pk.p[i] = xs[done + (i < kk ? i : 0)];
            if (i < kk && pk.p[i] == y) self = i;
        }
        const double *sc = (done == 0) ? x0 : nullptr;
#define FEDM_DF(K)                                                                                     \
    hipLaunchKernelGGL(dots_scatter_kernel<K>, dim3(grid), dim3(256), 0, c.stream, pk, y, (size_t)c.n_dot, \
                       c.d_partials, done, self, sc, c.neq)
        switch (kk) {
            case 1: FEDM_DF(1); break;
            case 2: FEDM_DF(2); break;
            case 3: FEDM_DF(3); break;
            case 4: FEDM_DF(4); break;
            case 5: FEDM_DF(5); break;
            case 6: FEDM_DF(6); break;
            case 7: FEDM_DF(7); break;
            default: FEDM_DF(8); break;
        }
#undef FEDM_DF
        done += kk;
    }
    if (!finish) {
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(k), dim3(64), 0, c.stream, c.d_partials, grid, k, c.d_red);
        return;
    }
    hipLaunchKernelGGL(reduce_finish_kernel, dim3(1), dim3(1024), 0, c.stream, c.d_partials, grid, k, c.d_red,
                       c.h_mail, c.d_mail_seq);
    if (!c.capturing) ++c.mail_seq;
}

// w = J z with the step's k = j + 2 reduction slots (xs[0 .. k-2] . w and w . w), finished and published: one GPU,
// three species-plus-potential equations, k <= 8, buffers of ensure_spmv_dots.  false: not applicable (nothing was
// launched; the caller runs launch_spmv + launch_dots_fused).
static bool spmv_dots_applicable(const Ctx &c, int k) {
    static const bool off = [] {
        const char *e = std::getenv("FEDM_SPMV_DOTS");
        return e && e[0] == '0';
    }();
    // (several GPUs: where the whole product is one launch -- deep halos -- with finish = false)
    return !off && c.neq == 3 && k >= 2 && k <= 8 && c.d_partials_wide;
}

bool launch_spmv_dots(Ctx &c, const double *x, double *y, const double *const *xs, int k, bool finish) {
    if (!spmv_dots_applicable(c, k)) return false;
    const int n = c.pat.n_slices;
    const dim3 g((n + 3) / 4), b(256);
    PtrPack8 pk;
    for (int i = 0; i < 8; ++i) pk.p[i] = xs[i < k - 1 ? i : 0];
    const int xcd = (c.xcd_remap ? 1 : 0) | spmv_nontemporal(c);
#define FEDM_SD(Z, K)                                                                                       \
    hipLaunchKernelGGL((spmv_dots_kernel<3, Z, K>), g, b, 0, c.stream, n, c.n_owned, c.d_slice_boff, c.d_colidx, \
                       c.d_val, x, y, pk, c.d_partials_wide, xcd)
#define FEDM_SD_K(Z)                                                                                        \
    do {                                                                                                    \
        if (k == 2) FEDM_SD(Z, 2);                                                                          \
        else if (k == 3) FEDM_SD(Z, 3);                                                                     \
        else if (k == 4) FEDM_SD(Z, 4);                                                                     \
        else if (k == 5) FEDM_SD(Z, 5);                                                                     \
        else if (k == 6) FEDM_SD(Z, 6);                                                                     \
        else if (k == 7) FEDM_SD(Z, 7);                                                                     \
        else FEDM_SD(Z, 8);                                                                                 \
    } while (0)
    switch (c.zero_plane_mask & 10u) {
        case 2u: FEDM_SD_K(2u); break;
        case 8u: FEDM_SD_K(8u); break;
        case 10u: FEDM_SD_K(10u); break;
        default: FEDM_SD_K(0u); break;
    }
#undef FEDM_SD_K
#undef FEDM_SD
    hipLaunchKernelGGL(spmv_dots_finish_kernel, dim3(1), dim3(1024), 0, c.stream, c.d_partials_wide, (int)g.x, k,
                       c.d_red, c.h_mail, c.d_mail_seq, finish ? 1 : 0);
    if (finish && !c.capturing) ++c.mail_seq;
    return true;
}

// (one partial per workgroup of the product and slot; allocated with the Krylov vectors, outside any capture)
int ensure_spmv_dots(Ctx &c) {
    if (c.d_partials_wide || c.neq != 3) return 0;
    const size_t blocks = (size_t)(c.pat.n_slices + 3) / 4;
    FEDM_HIP_CHECK(hipMalloc((void **)&c.d_partials_wide, sizeof(double) * 8 * blocks));
    return 0;
}

// y = (y - sum_i d_red[i] xs[i]) * d_red[RED_K-1], coefficients stay on the device
void launch_cgs_update(Ctx &c, int k, const double *const *xs, double *y) {
    int done = 0;
    while (done < k) {
        const int kk = (k - done) >= 8 ? 8 : (k - done);
        const bool fin = (done + kk == k);
        PtrPack8 pk;
        for (int i = 0; i < 8; ++i) pk.p[i] = xs[done + (i < kk ? i : 0)];
#define FEDM_CGS(K)                                                                               \
    if (fin) hipLaunchKernelGGL((cgs_update_kernel<K, true>), vec_grid(c), dim3(256), 0, c.stream, \
                                (size_t)c.np, c.d_red, done, pk, y);                               \
    else hipLaunchKernelGGL((cgs_update_kernel<K, false>), vec_grid(c), dim3(256), 0, c.stream,     \
                            (size_t)c.np, c.d_red, done, pk, y);
        switch (kk) {
            case 1: FEDM_CGS(1) break;
            case 2: FEDM_CGS(2) break;
            case 3: FEDM_CGS(3) break;
            case 4: FEDM_CGS(4) break;
            case 5: FEDM_CGS(5) break;
            case 6: FEDM_CGS(6) break;
            case 7: FEDM_CGS(7) break;
            default: FEDM_CGS(8) break;
        }
#undef FEDM_CGS
        done += kk;
    }
}

// One GPU, field split on the right with species sweeps: whoever completes a Krylov vector v_j also forms the first
// stage of the preconditioner's application to it -- g = Duu^-1 v_u (single precision, what the sweeps start from)
// and b0 = v_phi (the multigrid's right-hand side) -- while the vector's entries are in registers: the pointwise
// fs_species_kernel (amg.hip; 5.5 us of launch latency per Krylov step) is gone.  Same operations in the same
// order as cgs_update_kernel / scale_copy_kernel followed by fs_species_kernel.
template <int NS>
__device__ __forceinline__ void first_stage_of(int v, const double (&tv)[NS + 1], const double *__restrict__ dinv_uu,
                                               float *__restrict__ g32, double *__restrict__ b0) {
    const double *dp = dinv_uu + (size_t)(v >> 6) * NS * NS * SLICE + (v & 63);
#pragma unroll
    for (int r = 0; r < NS; ++r) {
        double acc = 0.0;
#pragma unroll
        for (int cidx = 0; cidx < NS; ++cidx) acc += dp[(size_t)(r * NS + cidx) * SLICE] * tv[cidx];
        g32[(size_t)v * NS + r] = (float)acc;
    }
    b0[v] = tv[NS];
}

template <int K, int NS>
__global__ __launch_bounds__(256) void cgs_update_fs_kernel(int nvp, const double *__restrict__ coef, PtrPack8 xs,
                                                            double *__restrict__ y,
                                                            const double *__restrict__ dinv_uu,
                                                            float *__restrict__ g32, double *__restrict__ b0) {
    constexpr int NEQ = NS + 1;
    double cf[K];
#pragma unroll
    for (int k = 0; k < K; ++k) cf[k] = coef[k];
    const double scale = coef[RED_K - 1];
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < nvp; v += gridDim.x * blockDim.x) {
        double tv[NEQ];
#pragma unroll
        for (int r = 0; r < NEQ; ++r) {
            const size_t i = (size_t)v * NEQ + r;
            double t = y[i];
#pragma unroll
            for (int k = 0; k < K; ++k) t -= cf[k] * xs.p[k][i];
            t *= scale;
            y[i] = t;
            tv[r] = t;
        }
        first_stage_of<NS>(v, tv, dinv_uu, g32, b0);
    }
}

template <int NS>
__global__ __launch_bounds__(256) void scale_copy_fs_kernel(int nvp, double a, const double *__restrict__ x,
                                                            double *__restrict__ y,
                                                            const double *__restrict__ dinv_uu,
                                                            float *__restrict__ g32, double *__restrict__ b0) {
    constexpr int NEQ = NS + 1;
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < nvp; v += gridDim.x * blockDim.x) {
        double tv[NEQ];
#pragma unroll
        for (int r = 0; r < NEQ; ++r) {
            const size_t i = (size_t)v * NEQ + r;
            tv[r] = a * x[i];
            y[i] = tv[r];
        }
        first_stage_of<NS>(v, tv, dinv_uu, g32, b0);
    }
}

// false: no fused kernel for this case (the caller launches the two kernels)
bool launch_cgs_update_fs(Ctx &c, int k, const double *const *xs, double *y, float *g32, double *b0) {
    if (c.ns != 2 || k < 1 || k > 4) return false;
    PtrPack8 pk;
    for (int i = 0; i < 8; ++i) pk.p[i] = xs[i < k ? i : 0];
    const dim3 g((c.nvp + 255) / 256), b(256);
#define FEDM_CGSF(K)                                                                                             \
    hipLaunchKernelGGL((cgs_update_fs_kernel<K, 2>), g, b, 0, c.stream, c.nvp, c.d_red, pk, y, c.d_dinv, g32, b0)
    switch (k) {
        case 1: FEDM_CGSF(1); break;
        case 2: FEDM_CGSF(2); break;
        case 3: FEDM_CGSF(3); break;
        default: FEDM_CGSF(4); break;
    }
#undef FEDM_CGSF
    return true;
}

bool launch_scale_copy_fs(Ctx &c, double a, const double *x, double *y, float *g32, double *b0) {
    if (c.ns != 2) return false;
    hipLaunchKernelGGL(scale_copy_fs_kernel<2>, dim3((c.nvp + 255) / 256), dim3(256), 0, c.stream, c.nvp, a, x, y,
                       c.d_dinv, g32, b0);
    return true;
}

__global__ void reduce_partials_slot_kernel(const double *__restrict__ partials, int nblocks,
                                            int k_src, double *__restrict__ out, int slot) {
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 64) s += partials[PARTIAL_AT(b, k_src)];
    s = wave_sum(s);
    if (threadIdx.x == 0) out[slot] = s;
}

void launch_norm2(Ctx &c, const double *x, int slot) {
    const int grid = red_grid(c);
    PtrPack8 pk;
    for (int i = 0; i < 8; ++i) pk.p[i] = x;
    hipLaunchKernelGGL(dots_kernel<1>, dim3(grid), dim3(256), 0, c.stream, pk, x, (size_t)c.n_dot, c.d_partials, RED_K - 1);
    hipLaunchKernelGGL(reduce_partials_slot_kernel, dim3(1), dim3(64), 0, c.stream, c.d_partials, grid, RED_K - 1, c.d_red, slot);
    comm_allreduce(c, c.d_red + slot, 1);
}

// wait for publication `seq` and point c.h_red at its slot
void wait_red_seq(Ctx &c, unsigned long long seq) {
    double *slot = c.h_mail + (seq & (MAIL_SLOTS - 1)) * (RED_K + 1);
    c.h_red = slot;
    const unsigned long long *tag = reinterpret_cast<const unsigned long long *>(slot + RED_K);
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; ++spins) {
        if (__atomic_load_n(tag, __ATOMIC_ACQUIRE) == seq) return;
        __builtin_ia32_pause();
        if ((spins & 0xffff) == 0xffff) {
            // a faulted or lost queue never publishes: fall back to the runtime's own wait
            const bool failed = hipStreamQuery(c.stream) != hipErrorNotReady &&
                                __atomic_load_n(tag, __ATOMIC_ACQUIRE) != seq;
            const bool late = std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120);
            // a peer that died leaves an RCCL kernel spinning on this stream: ask the communicator
            if (comm_poll_async_error(c)) {
                for (int i = 0; i < RED_K; ++i) slot[i] = std::nan("");
                return;
            }
            if (failed || late) {
                hipStreamSynchronize(c.stream);
                if (__atomic_load_n(tag, __ATOMIC_ACQUIRE) != seq) {
                    // (diagnostics of a publication that never came: expected tag, the tags in the mailbox, the
                    // device's counter)
                    unsigned long long dev_seq = 0;
                    hipMemcpy(&dev_seq, c.d_mail_seq, sizeof(dev_seq), hipMemcpyDeviceToHost);
                    std::fprintf(stderr, "fedm: publication %llu did not arrive (%s); mailbox tags", seq,
                                 failed ? "the queue is empty" : "waited 120 s");
                    for (int k = 0; k < MAIL_SLOTS; ++k)
                        std::fprintf(stderr, " %llu", *reinterpret_cast<const unsigned long long *>(
                                                          c.h_mail + (size_t)k * (RED_K + 1) + RED_K));
                    std::fprintf(stderr, ", host count %llu, device count %llu\n", c.mail_seq, dev_seq);
                    for (int i = 0; i < RED_K; ++i) slot[i] = std::nan("");
                }
                return;
            }
        }
    }
}

void wait_red(Ctx &c) { wait_red_seq(c, c.mail_seq); }

// |x|^2 into d_red[slot] and publication of d_red[0..k) in one go: on one GPU the reduction of the
// partial sums and the mailbox write are one kernel
__global__ __launch_bounds__(64) void reduce_slot_publish_kernel(const double *__restrict__ partials, int nblocks,
                                                                 int k_src, double *__restrict__ out, int slot,
                                                                 int k, double *mail, unsigned long long *seq) {
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 64) s += partials[PARTIAL_AT(b, k_src)];
    s = wave_sum(s);
    s = __shfl(s, 0, 64);
    if (threadIdx.x == 0) out[slot] = s;
    const unsigned long long tag = *seq + 1;
    double *m = mail + (tag & (MAIL_SLOTS - 1)) * (RED_K + 1);
    for (int i = threadIdx.x; i < k; i += 64) m[i] = (i == slot) ? s : out[i];
    __threadfence_system();
    if (threadIdx.x == 0) {
        *seq = tag;
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(m + RED_K), tag, __ATOMIC_RELEASE,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// |x|^2 into slot `slot` and d_red[0..k) into the mailbox, without the wait: the caller may queue
// more work behind the publication before it calls wait_red (what it queues runs while the
// numbers travel to the host)
void norm2_publish(Ctx &c, const double *x, int slot, int k, int k_sum) {
    if (c.comm) {  // the all-reduce sits between the reduction and the publication
        // k_sum > 1 (slot 0): d_red[1 .. k_sum) hold rank-local sums that have not been all-reduced yet
        // (|dx|^2, |x|^2 of the Newton update, the watched component's error sums): ONE all-reduce
        // carries them together with |x|^2 instead of one each
        if (slot == 0 && k_sum > 1) {
            const int grid = red_grid(c);
            PtrPack8 pk;
            for (int i = 0; i < 8; ++i) pk.p[i] = x;
            hipLaunchKernelGGL(dots_kernel<1>, dim3(grid), dim3(256), 0, c.stream, pk, x, (size_t)c.n_dot, c.d_partials, RED_K - 1);
            hipLaunchKernelGGL(reduce_partials_slot_kernel, dim3(1), dim3(64), 0, c.stream, c.d_partials, grid, RED_K - 1, c.d_red, 0);
            comm_allreduce(c, c.d_red, k_sum);
        } else {
            launch_norm2(c, x, slot);
        }
        hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(64), 0, c.stream, c.d_red, k, c.h_mail, c.d_mail_seq);
        ++c.mail_seq;
        return;
    }
    const int grid = red_grid(c);
    PtrPack8 pk;
    for (int i = 0; i < 8; ++i) pk.p[i] = x;
    hipLaunchKernelGGL(dots_kernel<1>, dim3(grid), dim3(256), 0, c.stream, pk, x, (size_t)c.n_dot, c.d_partials, RED_K - 1);
    hipLaunchKernelGGL(reduce_slot_publish_kernel, dim3(1), dim3(64), 0, c.stream, c.d_partials, grid, RED_K - 1,
                       c.d_red, slot, k, c.h_mail, c.d_mail_seq);
    ++c.mail_seq;
}

void norm2_read(Ctx &c, const double *x, int slot, int k) {
    norm2_publish(c, x, slot, k, 1);
    wait_red(c);
}

// src[0 .. k) into the mailbox without the wait; the publication's sequence number (wait_red_seq)
unsigned long long publish_values(Ctx &c, const double *src, int k) {
    hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(64), 0, c.stream, src, k, c.h_mail, c.d_mail_seq);
    return ++c.mail_seq;
}
// ... the launch alone (inside a stream capture: whoever replays the graph advances Ctx::mail_seq per replay)
void publish_values_queued(Ctx &c, const double *src, int k) {
    hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(64), 0, c.stream, src, k, c.h_mail, c.d_mail_seq);
}

void read_red(Ctx &c, int k) {
    hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(64), 0, c.stream, c.d_red, k, c.h_mail, c.d_mail_seq);
    ++c.mail_seq;
    wait_red(c);
}

// =============================================================================================
// Vector updates
// =============================================================================================
__global__ void axpy_kernel(size_t n, double a, const double *__restrict__ x, double *__restrict__ y) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        y[i] += a * x[i];
}
__global__ void scale_copy_kernel(size_t n, double a, const double *__restrict__ x, double *__restrict__ y) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        y[i] = a * x[i];
}

static dim3 vec_grid(const Ctx &c) {
    size_t blocks = ((size_t)c.np + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    return dim3((unsigned)blocks);
}

void launch_axpy(Ctx &c, double a, const double *x, double *y) {
    hipLaunchKernelGGL(axpy_kernel, vec_grid(c), dim3(256), 0, c.stream, (size_t)c.np, a, x, y);
}
// y = x / sqrt(d_red[slot]) (0 when that norm is 0 or not finite): normalisation without a host
// round trip
__global__ void normalise_copy_kernel(size_t n, const double *__restrict__ red, int slot,
                                      const double *__restrict__ x, double *__restrict__ y) {
    const double n2 = red[slot];
    const double a = (n2 > 0.0 && n2 < 1.7e308) ? 1.0 / sqrt(n2) : 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        y[i] = a * x[i];
}

void launch_normalise_copy(Ctx &c, int slot, const double *x, double *y) {
    hipLaunchKernelGGL(normalise_copy_kernel, vec_grid(c), dim3(256), 0, c.stream, (size_t)c.np, c.d_red, slot,
                       x, y);
}

void launch_scale_copy(Ctx &c, double a, const double *x, double *y) {
    hipLaunchKernelGGL(scale_copy_kernel, vec_grid(c), dim3(256), 0, c.stream, (size_t)c.np, a, x, y);
}

struct CoefPack8 {
    double c[8];
};

template <int K>
__global__ void multi_axpy_kernel(size_t n, CoefPack8 cf, PtrPack8 xs, double *__restrict__ y) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        double s = y[i];
#pragma unroll
        for (int k = 0; k < K; ++k) s += cf.c[k] * xs.p[k][i];
        y[i] = s;
    }
}

void launch_multi_axpy(Ctx &c, const double *coef_host, int k, const double *const *xs, double *y, double sign) {
    int done = 0;
    while (done < k) {
        const int kk = (k - done) >= 8 ? 8 : (k - done);
        PtrPack8 pk;
        CoefPack8 cf;
        for (int i = 0; i < 8; ++i) {
            pk.p[i] = xs[done + (i < kk ? i : 0)];
            cf.c[i] = i < kk ? sign * coef_host[done + i] : 0.0;
        }
        switch (kk) {
            case 1: hipLaunchKernelGGL(multi_axpy_kernel<1>, vec_grid(c), dim3(256), 0, c.stream, (size_t)c.np, cf, pk, y); break;
            case 2: hipLaunchKernelGGL(multi_axpy_kernel<2>, vec_grid(c), dim3(256), 0, c.stream, (size_t)c.np, cf, pk, y); break;
            case 3: hipLaunchKernelGGL(multi_axpy_kernel<3>, vec_grid(c), dim3(256), 0, c.stream, (size_t)c.np, cf, pk, y); break;
            case 4: hipLaunchKernelGGL(multi_axpy_kernel<4>, vec_grid(c), dim3(256), 0, c.stream, (size_t)c.np, cf, pk, y); break;
            case 5: hipLaunchKernelGGL(multi_axpy_kernel<5>, vec_grid(c), dim3(256), 0, c.stream, (size_t)c.np, cf, pk, y); break;
            case 6: hipLaunchKernelGGL(multi_axpy_kernel<6>, vec_grid(c), dim3(256), 0, c.stream, (size_t)c.np, cf, pk, y); break;
            case 7: hipLaunchKernelGGL(multi_axpy_kernel<7>, vec_grid(c), dim3(256), 0, c.stream, (size_t)c.np, cf, pk, y); break;
            default: hipLaunchKernelGGL(multi_axpy_kernel<8>, vec_grid(c), dim3(256), 0, c.stream, (size_t)c.np, cf, pk, y); break;
        }
        done += kk;
    }
}

// Newton update in one pass: delta = sum_i c_i z_i, u += delta, and the per-block partial sums of
// |delta|^2 and |u|^2 over the owned entries (slots 1, 2 of the reduction buffer) -- instead of a
// memset, a multi-axpy, an axpy and two norm kernels.
template <int K>
__global__ __launch_bounds__(256) void newton_update_kernel(size_t n, size_t n_dot, CoefPack8 cf, PtrPack8 zs,
                                                            double *__restrict__ u, double *__restrict__ delta,
                                                            double *__restrict__ partials) {
    double acc[2] = {0.0, 0.0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        double d = 0.0;
#pragma unroll
        for (int k = 0; k < K; ++k) d += cf.c[k] * zs.p[k][i];
        const double un = u[i] + d;
        u[i] = un;
        if (delta) delta[i] = d;   // (nobody reads it after a solve that converged in its first cycle)
        if (i < n_dot) {
            acc[0] += d * d;
            acc[1] += un * un;
        }
    }
    block_reduce_store<2>(acc, partials, 1);
}

__global__ void reduce_partials_range_kernel(const double *__restrict__ partials, int nblocks, int k0,
                                             double *__restrict__ out) {
    const int i = k0 + blockIdx.x;  // one wave per output; fixed summation order
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 64) s += partials[PARTIAL_AT(b, i)];
    s = wave_sum(s);
    if (threadIdx.x == 0) out[i] = s;
}

void launch_newton_update(Ctx &c, const double *coef_host, int k, const double *const *zs, double *u,
                          double *delta) {
    const int grid = red_grid(c);
    PtrPack8 pk;
    CoefPack8 cf;
    for (int i = 0; i < 8; ++i) {
        pk.p[i] = zs[i < k ? i : 0];
        cf.c[i] = i < k ? coef_host[i] : 0.0;
    }
#define FEDM_NU(K)                                                                                         \
    hipLaunchKernelGGL(newton_update_kernel<K>, dim3(grid), dim3(256), 0, c.stream, (size_t)c.np, (size_t)c.n_dot, \
                       cf, pk, u, delta, c.d_partials)
    switch (k) {
        case 1: FEDM_NU(1); break;
        case 2: FEDM_NU(2); break;
        case 3: FEDM_NU(3); break;
        case 4: FEDM_NU(4); break;
        case 5: FEDM_NU(5); break;
        case 6: FEDM_NU(6); break;
        case 7: FEDM_NU(7); break;
        default: FEDM_NU(8); break;
    }
#undef FEDM_NU
    hipLaunchKernelGGL(reduce_partials_range_kernel, dim3(2), dim3(64), 0, c.stream, c.d_partials, grid, 1, c.d_red);
    // several GPUs: d_red[1], d_red[2] stay rank-local; they are all-reduced together with the next |F|^2
    // (norm2_publish, k_sum = 3): Ctx::red12_local tells the Newton loop
    c.red12_local = c.comm != nullptr;
}

// |new - old + eps|^2 and |old + eps|^2 over one component (fedm/functions.py:1062-1064)
__global__ __launch_bounds__(256) void field_error_kernel(int nv, int neq, int comp,
                                                          const double *__restrict__ u,
                                                          const double *__restrict__ uold,
                                                          double *__restrict__ partials, int slot0) {
    const double eps = 3.0e-16;  // DOLFIN_EPS
    double acc[2] = {0.0, 0.0};
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += gridDim.x * blockDim.x) {
        const double a = u[(size_t)v * neq + comp], b = uold[(size_t)v * neq + comp];
        const double d = a - b + eps, o = b + eps;
        acc[0] += d * d;
        acc[1] += o * o;
    }
    block_reduce_store<2>(acc, partials, slot0);
}

void launch_field_error(Ctx &c, int comp) {
    int grid = (c.n_owned + 255) / 256;
    if (grid > RED_BLOCKS) grid = RED_BLOCKS;
    hipLaunchKernelGGL(field_error_kernel, dim3(grid), dim3(256), 0, c.stream, c.n_owned, c.neq, comp, c.d_u, c.d_uold, c.d_partials, 0);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(2), dim3(64), 0, c.stream, c.d_partials, grid, 2, c.d_red);
    comm_allreduce(c, c.d_red, 2);
}

// the same two sums into d_red[3], d_red[4]: they ride on the next publication (several GPUs: rank-local
// until norm2_publish all-reduces them with |F|^2)
void launch_field_error_slots34(Ctx &c, int comp) {
    int grid = (c.n_owned + 255) / 256;
    if (grid > RED_BLOCKS) grid = RED_BLOCKS;
    hipLaunchKernelGGL(field_error_kernel, dim3(grid), dim3(256), 0, c.stream, c.n_owned, c.neq, comp, c.d_u, c.d_uold, c.d_partials, 3);
    hipLaunchKernelGGL(reduce_partials_range_kernel, dim3(2), dim3(64), 0, c.stream, c.d_partials, grid, 3, c.d_red);
}

// =============================================================================================
// Copy ceiling of the box: 16-byte-per-lane copies (what MI355X_MICROARCH.md measures 6.29 TB/s
// with), read + write bytes over HIP-event time.  Buffers far beyond the 256 MiB Infinity Cache;
// bench.py prints the rate next to the 8 TB/s specification.  A ceiling has to be the best a copy
// reaches on the box, so several shapes are timed and the fastest one is reported: U loads in
// flight per lane before the first store (round 3's single load per trip left the memory system
// half empty: 4.8 TB/s, below what the Jacobian product moves), plain or non-temporal.
// =============================================================================================
typedef float copy_f4 __attribute__((ext_vector_type(4)));   // (the non-temporal builtins take native vectors)
template <int U, bool NT>
__global__ __launch_bounds__(256) void copy16_kernel(const copy_f4 *__restrict__ src, copy_f4 *__restrict__ dst, size_t n) {
    // a workgroup takes contiguous chunks of U * 256 pieces; the grid strides over the chunks
    const size_t chunk = (size_t)U * 256;
    for (size_t base = (size_t)blockIdx.x * chunk; base < n; base += (size_t)gridDim.x * chunk) {
        copy_f4 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const size_t i = base + (size_t)k * 256 + threadIdx.x;
            if (i < n) {
                if (NT) v[k] = __builtin_nontemporal_load(src + i);
                else v[k] = src[i];
            }
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const size_t i = base + (size_t)k * 256 + threadIdx.x;
            if (i < n) {
                if (NT) __builtin_nontemporal_store(v[k], dst + i);
                else dst[i] = v[k];
            }
        }
    }
}

int copy_bandwidth(int device, int64_t bytes, int repeats, double *gbs) {
    FEDM_HIP_CHECK(hipSetDevice(device));
    const size_t n = (size_t)bytes / sizeof(copy_f4);
    copy_f4 *a = nullptr, *b = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = -1;
    do {
        if (hipMalloc((void **)&a, n * sizeof(copy_f4)) != hipSuccess) break;
        if (hipMalloc((void **)&b, n * sizeof(copy_f4)) != hipSuccess) break;
        if (hipMemset(a, 1, n * sizeof(copy_f4)) != hipSuccess) break;
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) break;
        double best = 0.0;
        bool failed = false;
        for (int variant = 0; variant < 6 && !failed; ++variant) {
            // grids: as many workgroups as there are chunks, up to 16 per CU (256 CUs)
            static const int unroll_of[6] = {1, 4, 4, 8, 8, 2};
            const size_t per = (size_t)unroll_of[variant] * 256;
            const unsigned g = (unsigned)std::min<size_t>((n + per - 1) / per, (size_t)256 * 16);
            auto launch = [&]() {
                switch (variant) {
                    case 0: hipLaunchKernelGGL((copy16_kernel<1, false>), dim3(256 * 8), dim3(256), 0, 0, a, b, n); break;
                    case 1: hipLaunchKernelGGL((copy16_kernel<4, false>), dim3(g), dim3(256), 0, 0, a, b, n); break;
                    case 2: hipLaunchKernelGGL((copy16_kernel<4, true>), dim3(g), dim3(256), 0, 0, a, b, n); break;
                    case 3: hipLaunchKernelGGL((copy16_kernel<8, false>), dim3(g), dim3(256), 0, 0, a, b, n); break;
                    case 4: hipLaunchKernelGGL((copy16_kernel<8, true>), dim3(g), dim3(256), 0, 0, a, b, n); break;
                    default: hipLaunchKernelGGL((copy16_kernel<2, false>), dim3(g), dim3(256), 0, 0, a, b, n); break;
                }
            };
            launch();
            if (hipDeviceSynchronize() != hipSuccess) { failed = true; break; }
            hipEventRecord(e0, 0);
            for (int i = 0; i < repeats; ++i) launch();
            hipEventRecord(e1, 0);
            if (hipEventSynchronize(e1) != hipSuccess) { failed = true; break; }
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            const double rate = 2.0 * (double)(n * sizeof(copy_f4)) * repeats / ((double)ms * 1e-3) / 1e9;
            if (std::getenv("FEDM_COPY_VERBOSE")) fprintf(stderr, "copy_bandwidth variant %d: %.0f GB/s\n", variant, rate);
            best = std::max(best, rate);
        }
        if (failed) break;
        *gbs = best;
        rc = 0;
    } while (false);
    if (rc) set_error("copy_bandwidth: HIP call failed (out of memory?)");
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    if (a) hipFree(a);
    if (b) hipFree(b);
    return rc;
}

}  // namespace fedm
