// Device side of the multigrid preconditioner for the potential block and the field-split
// (block-triangular) combination with point-block Jacobi on the species rows.
// Every operator is a scalar sliced-ELL matrix (64 rows per slice, lanes contiguous), so the
// V-cycle is a sequence of coalesced, HBM/L2-bound SpMV kernels with fused epilogues.
#include "amg.hpp"
#include "comm.hpp"
#include "species_planes.hpp"

#include <algorithm>
#include <cmath>
#include <vector>
#include <cstdlib>

namespace fedm {

// ---- scalar sliced-ELL --------------------------------------------------------------------
int EllMat::from_csr(const fedm_csr &m, bool want_dinv, int l2s) {
    n_rows = m.n_rows;
    n_cols = m.n_cols;
    n_rows_p = ((n_rows + SLICE - 1) / SLICE) * SLICE;
    if (const char *e = std::getenv("FEDM_ELL_SPLIT")) l2s = std::atoi(e);
    if (l2s < 0) {
        const double avg = n_rows ? (double)m.indptr[n_rows] / n_rows : 0.0;
        l2s = 0;
        while (l2s < 4 && avg / (1 << l2s) > 6.0 && ((int64_t)n_rows << l2s) < (int64_t)1 << 20) ++l2s;
    }
    log2_split = l2s;
    const int split = 1 << l2s;
    const int64_t n_lr = (int64_t)n_rows * split;  // lane-rows
    n_slices = (int)((n_lr + SLICE - 1) / SLICE);
    auto len_of = [&](int64_t lr) -> int {
        const int r = (int)(lr >> l2s), k = (int)(lr & (split - 1));
        const int len = (int)(m.indptr[r + 1] - m.indptr[r]);
        return (len - k + split - 1) / split;
    };
    std::vector<int> boff_h(n_slices + 1, 0);
    for (int s = 0; s < n_slices; ++s) {
        int w = 0;
        for (int l = 0; l < SLICE; ++l) {
            const int64_t lr = (int64_t)s * SLICE + l;
            if (lr < n_lr) w = std::max(w, len_of(lr));
        }
        boff_h[s + 1] = boff_h[s] + w;
    }
    // uniform width (no offset lookup in the kernels) when padding to the widest slice costs
    // little: <= 25 % more entries, or a matrix so small that only latency matters
    {
        int wmax = 0;
        for (int s = 0; s < n_slices; ++s) wmax = std::max(wmax, boff_h[s + 1] - boff_h[s]);
        const int64_t uniform = (int64_t)wmax * n_slices;
        width = 0;
        if (uniform <= boff_h[n_slices] + boff_h[n_slices] / 4 || uniform * SLICE <= (int64_t)1 << 20) {
            width = std::max(wmax, 1);
            for (int s = 0; s <= n_slices; ++s) boff_h[s] = s * width;
        }
    }
    total_bc = boff_h[n_slices];
    std::vector<int> col_h((size_t)total_bc * SLICE, 0);
    std::vector<double> val_h((size_t)total_bc * SLICE, 0.0);
    std::vector<double> dinv_h(n_rows_p, 1.0);
    for (int s = 0; s < n_slices; ++s)
        for (int l = 0; l < SLICE; ++l) {
            const int64_t lr = (int64_t)s * SLICE + l;
            if (lr >= n_lr) continue;
            const int r = (int)(lr >> l2s), k0 = (int)(lr & (split - 1));
            int j = 0;
            for (int64_t k = m.indptr[r] + k0; k < m.indptr[r + 1]; k += split, ++j) {
                const size_t slot = (size_t)(boff_h[s] + j) * SLICE + l;
                if (m.indices[k] < 0 || m.indices[k] >= n_cols) return -2;
                col_h[slot] = m.indices[k];
                val_h[slot] = m.values[k];
                if (want_dinv && m.indices[k] == r && m.values[k] != 0.0) dinv_h[r] = 1.0 / m.values[k];
            }
        }
    if (hipMalloc((void **)&boff, sizeof(int) * boff_h.size()) != hipSuccess) return -1;
    if (hipMalloc((void **)&col, sizeof(int) * std::max<size_t>(col_h.size(), 1)) != hipSuccess) return -1;
    hipMemcpy(boff, boff_h.data(), sizeof(int) * boff_h.size(), hipMemcpyHostToDevice);
    hipMemcpy(col, col_h.data(), sizeof(int) * col_h.size(), hipMemcpyHostToDevice);
    if (single) {
        std::vector<float> v32(val_h.begin(), val_h.end());
        if (hipMalloc((void **)&val32, sizeof(float) * std::max<size_t>(v32.size(), 1)) != hipSuccess) return -1;
        hipMemcpy(val32, v32.data(), sizeof(float) * v32.size(), hipMemcpyHostToDevice);
    } else {
        if (hipMalloc((void **)&val, sizeof(double) * std::max<size_t>(val_h.size(), 1)) != hipSuccess) return -1;
        hipMemcpy(val, val_h.data(), sizeof(double) * val_h.size(), hipMemcpyHostToDevice);
    }
    if (want_dinv) {
        if (hipMalloc((void **)&dinv, sizeof(double) * n_rows_p) != hipSuccess) return -1;
        hipMemcpy(dinv, dinv_h.data(), sizeof(double) * n_rows_p, hipMemcpyHostToDevice);
    }
    return 0;
}

void EllMat::release() {
    if (boff) hipFree(boff);
    if (col) hipFree(col);
    if (val) hipFree(val);
    if (val32) hipFree(val32);
    if (dinv) hipFree(dinv);
    boff = col = nullptr;
    val = dinv = nullptr;
    val32 = nullptr;
}

// MODE 0: y = A x      1: y = b - A x      2: y = x + omega*dinv*(b - A x)     3: y += A x
// MODE 4: first sweep from a zero guess fused with the residual:
//         x1 = omega*dinv*b (written to `aux`),  y = b - A x1   (x is unused)
// These kernels move little data; their duration is a chain of dependent memory round trips
// (~1 us each).  So: no slice-offset lookup when the matrix has a uniform width (`width` > 0),
// the row's own operands (b, dinv, x, y of the epilogue) are requested before the gather loop,
// and four independent gather chains are in flight.
template <int MODE, typename VT>
__global__ __launch_bounds__(256) void ell_spmv_kernel(int n_slices, int n_rows, int log2_split, int width,
                                                       const int *__restrict__ boff,
                                                       const int *__restrict__ col,
                                                       const VT *__restrict__ val,
                                                       const double *__restrict__ dinv,
                                                       const double *__restrict__ x,
                                                       const double *__restrict__ b,
                                                       double *__restrict__ y, double omega,
                                                       double *__restrict__ aux, int ystride, int yoff,
                                                       const int *__restrict__ slice_list, int xcd) {
    // xcd: a contiguous range of slices per XCD (neighbouring rows share most of their x entries: one L2)
    int blk = blockIdx.x;
    if (xcd) {
        const int per = gridDim.x >> 3, full = per << 3;
        blk = blk < full ? (blk & 7) * per + (blk >> 3) : blk;
    }
    const int wave_id = blk * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (wave_id >= n_slices) return;   // n_slices: number of slices this launch covers
    const int slice = slice_list ? slice_list[wave_id] : wave_id;
    const int b0 = width > 0 ? slice * width : boff[slice];
    const int b1 = width > 0 ? b0 + width : boff[slice + 1];
    const size_t r = ((size_t)slice * SLICE + lane) >> log2_split;
    const bool writer = (lane & ((1 << log2_split) - 1)) == 0 && r < (size_t)n_rows;
    double e_b = 0.0, e_d = 0.0, e_x = 0.0;  // epilogue operands of this row, loaded up front
    if (writer) {
        if (MODE == 1 || MODE == 2 || MODE == 4 || MODE == 5) e_b = b[r];
        if (MODE == 2 || MODE == 4 || MODE == 5) e_d = dinv[r];
        if (MODE == 2) e_x = x[r];
        if (MODE == 3) e_x = y[r];
    }
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int bc = b0;
    for (; bc + 4 <= b1; bc += 4) {
        const size_t k = (size_t)bc * SLICE + lane;
        const int c0 = col[k], c1 = col[k + SLICE], c2 = col[k + 2 * SLICE], c3 = col[k + 3 * SLICE];
        const double v0 = val[k], v1 = val[k + SLICE], v2 = val[k + 2 * SLICE], v3 = val[k + 3 * SLICE];
        double x0, x1, x2, x3;
        if (MODE == 4) {
            x0 = omega * dinv[c0] * b[c0];
            x1 = omega * dinv[c1] * b[c1];
            x2 = omega * dinv[c2] * b[c2];
            x3 = omega * dinv[c3] * b[c3];
        } else {
            x0 = x[c0];
            x1 = x[c1];
            x2 = x[c2];
            x3 = x[c3];
        }
        a0 += v0 * x0;
        a1 += v1 * x1;
        a2 += v2 * x2;
        a3 += v3 * x3;
    }
    for (; bc < b1; ++bc) {
        const size_t k = (size_t)bc * SLICE + lane;
        const int cc = col[k];
        a0 += val[k] * (MODE == 4 ? omega * dinv[cc] * b[cc] : x[cc]);
    }
    double acc = (a0 + a1) + (a2 + a3);
    for (int off = (1 << log2_split) >> 1; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (!writer) return;
    if (MODE == 0) y[r] = acc;
    if (MODE == 1) y[r] = e_b - acc;
    if (MODE == 2) y[r * ystride + yoff] = e_x + omega * e_d * (e_b - acc);  // strided: see Amg::out
    if (MODE == 3) y[r] = e_x + acc;
    if (MODE == 5) y[r] = omega * e_d * e_b + acc;
    if (MODE == 4) {
        y[r] = e_b - acc;
        aux[r] = omega * e_d * e_b;
    }
}

static void ell_launch(Ctx &c, const EllMat &A, int mode, const double *x, const double *b,
                       double *y, double omega, double *aux = nullptr, int ystride = 1, int yoff = 0,
                       const double *dinv = nullptr /* default: the matrix's own */,
                       const int *slice_list = nullptr, int n_list = 0 /* only these matrix slices */) {
    const int n = slice_list ? n_list : A.n_slices;
    if (n == 0) return;
    const dim3 g((n + 3) / 4), bl(256);
    if (!dinv) dinv = A.dinv;
    static const bool xcd_ok = [] {
        const char *e = std::getenv("FEDM_ELL_XCD");
        return !(e && e[0] == '0');
    }();
    const int xcd = (xcd_ok && !slice_list && g.x >= 64) ? 1 : 0;
#define FEDM_ELL(M)                                                                                          \
    do {                                                                                                     \
        if (A.val32)                                                                                         \
            hipLaunchKernelGGL((ell_spmv_kernel<M, float>), g, bl, 0, c.stream, n, A.n_rows, A.log2_split,   \
                               A.width, A.boff, A.col, A.val32, dinv, x, b, y, omega, aux, ystride, yoff,    \
                               slice_list, xcd);                                                             \
        else                                                                                                 \
            hipLaunchKernelGGL((ell_spmv_kernel<M, double>), g, bl, 0, c.stream, n, A.n_rows, A.log2_split,  \
                               A.width, A.boff, A.col, A.val, dinv, x, b, y, omega, aux, ystride, yoff,      \
                               slice_list, xcd);                                                             \
    } while (0)
    switch (mode) {
        case 0: FEDM_ELL(0); break;
        case 1: FEDM_ELL(1); break;
        case 2: FEDM_ELL(2); break;
        case 3: FEDM_ELL(3); break;
        case 5: FEDM_ELL(5); break;
        default: FEDM_ELL(4); break;
    }
#undef FEDM_ELL
}

void ell_apply(Ctx &c, const EllMat &A, int mode, const double *x, const double *b, double *y,
               double omega, double *aux) {
    ell_launch(c, A, mode, x, b, y, omega, aux);
}

__global__ void jacobi_first_kernel(int n, const double *__restrict__ dinv,
                                    const double *__restrict__ b, double *__restrict__ x, double omega) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = omega * dinv[i] * b[i];
}

// y = Minv b for the dense coarsest operator (rows padded to ld = 256 k): one wave per row,
// 16-byte loads, two independent partial sums
__global__ __launch_bounds__(256) void dense_gemv_kernel(int n, int ld, const double *__restrict__ M,
                                                         const double *__restrict__ b, double *__restrict__ y) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= n) return;
    const double2 *Mr = reinterpret_cast<const double2 *>(M + (size_t)row * ld);
    double s0 = 0.0, s1 = 0.0;
    for (int j2 = lane; j2 * 2 < n; j2 += 64) {
        const double2 m = Mr[j2];
        const int j = j2 * 2;
        s0 += m.x * b[j];
        if (j + 1 < n) s1 += m.y * b[j + 1];
    }
    double s = s0 + s1;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) y[row] = s;
}

// place the rank's coarsest right-hand side into the (zeroed) global coarse vector
__global__ void coarse_place_kernel(int n_global, int offset, int n_local, const double *__restrict__ b,
                                    double *__restrict__ gb) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_global) return;
    const int k = i - offset;
    gb[i] = (k >= 0 && k < n_local) ? b[k] : 0.0;
}

// Buffers never swap on the host: with (nu-1) + nu Jacobi sweeps after the first one the
// iterate ping-pongs an odd number of times, so it starts in x2 and always ends in x.  The
// launch sequence is therefore static and is replayed as one hipGraph (Amg::run).
// phase 0: whole cycle.  Across GPUs everything below the finest level is global (Amg::global):
// phase 1 is the down leg up to the rank's segment of the global right-hand side (d_gb), the
// caller all-reduces d_gb, phase 2 is the replicated global cycle and the up leg.
void Amg::vcycle(Ctx &c, int l, int phase) {
    Level &L = levels[l];
    const bool down = phase != 2, up = phase != 1;
    if (l == (int)levels.size() - 1) {
        if (global) {
            if (down)
                hipLaunchKernelGGL(coarse_place_kernel, dim3((n_global + 255) / 256), dim3(256), 0, c.stream,
                                   n_global, g_offset, n_coarse, L.b, d_gb);
            if (up) {
                if (global->graph_exec && !c.capturing) hipGraphLaunch(global->graph_exec, c.stream);
                else global->vcycle(c, 0, 0);
                hipMemcpyAsync(L.x, global->levels[0].x + g_offset, sizeof(double) * n_coarse,
                               hipMemcpyDeviceToDevice, c.stream);
            }
        } else if (up && coarse_inv) {
            hipLaunchKernelGGL(dense_gemv_kernel, dim3((n_coarse + 3) / 4), dim3(256), 0, c.stream,
                               n_coarse, coarse_ld, coarse_inv, L.b, L.x);
        } else if (up) {  // neither a dense inverse nor a global hierarchy yet: no coarse correction
            hipMemsetAsync(L.x, 0, sizeof(double) * n_coarse, c.stream);
        }
        return;
    }
    if (L.composite) {
        if (down) ell_launch(c, L.C, 0, L.b, nullptr, levels[l + 1].b, 0.0);   // b_c = R (b - A w Dinv b)
        vcycle(c, l + 1, phase);
        if (up) ell_launch(c, L.GQ, 0, L.b, nullptr, L.x, 0.0);                // [b ; x_c] -> x
        return;
    }
    if (poly) {
        // finest level of a polynomial-smoother hierarchy (one GPU): leg down as one product
        // b_c = R (I - A S) b; leg up x = S b + P x_c as one product on [b ; x_c] (the next level's x
        // is the tail of this level's b), then the k sweeps backwards.  k buffer swaps must end in L.x.
        if (down) ell_launch(c, L.C, 0, L.b, nullptr, levels[l + 1].b, 0.0);
        vcycle(c, l + 1, phase);
        if (!up) return;
        const int k = (int)L.w.size();
        double *x = (k % 2 == 0) ? L.x : L.x2, *y = (k % 2 == 0) ? L.x2 : L.x;
        ell_launch(c, L.S, 0, L.b, nullptr, x, 0.0);  // [S | P] on [b ; x_c]
        if (l == 0 && k <= 4) {
            // one GPU: the k sweeps in one launch on tiles (fs_tiles.hip); weights in the order they are applied
            double wk[4];
            for (int i = 0; i < k; ++i) wk[i] = L.w[k - 1 - i];
            if (mg_tiles_sweeps(c, L.A, L.b, x, k, wk, out ? out : L.x, out ? out_stride : 1, out ? out_offset : 0))
                return;
        }
        for (int s_ = k - 1; s_ >= 0; --s_) {
            if (l == 0 && s_ == 0 && out) {
                ell_launch(c, L.A, 2, x, L.b, out, L.w[s_], nullptr, out_stride, out_offset);
                break;
            }
            ell_launch(c, L.A, 2, x, L.b, y, L.w[s_]);
            std::swap(x, y);
        }
        return;
    }
    const int np = L.A.n_rows_p;
    double *x = L.x2, *y = L.x;  // x: current iterate, y: the other buffer
    if (!pre_smooth) {
        // V(0,nu): x = 0, so the residual is b itself and the correction is P x_c
        if (down) ell_launch(c, L.R, 0, L.b, nullptr, levels[l + 1].b, 0.0);
        vcycle(c, l + 1, phase);
        if (!up) return;
        if (nu % 2 == 0) std::swap(x, y);                        // nu swaps must end in L.x
        ell_launch(c, L.P, 0, levels[l + 1].x, nullptr, x, 0.0);
        for (int s = 0; s < nu; ++s) {
            ell_launch(c, L.A, 2, x, L.b, y, omega);
            std::swap(x, y);
        }
        return;
    }
    // Global mode: the finest level is a truly distributed operator -- the ghost entries of the
    // right-hand side (before the first sweep, whose neighbours' values are formed from b) and of
    // the iterate (before the post-smoothing) come from their owners.
    // (deep halos: the right-hand side is exact on enough ghost layers for both smoothings -- no exchange)
    const bool exact0 = l == 0 && global && c.comm && !c.capturing && !deep_halo_active(c);
    // (when the slices of A are classified -- interior: no ghost column -- the exchange runs on the
    // communication stream while the interior slices are processed)
    static const bool mg_overlap_ok = [] {
        const char *e = std::getenv("FEDM_MG_HALO_OVERLAP");
        return !(e && e[0] == '0');
    }();
    const bool split0 = exact0 && mg_overlap_ok && n_interior0 > 0 && nu == 1;
    auto launch_A_halo = [&](int mode, double *halo_vec, const double *xx, double *yy, double *aux_,
                             int ystr, int yoffs) {
        comm_halo_begin(c);
        ell_launch(c, L.A, mode, xx, L.b, yy, omega, aux_, ystr, yoffs, nullptr, d_interior0, n_interior0);
        comm_halo_exchange_scalar(c, halo_vec);
        ell_launch(c, L.A, mode, xx, L.b, yy, omega, aux_, ystr, yoffs, nullptr, d_boundary0, n_boundary0);
    };
    if (exact0 && down && !split0) comm_halo_scalar(c, L.b);
    const bool dc = L.down_composite && nu == 1;
    if (dc) {
        if (down) ell_launch(c, L.C, 0, L.b, nullptr, levels[l + 1].b, 0.0);  // b_c = R (b - A w Dinv b)
    } else if (nu == 1) {
        if (down && split0) launch_A_halo(4, L.b, nullptr, L.r, x, 1, 0);
        else if (down) ell_launch(c, L.A, 4, nullptr, L.b, L.r, omega, x);    // x = w Dinv b;  r = b - A x
    } else {
        if (down)
            hipLaunchKernelGGL(jacobi_first_kernel, dim3((np + 255) / 256), dim3(256), 0, c.stream, np,
                               L.A.dinv, L.b, x, omega);
        for (int s = 1; s < nu; ++s) {
            if (down) ell_launch(c, L.A, 2, x, L.b, y, omega);
            std::swap(x, y);
        }
        if (down) ell_launch(c, L.A, 1, x, L.b, L.r, 0.0);               // r = b - A x
    }
    if (down && !dc) ell_launch(c, L.R, 0, L.r, nullptr, levels[l + 1].b, 0.0);  // b_c = R r
    vcycle(c, l + 1, phase);
    if (!up) return;
    if (dc) ell_launch(c, L.P, 5, levels[l + 1].x, L.b, x, omega, nullptr, 1, 0, L.A.dinv);  // x = w Dinv b + P x_c
    else ell_launch(c, L.P, 3, levels[l + 1].x, nullptr, x, 0.0);    // x += P x_c
    for (int s = 0; s < nu; ++s) {
        if (exact0 && !split0) comm_halo_scalar(c, x);
        if (l == 0 && s == nu - 1 && out) {
            // the cycle's result goes straight into the potential component of the caller's
            // interleaved vector (no scatter kernel afterwards)
            if (split0) launch_A_halo(2, x, x, out, nullptr, out_stride, out_offset);
            else ell_launch(c, L.A, 2, x, L.b, out, omega, nullptr, out_stride, out_offset);
            break;
        }
        if (split0) {
            launch_A_halo(2, x, x, y, nullptr, 1, 0);
            std::swap(x, y);
            continue;
        }
        ell_launch(c, L.A, 2, x, L.b, y, omega);
        std::swap(x, y);
    }
    // (2 nu - 1) swaps from x2 -> the result is in L.x
}

int Amg::capture(Ctx &c) {
    if (graph_exec) {
        hipGraphExecDestroy(graph_exec);
        graph_exec = nullptr;
    }
    hipGraph_t graph = nullptr;
    if (global || !coarse_inv) return 0;  // the cycle contains an all-reduce: plain launches in Amg::run
    if (hipStreamBeginCapture(c.stream, hipStreamCaptureModeThreadLocal) != hipSuccess) return -1;
    vcycle(c, 0, 0);
    if (hipStreamEndCapture(c.stream, &graph) != hipSuccess || !graph) return -1;
    const hipError_t e = hipGraphInstantiate(&graph_exec, graph, nullptr, nullptr, 0);
    hipGraphDestroy(graph);
    return e == hipSuccess ? 0 : -1;
}

// sum of the ranks' contributions to the global level-1 right-hand side
// (it feeds a preconditioner: single precision on the wire halves the largest message of a Krylov step;
// FEDM_MG_ALLREDUCE_F32=0: double precision)
void Amg::allreduce_level1(Ctx &c) {
    static const bool f32 = [] {
        const char *e = std::getenv("FEDM_MG_ALLREDUCE_F32");
        return !(e && e[0] == '0');
    }();
    if (f32) comm_allreduce_f32_payload(c, d_gb, n_global);
    else comm_allreduce(c, d_gb, n_global);
}

void Amg::run(Ctx &c) {
    if (c.capturing) {  // part of a whole-iteration graph: its kernels become nodes of that graph
        vcycle(c, 0, 0);
        return;
    }
    prof_begin(c, 3);
    if (global) {
        vcycle(c, 0, 1);
        allreduce_level1(c);
        vcycle(c, 0, 2);
    } else if (graph_exec) {
        hipGraphLaunch(graph_exec, c.stream);
    } else {
        vcycle(c, 0, 0);
    }
    prof_end(c);
}

void Amg::release() {
    for (auto &L : levels) {
        L.A.release();
        L.P.release();
        L.R.release();
        L.C.release();
        L.GQ.release();
        L.S.release();
        if (L.x_is_alias) L.x = nullptr;
        for (double *p : {L.x, L.x2, L.b, L.r})
            if (p) hipFree(p);
    }
    levels.clear();
    if (d_interior0) hipFree(d_interior0);
    if (d_boundary0) hipFree(d_boundary0);
    d_interior0 = d_boundary0 = nullptr;
    n_interior0 = n_boundary0 = 0;
    if (graph_exec) hipGraphExecDestroy(graph_exec);
    graph_exec = nullptr;
    if (coarse_inv) hipFree(coarse_inv);
    coarse_inv = nullptr;
    if (global) {
        global->release();
        delete global;
        global = nullptr;
    }
    d_gb = nullptr;
    n_coarse = n_global = 0;
}

// ---- field split -----------------------------------------------------------------------------
// z_u = omega * Duu^-1 (alpha t_u) per vertex; b0 = alpha t_phi.  compact32: the species iterate is
// the compact single-precision vector the sweeps work on ([vertex][NS] floats), else the
// interleaved fp64 vector itself (no sweeps to follow)
template <int NS>
__global__ void fs_species_kernel(int nvp, const double *__restrict__ dinv_uu,
                                  const double *__restrict__ t, double *__restrict__ z,
                                  double *__restrict__ b0, double alpha, double omega, int compact32) {
    constexpr int NEQ = NS + 1;
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nvp) return;
    const int slice = v >> 6, lane = v & 63;
    const double *dp = dinv_uu + (size_t)slice * NS * NS * SLICE + lane;
    double tv[NEQ];
#pragma unroll
    for (int s = 0; s < NEQ; ++s) tv[s] = alpha * t[(size_t)v * NEQ + s];
#pragma unroll
    for (int r = 0; r < NS; ++r) {
        double acc = 0.0;
#pragma unroll
        for (int cidx = 0; cidx < NS; ++cidx) acc += dp[(size_t)(r * NS + cidx) * SLICE] * tv[cidx];
        if (compact32) reinterpret_cast<float *>(z)[(size_t)v * NS + r] = (float)(omega * acc);
        else z[(size_t)v * NEQ + r] = omega * acc;
    }
    if (!compact32) z[(size_t)v * NEQ + NS] = 0.0;
    b0[v] = tv[NS];
}

// one more damped block-Jacobi sweep on the species block, with g = Duu^-1 (alpha t_u) from the
// first stage and S = Duu^-1 J_uu:
//   z_out_u = zs z_in_u + omega * (g_u - S (zs z_in_u))
// (zs = the first stage's weight in the first sweep, whose z_in is g itself; 1 afterwards).
// S is streamed from a half-precision copy (species_planes_kernel): scaled by the block diagonal
// its entries are O(1), and a preconditioner's matrix needs no more than that -- a quarter of the
// fp64 bytes, and neither t nor Duu^-1 is read again.  Vectors and accumulation stay fp64.
// Block Jacobi alone leaves a mass-matrix-like operator with eigenvalues in ~[0.5, 2]; a few
// damped sweeps cut the outer GMRES iterations from 8 to 5 per Newton step.
// The iterate is a compact single-precision vector between the sweeps ([vertex][NS] floats: a
// third of the bytes of the interleaved fp64 vector; accumulation in fp64); the LAST sweep writes
// the interleaved fp64 vector that the Krylov method sees.
// ZS: bit (r * NS + c) marks a plane of S that is structurally zero (the Jacobian plane is, and the
// block diagonal is then triangular, so the scaling keeps the zero): not loaded.
template <int NS, bool LAST, unsigned ZS = 0u>
__global__ __launch_bounds__(256) void fs_species_sweep_kernel(
    int n_slices, const int *__restrict__ boff, const int *__restrict__ colidx,
    const _Float16 *__restrict__ s16, const float *__restrict__ g, const float *__restrict__ zin,
    float *__restrict__ zout32, double *__restrict__ zout, double zs, double omega,
    const int *__restrict__ slice_list, const double *__restrict__ x0,
    const float *__restrict__ cpl32 = nullptr, double *__restrict__ b0 = nullptr) {
    constexpr int NEQ = NS + 1, PL = NS * NS;
    const int wave_id = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (wave_id >= n_slices) return;   // n_slices: number of slices this launch covers
    const int slice = slice_list ? slice_list[wave_id] : wave_id;
    // this row's own operands first: their latency hides behind the gather loop
    const size_t v = (size_t)slice * SLICE + lane;
    // (single precision with the half-precision entry as an operand of the fused multiply-add, the very operations of
    // fs_tile_sweeps_kernel, fs_tiles.hip: the two give identical bits; the iterate between sweeps is single
    // precision anyway and a row has seven to a dozen entries)
    float gv[NS], zv[NS];
#pragma unroll
    for (int r = 0; r < NS; ++r) {
        gv[r] = g[v * NS + r];
        zv[r] = zin[v * NS + r];
    }
    float acc[NS];
    double accp = 0.0;
#pragma unroll
    for (int r = 0; r < NS; ++r) acc[r] = 0.f;
    // LAST with cpl32: the coupling product of the lower-triangular split rides on this sweep's gather
    // of the neighbours' iterate, b_phi -= J_phi,u (zs z_in) -- the iterate BEFORE this last sweep
    // (a preconditioner may lag; the separate pass over the potential row's planes and z is saved)
    const bool couple = LAST && cpl32 != nullptr;
    const double own_b0 = couple ? b0[v] : 0.0;
    const int bb0 = boff[slice], bb1 = boff[slice + 1];
    for (int bc = bb0; bc < bb1; ++bc) {
        const int col = colidx[(size_t)bc * SLICE + lane];
        float zj[NS];
#pragma unroll
        for (int cidx = 0; cidx < NS; ++cidx) zj[cidx] = zin[(size_t)col * NS + cidx];
        const _Float16 *vp = s16 + ((size_t)bc * SLICE + lane) * PL;   // the PL planes of an entry side by side
#pragma unroll
        for (int r = 0; r < NS; ++r)
#pragma unroll
            for (int cidx = 0; cidx < NS; ++cidx)
                if (!((ZS >> (r * NS + cidx)) & 1u)) acc[r] = __builtin_fmaf((float)vp[r * NS + cidx], zj[cidx], acc[r]);
        if (couple) {
            const float *cp = cpl32 + (size_t)bc * NS * SLICE + lane;
#pragma unroll
            for (int cidx = 0; cidx < NS; ++cidx)
                accp = __builtin_fma((double)cp[(size_t)cidx * SLICE], (double)zj[cidx], accp);
        }
    }
    if (couple) b0[v] = __builtin_fma(-zs, accp, own_b0);
    const float zsf = (float)zs, omf = (float)omega;
#pragma unroll
    for (int r = 0; r < NS; ++r) {
        const float zn = __builtin_fmaf(omf, __builtin_fmaf(-zsf, acc[r], gv[r]), zsf * zv[r]);
        if (LAST) zout[v * NEQ + r] = (double)zn;
        else zout32[v * NS + r] = zn;
    }
    // full-line stores; the potential entry is the V-cycle's result when that ran first (upper-
    // triangular order), else it is set by the V-cycle that follows
    if (LAST) zout[v * NEQ + NS] = x0 ? x0[v] : 0.0;
}

// Upper-triangular order, first stage: the potential correction x0 = z_phi is known (the V-cycle ran
// first), the species block sees t_u - J_u,phi z_phi:
//   g = Duu^-1 (alpha t_u - J_u,phi x0)
// (compact32: into the single-precision vector the sweeps work on; else z itself, potential entry
// included -- plain block Jacobi, no sweeps follow).  J_u,phi comes from the fp32 planes.
template <int NS>
__global__ __launch_bounds__(256) void fs_species_upper_kernel(
    int n_slices, const int *__restrict__ boff, const int *__restrict__ colidx,
    const float *__restrict__ val32, const double *__restrict__ dinv_uu, const double *__restrict__ t,
    const double *__restrict__ x0, double *__restrict__ z, double alpha, int compact32,
    const int *__restrict__ slice_list) {
    constexpr int NEQ = NS + 1;
    const int wave_id = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (wave_id >= n_slices) return;
    const int slice = slice_list ? slice_list[wave_id] : wave_id;
    const size_t v = (size_t)slice * SLICE + lane;
    double tv[NS], acc[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        tv[s] = alpha * t[v * NEQ + s];  // requested before the gather loop
        acc[s] = 0.0;
    }
    const double own = x0[v];
    const int b0 = boff[slice], b1 = boff[slice + 1];
    for (int bc = b0; bc < b1; ++bc) {
        const double xp = x0[colidx[(size_t)bc * SLICE + lane]];
        const float *vp = val32 + (size_t)bc * NS * SLICE + lane;
#pragma unroll
        for (int s = 0; s < NS; ++s) acc[s] += (double)vp[(size_t)s * SLICE] * xp;
    }
    const double *dp = dinv_uu + (size_t)slice * NS * NS * SLICE + lane;
#pragma unroll
    for (int r = 0; r < NS; ++r) {
        double g = 0.0;
#pragma unroll
        for (int cidx = 0; cidx < NS; ++cidx) g += dp[(size_t)(r * NS + cidx) * SLICE] * (tv[cidx] - acc[cidx]);
        if (compact32) reinterpret_cast<float *>(z)[v * NS + r] = (float)g;
        else z[v * NEQ + r] = g;
    }
    if (!compact32) z[v * NEQ + NS] = own;
}

// b0 = alpha t_phi: right-hand side of the V-cycle when it runs first
__global__ void fs_gather_kernel(int nvp, int neq, const double *__restrict__ t, double *__restrict__ b0,
                                 double alpha) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < nvp) b0[v] = alpha * t[(size_t)v * neq + neq - 1];
}

// b0 -= J_phi,u z_u   (reads only the n_species value planes of the potential row)
template <int NS>
__global__ __launch_bounds__(256) void fs_coupling_kernel(int n_slices, const int *__restrict__ boff,
                                                          const int *__restrict__ colidx,
                                                          const float *__restrict__ val32,
                                                          const double *__restrict__ z,
                                                          double *__restrict__ b0) {
    constexpr int NEQ = NS + 1;
    const int slice = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (slice >= n_slices) return;
    const double own = b0[(size_t)slice * SLICE + lane];  // requested before the gather loop
    double acc = 0.0;
    const int bb0 = boff[slice], bb1 = boff[slice + 1];
    for (int bc = bb0; bc < bb1; ++bc) {
        const int col = colidx[(size_t)bc * SLICE + lane];
        const float *vp = val32 + (size_t)bc * NS * SLICE + lane;
#pragma unroll
        for (int s = 0; s < NS; ++s) acc += (double)vp[(size_t)s * SLICE] * z[(size_t)col * NEQ + s];
    }
    b0[(size_t)slice * SLICE + lane] = own - acc;
}

template <int NS>
__global__ void fs_scatter_kernel(int nvp, const double *__restrict__ x0, double *__restrict__ z) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < nvp) z[(size_t)v * (NS + 1) + NS] = x0[v];
}


// With sweeps, the first stage leaves g = Duu^-1 (alpha t_u) (weight 1) in its own vector, which
// every sweep reads; the iterate then ping-pongs between z and the scratch vector so that the last
// sweep writes z.  Without sweeps the first stage writes z itself.
static double *fs_first_target(Ctx &c, double *z) { return c.fs_sweeps > 1 ? c.d_fs_g : z; }
static bool fs_first_compact(const Ctx &c) { return c.fs_sweeps > 1; }

// stages after the first: remaining species sweeps, coupling, V-cycle on the potential block
template <int NS>
static void fs_finish_t(Ctx &c, Amg &amg, double *z, bool scatter = true, bool with_cycle = true,
                        bool upper = false) {
    const dim3 gv((c.nvp + 255) / 256), bv(256);
    const dim3 gs((c.pat.n_slices + 3) / 4);
    const int n_sweeps = (c.fs_sweeps < 1 ? 1 : c.fs_sweeps) - 1;
    // g in c.d_fs_g, the iterate ping-pongs between the two halves of c.d_fs (floats), the last
    // sweep writes z
    const float *g32 = reinterpret_cast<const float *>(c.d_fs_g), *in = g32;
    float *ping[2] = {reinterpret_cast<float *>(c.d_fs), reinterpret_cast<float *>(c.d_fs) + (size_t)c.nvp * NS};
    // several GPUs, c.fs_halo: the species iterate of the ghost vertices comes from their owners
    // before every sweep (otherwise the sweeps see zeros there: block Jacobi over the ranks)
    // (across GPUs the field split is always on the right -- flexible GMRES --, so the applications
    // need not be one fixed operator; inside a captured step there is no exchange)
    const bool halo = c.comm && c.fs_halo && !deep_halo_active(c) && !c.capturing && n_sweeps > 0;
    // ... and the exchange overlaps with the sweep's interior slices (those without ghost columns):
    // mark the iterate complete, sweep the interior, exchange on the communication stream, wait,
    // sweep the boundary slices -- as the Krylov halo does with the Jacobian product
    // Off by default: a sweep's interior slices take 7 us, the two event-fenced hops between the streams
    // cost 20 us (measured with the several-GPU solver on one rank over RCCL, tools/one_rank_rccl_overhead.py:
    // 2.49 -> 1.88 ms per step without them) -- more than the exchange they could hide behind those 7 us.
    // FEDM_FS_HALO_OVERLAP=1 switches the overlap on.
    static const bool overlap_ok = [] {
        const char *e = std::getenv("FEDM_FS_HALO_OVERLAP");
        return e && e[0] == '1';
    }();
    const bool overlap = halo && overlap_ok && c.comm->n_interior > 0;
    // lower-triangular order with sweeps: the coupling product is part of the last sweep (lagged by it)
    const bool lagged = !upper && c.fs_lagged_coupling && n_sweeps > 0;
    // two species: the off-diagonal planes (0,1) / (1,0) of S may be structurally zero
    unsigned zs_mask = 0u;
    if (NS == 2) zs_mask = ((c.zero_plane_mask >> 1) & 1u) << 1 | ((c.zero_plane_mask >> 3) & 1u) << 2;
    auto sweep = [&](bool last, const float *in_, float *out32, double zs, double w, const int *list, int n) {
        if (n == 0) return;
        const dim3 g((n + 3) / 4);
        const double *x0 = upper ? (const double *)amg.levels[0].x : (const double *)nullptr;
#define FEDM_SWEEP(Z)                                                                                          \
    do {                                                                                                       \
        if (last)                                                                                              \
            hipLaunchKernelGGL((fs_species_sweep_kernel<NS, true, Z>), g, dim3(256), 0, c.stream, n,           \
                               c.d_slice_boff, c.d_colidx, c.d_s16, g32, in_, (float *)nullptr, z, zs, w, list, x0, \
                               lagged ? (const float *)c.d_val32 : (const float *)nullptr,                     \
                               lagged ? amg.levels[0].b : (double *)nullptr);                                  \
        else                                                                                                   \
            hipLaunchKernelGGL((fs_species_sweep_kernel<NS, false, Z>), g, dim3(256), 0, c.stream, n,          \
                               c.d_slice_boff, c.d_colidx, c.d_s16, g32, in_, out32, (double *)nullptr, zs, w, \
                               list, (const double *)nullptr);                                                 \
    } while (0)
        if (NS == 2 && zs_mask == 2u) FEDM_SWEEP(2u);
        else if (NS == 2 && zs_mask == 4u) FEDM_SWEEP(4u);
        else if (NS == 2 && zs_mask == 6u) FEDM_SWEEP(6u);
        else FEDM_SWEEP(0u);
#undef FEDM_SWEEP
    };
    // one GPU: all sweeps in one launch (tiles of slices with their vertex layers in LDS, fs_tiles.hip)
    const bool tiled = !halo && (!c.comm || deep_halo_active(c)) && n_sweeps > 0 &&
                       fs_tiles_sweeps(c, n_sweeps, zs_mask, g32, ping[0], ping[1], z,
                                       upper ? (const double *)amg.levels[0].x : (const double *)nullptr,
                                       lagged ? (const float *)c.d_val32 : (const float *)nullptr,
                                       lagged ? amg.levels[0].b : (double *)nullptr);
    for (int s = 1; s <= n_sweeps && !tiled; ++s) {
        const double zs = s == 1 ? c.fs_w[0] : 1.0;
        const bool last = s == n_sweeps;
        float *out = last ? nullptr : ping[s & 1];
        if (overlap) {
            comm_halo_begin(c);
            sweep(last, in, out, zs, c.fs_w[s], c.comm->d_interior, c.comm->n_interior);
            comm_halo_exchange_f32(c, const_cast<float *>(in), NS);
            sweep(last, in, out, zs, c.fs_w[s], c.comm->d_boundary, c.comm->n_boundary);
        } else {
            if (halo) comm_halo_f32(c, const_cast<float *>(in), NS);
            sweep(last, in, out, zs, c.fs_w[s], nullptr, c.pat.n_slices);
        }
        if (!last) in = out;
    }
    if (upper) return;  // the potential came first
    if (!lagged)
        hipLaunchKernelGGL(fs_coupling_kernel<NS>, gs, dim3(256), 0, c.stream, c.pat.n_slices,
                           c.d_slice_boff, c.d_colidx, c.d_val32, z, amg.levels[0].b);
    if (!with_cycle) return;  // the caller runs the V-cycle (collectives inside it) and scatters
    amg.run(c);
    if (scatter) hipLaunchKernelGGL(fs_scatter_kernel<NS>, gv, bv, 0, c.stream, c.nvp, amg.levels[0].x, z);
}

bool deep_halo_active(const Ctx &c) {
    return c.comm && c.deep_halo && c.halo_depth > 1 && c.halo_depth >= c.fs_sweeps + 2 && !c.fs_upper && c.amg &&
           c.amg->global && c.amg->pre_smooth && c.amg->nu == 1;
}

bool fieldsplit_upper(const Ctx &c) { return c.fs_upper && (c.right_precond || c.comm) && c.amg && c.poisson; }

// upper-triangular order, part 1: z_phi = V-cycle(alpha t_phi) into amg.levels[0].x (ghost entries
// from their owners across GPUs: the coupling product reads them)
void fieldsplit_upper_potential(Ctx &c, Amg &amg, const double *t, double alpha) {
    hipLaunchKernelGGL(fs_gather_kernel, dim3((c.nvp + 255) / 256), dim3(256), 0, c.stream, c.nvp, c.neq, t,
                       amg.levels[0].b, alpha);
    amg.run(c);
    if (c.comm && !c.capturing) comm_halo_scalar(c, amg.levels[0].x);
}

// part 2: species sweeps on alpha t_u - J_u,phi z_phi; the last one writes z, potential entry included
template <int NS>
static void fs_upper_species_t(Ctx &c, Amg &amg, const double *t, double *z, double alpha) {
    const dim3 gs((c.pat.n_slices + 3) / 4);
    hipLaunchKernelGGL(fs_species_upper_kernel<NS>, gs, dim3(256), 0, c.stream, c.pat.n_slices, c.d_slice_boff,
                       c.d_colidx, c.d_val32, c.d_dinv, t, amg.levels[0].x, fs_first_target(c, z), alpha,
                       fs_first_compact(c) ? 1 : 0, (const int *)nullptr);
    if (fs_first_compact(c)) fs_finish_t<NS>(c, amg, z, false, false, true);
}

void fieldsplit_upper_species(Ctx &c, Amg &amg, const double *t, double *z, double alpha) {
    switch (c.ns) {
        case 1: fs_upper_species_t<1>(c, amg, t, z, alpha); break;
        case 2: fs_upper_species_t<2>(c, amg, t, z, alpha); break;
        case 3: fs_upper_species_t<3>(c, amg, t, z, alpha); break;
        case 4: fs_upper_species_t<4>(c, amg, t, z, alpha); break;
        case 5: fs_upper_species_t<5>(c, amg, t, z, alpha); break;
    }
}

template <int NS>
static void fs_apply_t(Ctx &c, Amg &amg, const double *t, double *z, double alpha, bool scatter, bool with_cycle) {
    const dim3 gv((c.nvp + 255) / 256), bv(256);
    // (GMRES on one GPU: the kernel that completed t has formed this stage already)
    if (!(c.fs_first_by_producer && fs_first_compact(c) && alpha == 1.0))
        hipLaunchKernelGGL(fs_species_kernel<NS>, gv, bv, 0, c.stream, c.nvp, c.d_dinv, t, fs_first_target(c, z),
                           amg.levels[0].b, alpha, 1.0, fs_first_compact(c) ? 1 : 0);
    fs_finish_t<NS>(c, amg, z, scatter, with_cycle);
}

void fieldsplit_first_stage(Ctx &c, Amg &amg, const double *t) {
    const dim3 gv((c.nvp + 255) / 256), bv(256);
    double *g = c.d_fs_g;
#define FEDM_FIRST(NS_)                                                                                    \
    hipLaunchKernelGGL(fs_species_kernel<NS_>, gv, bv, 0, c.stream, c.nvp, c.d_dinv, t, g, amg.levels[0].b, 1.0, \
                       1.0, 1)
    switch (c.ns) {
        case 1: FEDM_FIRST(1); break;
        case 2: FEDM_FIRST(2); break;
        case 3: FEDM_FIRST(3); break;
        case 4: FEDM_FIRST(4); break;
        case 5: FEDM_FIRST(5); break;
    }
#undef FEDM_FIRST
}

void fieldsplit_scatter(Ctx &c, Amg &amg, double *z) {
    const dim3 gv((c.nvp + 255) / 256), bv(256);
    switch (c.ns) {
        case 1: hipLaunchKernelGGL(fs_scatter_kernel<1>, gv, bv, 0, c.stream, c.nvp, amg.levels[0].x, z); break;
        case 2: hipLaunchKernelGGL(fs_scatter_kernel<2>, gv, bv, 0, c.stream, c.nvp, amg.levels[0].x, z); break;
        case 3: hipLaunchKernelGGL(fs_scatter_kernel<3>, gv, bv, 0, c.stream, c.nvp, amg.levels[0].x, z); break;
        case 4: hipLaunchKernelGGL(fs_scatter_kernel<4>, gv, bv, 0, c.stream, c.nvp, amg.levels[0].x, z); break;
        case 5: hipLaunchKernelGGL(fs_scatter_kernel<5>, gv, bv, 0, c.stream, c.nvp, amg.levels[0].x, z); break;
    }
}

void fieldsplit_apply(Ctx &c, Amg &amg, const double *t, double *z, double alpha, bool scatter, bool with_cycle) {
    if (fieldsplit_upper(c)) {  // (scatter / with_cycle belong to the lower-triangular order)
        fieldsplit_upper_potential(c, amg, t, alpha);
        fieldsplit_upper_species(c, amg, t, z, alpha);
        return;
    }
    switch (c.ns) {
        case 1: fs_apply_t<1>(c, amg, t, z, alpha, scatter, with_cycle); break;
        case 2: fs_apply_t<2>(c, amg, t, z, alpha, scatter, with_cycle); break;
        case 3: fs_apply_t<3>(c, amg, t, z, alpha, scatter, with_cycle); break;
        case 4: fs_apply_t<4>(c, amg, t, z, alpha, scatter, with_cycle); break;
        case 5: fs_apply_t<5>(c, amg, t, z, alpha, scatter, with_cycle); break;
    }
}

// z = Minv (J v): the SpMV's epilogue is the first stage (t = J v is kept for the sweeps)
void fieldsplit_apply_operator(Ctx &c, Amg &amg, const double *v, double *t, double *z, bool scatter) {
    prof_begin(c, 1);
    launch_spmv_fieldsplit(c, v, t, fs_first_target(c, z), amg.levels[0].b, 1.0, nullptr, 0, fs_first_compact(c));
    prof_end(c);
    switch (c.ns) {
        case 1: fs_finish_t<1>(c, amg, z, scatter); break;
        case 2: fs_finish_t<2>(c, amg, z, scatter); break;
        case 3: fs_finish_t<3>(c, amg, z, scatter); break;
        case 4: fs_finish_t<4>(c, amg, z, scatter); break;
        case 5: fs_finish_t<5>(c, amg, z, scatter); break;
    }
}

void fieldsplit_apply_operator_part(Ctx &c, Amg &amg, const double *v, double *t, double *z, bool scatter,
                                    int part, const int *slices, int n_slices) {
    launch_spmv_fieldsplit(c, v, t, fs_first_target(c, z), amg.levels[0].b, 1.0, slices, n_slices, fs_first_compact(c));
    if (part == 0) return;
    const bool cyc = part == 1;  // part 2: stop after the coupling product (amg.levels[0].b is ready)
    switch (c.ns) {
        case 1: fs_finish_t<1>(c, amg, z, scatter, cyc); break;
        case 2: fs_finish_t<2>(c, amg, z, scatter, cyc); break;
        case 3: fs_finish_t<3>(c, amg, z, scatter, cyc); break;
        case 4: fs_finish_t<4>(c, amg, z, scatter, cyc); break;
        case 5: fs_finish_t<5>(c, amg, z, scatter, cyc); break;
    }
}

// Set-up of the field split after a Jacobian assembly, one wave per slice: the inverse of the
// species part of every diagonal block (D_uu^-1, kept in c.d_dinv for the first stage), and the
// copies of the species columns the preconditioner streams: the species rows scaled by it,
// S = Duu^-1 J_uu, in half precision (the sweeps), and the potential row J_phi,u in single
// precision (the coupling product).
template <int NS>
__global__ __launch_bounds__(256) void species_planes_kernel(int n_slices, const int *__restrict__ boff,
                                                             const double *__restrict__ val,
                                                             const uint32_t *__restrict__ diag_slot,
                                                             double *__restrict__ dinv_uu,
                                                             _Float16 *__restrict__ s16, float *__restrict__ val32,
                                                             int upper, unsigned zs) {
    // zs: bit (r * NS + c) marks a species plane of the Jacobian that is structurally zero (the sweeps'
    // ZS): neither read here nor written to the half-precision copy, which nobody reads there
    // A workgroup per slice, its four waves share the slice's block columns (each forms the 2x2 .. 5x5 inverse from four
    // .. 25 loads of its own): with a wave per slice the eight or so dependent iterations of a wave kept too few bytes
    // in flight for a kernel that only streams (r4: 31 -> see DESIGN.md section 3).
    constexpr int NEQ = NS + 1, NEQ2 = NEQ * NEQ;
    const int slice = blockIdx.x, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    const int lane = threadIdx.x & 63;
    if (slice >= n_slices) return;
    double d[NS][NS];
    {
        const uint32_t ds = diag_slot[(size_t)slice * SLICE + lane];
        double A[NS][NS];
#pragma unroll
        for (int r = 0; r < NS; ++r)
#pragma unroll
            for (int cidx = 0; cidx < NS; ++cidx)
                A[r][cidx] = val[((size_t)(ds >> 6) * NEQ2 + r * NEQ + cidx) * SLICE + (ds & 63)];
        invert_species_block<NS>(A, d);
        if (wave == 0) {
#pragma unroll
            for (int e = 0; e < NS * NS; ++e)
                dinv_uu[((size_t)slice * NS * NS + e) * SLICE + lane] = d[e / NS][e % NS];
        }
    }
    const int b0 = boff[slice], b1 = boff[slice + 1];
    for (int bc = b0 + wave; bc < b1; bc += n_waves) {
        const double *vp = val + (size_t)bc * NEQ2 * SLICE + lane;
        double J[NS][NS];
#pragma unroll
        for (int r = 0; r < NS; ++r)
#pragma unroll
            for (int cidx = 0; cidx < NS; ++cidx)
                J[r][cidx] = ((zs >> (r * NS + cidx)) & 1u) ? 0.0 : vp[(size_t)(r * NEQ + cidx) * SLICE];
        _Float16 row16[NS * NS];
        species_plane_entry<NS>(d, J, zs, row16);
        // the planes of an entry side by side, [(bc * 64 + lane) * NS^2 + plane]: one 8-byte word per entry and lane
        // for two species, which the sweeps load as such
        _Float16 *dst = s16 + ((size_t)bc * SLICE + lane) * (NS * NS);
#pragma unroll
        for (int e = 0; e < NS * NS; ++e) dst[e] = row16[e];
#pragma unroll
        for (int cidx = 0; cidx < NS; ++cidx)
            val32[((size_t)bc * NS + cidx) * SLICE + lane] =
                (float)vp[(size_t)(upper ? cidx * NEQ + NS : NS * NEQ + cidx) * SLICE];
    }
}

// The same for listed vertices only: the rows that the boundary facets and the Dirichlet values touch AFTER the
// volume assembly, when that has formed the planes of all rows itself (assemble3.hip, Ctx::planes_fused).
template <int NS>
__global__ void species_planes_rows_kernel(int n_rows, const int *__restrict__ rows, const int *__restrict__ boff,
                                           const double *__restrict__ val, const uint32_t *__restrict__ diag_slot,
                                           double *__restrict__ dinv_uu, _Float16 *__restrict__ s16,
                                           float *__restrict__ val32, int upper, unsigned zs) {
    constexpr int NEQ = NS + 1, NEQ2 = NEQ * NEQ;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_rows) return;
    const int vtx = rows[t], slice = vtx >> 6, lane = vtx & 63;
    double d[NS][NS];
    {
        const uint32_t ds = diag_slot[vtx];
        double A[NS][NS];
#pragma unroll
        for (int r = 0; r < NS; ++r)
#pragma unroll
            for (int cidx = 0; cidx < NS; ++cidx)
                A[r][cidx] = val[((size_t)(ds >> 6) * NEQ2 + r * NEQ + cidx) * SLICE + (ds & 63)];
        invert_species_block<NS>(A, d);
#pragma unroll
        for (int e = 0; e < NS * NS; ++e)
            dinv_uu[((size_t)slice * NS * NS + e) * SLICE + lane] = d[e / NS][e % NS];
    }
    for (int bc = boff[slice]; bc < boff[slice + 1]; ++bc) {
        const double *vp = val + (size_t)bc * NEQ2 * SLICE + lane;
        double J[NS][NS];
#pragma unroll
        for (int r = 0; r < NS; ++r)
#pragma unroll
            for (int cidx = 0; cidx < NS; ++cidx)
                J[r][cidx] = ((zs >> (r * NS + cidx)) & 1u) ? 0.0 : vp[(size_t)(r * NEQ + cidx) * SLICE];
        _Float16 row16[NS * NS];
        species_plane_entry<NS>(d, J, zs, row16);
        _Float16 *dst = s16 + ((size_t)bc * SLICE + lane) * (NS * NS);
#pragma unroll
        for (int e = 0; e < NS * NS; ++e) dst[e] = row16[e];
#pragma unroll
        for (int cidx = 0; cidx < NS; ++cidx)
            val32[((size_t)bc * NS + cidx) * SLICE + lane] =
                (float)vp[(size_t)(upper ? cidx * NEQ + NS : NS * NEQ + cidx) * SLICE];
    }
}

// zs of the species planes kernels: the same mask the sweeps are compiled for (fs_finish_t): two species, a zero
// off-diagonal plane
unsigned fieldsplit_zero_species_planes(const Ctx &c) {
    if (c.ns != 2) return 0u;
    return ((c.zero_plane_mask >> 1) & 1u) << 1 | ((c.zero_plane_mask >> 3) & 1u) << 2;
}

void fieldsplit_setup(Ctx &c) {
    const dim3 b(256);
    if (c.fs_sweeps > 1 || c.fs_main_sweeps > 1 || c.fs_alt_sweeps > 1) fs_tiles_prepare(c);
    c.planes_last_fused = c.planes_fused;
    if (c.planes_fused) {
        // the volume assembly has written the planes of every row from its accumulators; what is left are the rows
        // changed behind it
        c.planes_fused = false;
        if (c.n_planes_rows > 0)
            hipLaunchKernelGGL(species_planes_rows_kernel<2>, dim3((c.n_planes_rows + 127) / 128), dim3(128), 0, c.stream,
                               c.n_planes_rows, c.d_planes_rows, c.d_slice_boff, c.d_val, c.d_diag_slot, c.d_dinv, c.d_s16,
                               c.d_val32, fieldsplit_upper(c) ? 1 : 0, fieldsplit_zero_species_planes(c));
        return;
    }
    const size_t n_entries = (size_t)c.pat.total_bc * SLICE;
    if (!c.d_val32 && (hipMalloc((void **)&c.d_val32, sizeof(float) * n_entries * c.ns) != hipSuccess ||
                       hipMalloc((void **)&c.d_s16, sizeof(_Float16) * n_entries * c.ns * c.ns) != hipSuccess)) {
        set_error("hipMalloc of the preconditioner's species planes failed");
        if (c.d_val32) hipFree(c.d_val32);
        c.d_val32 = nullptr;
        return;
    }
    const dim3 gs(c.pat.n_slices);
    const unsigned zs = fieldsplit_zero_species_planes(c);
#define FEDM_PLANES(NS_)                                                                                   \
    hipLaunchKernelGGL(species_planes_kernel<NS_>, gs, b, 0, c.stream, c.pat.n_slices, c.d_slice_boff, c.d_val, \
                       c.d_diag_slot, c.d_dinv, c.d_s16, c.d_val32, fieldsplit_upper(c) ? 1 : 0, zs)
    switch (c.ns) {
        case 1: FEDM_PLANES(1); break;
        case 2: FEDM_PLANES(2); break;
        case 3: FEDM_PLANES(3); break;
        case 4: FEDM_PLANES(4); break;
        case 5: FEDM_PLANES(5); break;
    }
#undef FEDM_PLANES
}

// Test hook (fedm_debug_species_planes_check): the planes as they stand -- formed by the assembly itself plus the
// listed rows, or by the separate pass -- against that separate pass over the matrix as it stands, run into scratch
// arrays.  out = {max |d dinv| / max |dinv|, max |d s16|, max |d val32| / max |val32|, was the last set-up the
// fused one}.
int fieldsplit_planes_check(Ctx &c, double *out, bool last_fused) {
    if (!c.d_s16 || !c.d_val32 || !c.d_dinv || c.ns < 1 || c.ns > 5) {
        set_error("fedm_debug_species_planes_check: no field split set up");
        return -2;
    }
    const size_t n_entries = (size_t)c.pat.total_bc * SLICE, n_s16 = n_entries * c.ns * c.ns, n_v32 = n_entries * c.ns;
    const size_t n_dinv = (size_t)c.pat.n_slices * c.ns * c.ns * SLICE;
    double *t_dinv = nullptr;
    _Float16 *t_s16 = nullptr;
    float *t_v32 = nullptr;
    if (hipMalloc((void **)&t_dinv, sizeof(double) * n_dinv) != hipSuccess ||
        hipMalloc((void **)&t_s16, sizeof(_Float16) * n_s16) != hipSuccess ||
        hipMalloc((void **)&t_v32, sizeof(float) * n_v32) != hipSuccess) {
        set_error("fedm_debug_species_planes_check: hipMalloc failed");
        return -1;
    }
    const unsigned zs = fieldsplit_zero_species_planes(c);
#define FEDM_PLANES(NS_)                                                                                                  \
    hipLaunchKernelGGL(species_planes_kernel<NS_>, dim3(c.pat.n_slices), dim3(256), 0, c.stream, c.pat.n_slices, c.d_slice_boff, \
                       c.d_val, c.d_diag_slot, t_dinv, t_s16, t_v32, fieldsplit_upper(c) ? 1 : 0, zs)
    switch (c.ns) {
        case 1: FEDM_PLANES(1); break;
        case 2: FEDM_PLANES(2); break;
        case 3: FEDM_PLANES(3); break;
        case 4: FEDM_PLANES(4); break;
        case 5: FEDM_PLANES(5); break;
    }
#undef FEDM_PLANES
    std::vector<double> a(n_dinv), b(n_dinv);
    std::vector<_Float16> sa(n_s16), sb(n_s16);
    std::vector<float> va(n_v32), vb(n_v32);
    hipStreamSynchronize(c.stream);
    hipMemcpy(a.data(), c.d_dinv, sizeof(double) * n_dinv, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), t_dinv, sizeof(double) * n_dinv, hipMemcpyDeviceToHost);
    hipMemcpy(sa.data(), c.d_s16, sizeof(_Float16) * n_s16, hipMemcpyDeviceToHost);
    hipMemcpy(sb.data(), t_s16, sizeof(_Float16) * n_s16, hipMemcpyDeviceToHost);
    hipMemcpy(va.data(), c.d_val32, sizeof(float) * n_v32, hipMemcpyDeviceToHost);
    hipMemcpy(vb.data(), t_v32, sizeof(float) * n_v32, hipMemcpyDeviceToHost);
    hipFree(t_dinv);
    hipFree(t_s16);
    hipFree(t_v32);
    double dd = 0.0, dm = 0.0, ds = 0.0, dv = 0.0, vm = 0.0;
    for (size_t i = 0; i < n_dinv; ++i) {
        dd = std::max(dd, std::fabs(a[i] - b[i]));
        dm = std::max(dm, std::fabs(b[i]));
    }
    for (size_t i = 0; i < n_s16; ++i) ds = std::max(ds, (double)std::fabs((float)sa[i] - (float)sb[i]));
    for (size_t i = 0; i < n_v32; ++i) {
        dv = std::max(dv, (double)std::fabs(va[i] - vb[i]));
        vm = std::max(vm, (double)std::fabs(vb[i]));
    }
    out[0] = dm > 0.0 ? dd / dm : dd;
    out[1] = ds;
    out[2] = vm > 0.0 ? dv / vm : dv;
    out[3] = last_fused ? 1.0 : 0.0;
    return 0;
}

// z = [0, V-cycle(t_phi)] : preconditioner of the Poisson-only CG (species rows are identity
// with zero residual there)
__global__ void gather_comp_kernel(int nvp, int neq, int comp, const double *__restrict__ t,
                                   double *__restrict__ b0) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < nvp) b0[v] = t[(size_t)v * neq + comp];
}
__global__ void scatter_comp_kernel(int nvp, int neq, int comp, const double *__restrict__ x0,
                                    double *__restrict__ z) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nvp) return;
    for (int s = 0; s < neq; ++s) z[(size_t)v * neq + s] = (s == comp) ? x0[v] : 0.0;
}

void poisson_precondition(Ctx &c, Amg &amg, const double *r, double *z) {
    const dim3 g((c.nvp + 255) / 256), b(256);
    hipLaunchKernelGGL(gather_comp_kernel, g, b, 0, c.stream, c.nvp, c.neq, c.neq - 1, r, amg.levels[0].b);
    // deep halos: the cycle exchanges nothing itself and needs its right-hand side on the ghost layers
    if (deep_halo_active(c)) comm_halo_scalar(c, amg.levels[0].b);
    amg.run(c);
    hipLaunchKernelGGL(scatter_comp_kernel, g, b, 0, c.stream, c.nvp, c.neq, c.neq - 1, amg.levels[0].x, z);
}

}  // namespace fedm
