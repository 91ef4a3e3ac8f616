// Per-step refresh of the LMEA coefficient fields on the device: reduced electric field by a
// consistent P1 projection (Jacobi-CG on the mass matrix), table look-ups with np.interp's
// semantics, derived rows, mean-energy bookkeeping.  Replaces the host code between two solves
// of examples/glow_discharge/fedm-gd.py:424-443 and :452 (fedm/functions.py:531-750).
#include "amg.hpp"
#include "fedm_internal.hpp"

namespace fedm {

struct GdPrep {
    EllMat M;
    int n_tables = 0, n_rows = 0;
    int *d_tab_ptr = nullptr;
    double *d_tab_x = nullptr, *d_tab_y = nullptr;
    fedm_gd_field_prog *d_progs = nullptr;
    std::vector<fedm_gd_field_prog> progs;
    double *d_redE = nullptr, *d_b = nullptr, *d_r = nullptr, *d_p = nullptr, *d_q = nullptr;
    double *d_cg = nullptr;     // the CG's scalars on the device: r.z, p.q, r.z of the new residual, r.r
    hipGraphExec_t cg_graph = nullptr;   // one CG iteration (its nine launches), replayed
    const double *cg_graph_x = nullptr;  // ... recorded for this solution vector
    bool cg_graph_ok = true;
    unsigned *d_cg_counter = nullptr;    // the single-launch CG: barrier counter, the workgroups' partial sums
    double *d_cg_partials = nullptr;
    bool cg_one_launch_ok = true;
    int nvp = 0;
    void release() {
        if (cg_graph) hipGraphExecDestroy(cg_graph);
        cg_graph = nullptr;
        M.release();
        for (void *p : {(void *)d_tab_ptr, (void *)d_tab_x, (void *)d_tab_y, (void *)d_progs, (void *)d_redE,
                        (void *)d_b, (void *)d_r, (void *)d_p, (void *)d_q, (void *)d_cg, (void *)d_cg_counter,
                        (void *)d_cg_partials})
            if (p) hipFree(p);
    }
};

void gd_prep_release(Ctx &c) {
    if (c.gd_prep) {
        c.gd_prep->release();
        delete c.gd_prep;
        c.gd_prep = nullptr;
    }
}

// b_v += f_c |detJ| / 6 with f_c = 1e21 |grad Phi| / N0 (cell-wise constant): colour by colour
__global__ void redfield_rhs_kernel(int n_cells, const int *__restrict__ cell_list,
                                    const int *__restrict__ cells, const double *__restrict__ coords,
                                    const double *__restrict__ u, int neq, double N0,
                                    double *__restrict__ b) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_cells) return;
    const int c = cell_list[t];
    int v[3];
    double x[3][2], phi[3];
    for (int a = 0; a < 3; ++a) {
        v[a] = cells[3 * c + a];
        x[a][0] = coords[2 * v[a]];
        x[a][1] = coords[2 * v[a] + 1];
        phi[a] = u[(size_t)v[a] * neq + (neq - 1)];
    }
    const double det = (x[1][0] - x[0][0]) * (x[2][1] - x[0][1]) - (x[1][1] - x[0][1]) * (x[2][0] - x[0][0]);
    const double gx = (phi[0] * (x[1][1] - x[2][1]) + phi[1] * (x[2][1] - x[0][1]) + phi[2] * (x[0][1] - x[1][1])) / det;
    const double gy = (phi[0] * (x[2][0] - x[1][0]) + phi[1] * (x[0][0] - x[2][0]) + phi[2] * (x[1][0] - x[0][0])) / det;
    const double f = 1e21 * sqrt(gx * gx + gy * gy) / N0;
    const double w = f * fabs(det) / 6.0;
    for (int a = 0; a < 3; ++a) b[v[a]] += w;
}

// deterministic two-stage dot products on scalar vectors of length n
__global__ __launch_bounds__(256) void sdot2_kernel(int n, const double *__restrict__ a,
                                                    const double *__restrict__ b,
                                                    const double *__restrict__ c2,
                                                    const double *__restrict__ d,
                                                    double *__restrict__ partials) {
    __shared__ double sm[2][4];
    double s0 = 0.0, s1 = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        s0 += a[i] * b[i];
        s1 += c2[i] * d[i];
    }
    for (int off = 32; off > 0; off >>= 1) {
        s0 += __shfl_down(s0, off, 64);
        s1 += __shfl_down(s1, off, 64);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
        sm[0][wave] = s0;
        sm[1][wave] = s1;
    }
    __syncthreads();
    if (threadIdx.x < 2)
        partials[PARTIAL_AT(blockIdx.x, threadIdx.x)] =
            sm[threadIdx.x][0] + sm[threadIdx.x][1] + sm[threadIdx.x][2] + sm[threadIdx.x][3];
}
__global__ void sreduce_kernel(const double *__restrict__ partials, int nblocks, double *__restrict__ out) {
    const int i = blockIdx.x;
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 64) s += partials[PARTIAL_AT(b, i)];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (threadIdx.x == 0) out[i] = s;
}
__global__ void smul_kernel(int n, const double *__restrict__ dinv, const double *__restrict__ r,
                            double *__restrict__ z) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) z[i] = dinv[i] * r[i];
}

// ---- the whole CG as ONE launch ---------------------------------------------------------------------------------
// The mass matrix of the glow-discharge mesh has 40 000 rows: a CG iteration is a few microseconds of work in nine
// dependent launches, each of which costs its 5-8 us launch-to-drain floor whether or not the step is replayed as a
// graph (72 us an iteration, 1 ms a time step, a quarter of the step).  Here a small grid of resident workgroups runs
// all iterations itself, with a counter barrier in device memory between the three phases of an iteration (1.7-3.4 us
// for 8-32 workgroups, tools/grid_barrier_bench): the product with its partial p.q | the updates with the partial sums
// r.z, r.r | the new direction.  Every workgroup sums the partials itself, in a fixed order (reproducible); the spin
// of the barrier is bounded (a barrier that gives up sets info[1] and the caller repeats the solve launch by launch).
__device__ __forceinline__ bool cg_grid_barrier(unsigned *counter, unsigned target) {
    __shared__ int ok;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < (1u << 22))
            __builtin_amdgcn_s_sleep(1);
        ok = spins < (1u << 22);
        __threadfence();
    }
    __syncthreads();
    return ok != 0;
}

// the two sums of a phase: this workgroup's into partials[2 * block + {0, 1}]
__device__ __forceinline__ void cg_block_sums(double s0, double s1, double *partials) {
    __shared__ double sm[2][16];
    for (int off = 32; off > 0; off >>= 1) {
        s0 += __shfl_down(s0, off, 64);
        s1 += __shfl_down(s1, off, 64);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    if (lane == 0) {
        sm[0][wave] = s0;
        sm[1][wave] = s1;
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        double t = 0.0;
        for (int w = 0; w < nw; ++w) t += sm[threadIdx.x][w];
        partials[2 * blockIdx.x + threadIdx.x] = t;
    }
}
// ... and of all workgroups (after a barrier): every thread gets both totals
__device__ __forceinline__ void cg_grid_sums(const double *partials, double &t0, double &t1) {
    double a = 0.0, b = 0.0;
    for (int g = 0; g < (int)gridDim.x; ++g) {       // (at most 64 workgroups: a broadcast read each, the same order everywhere)
        a += partials[2 * g];
        b += partials[2 * g + 1];
    }
    t0 = a;
    t1 = b;
}

__global__ __launch_bounds__(512) void mass_cg_kernel(int n, int n_slices, int log2_split, int width,
                                                      const int *__restrict__ boff, const int *__restrict__ col,
                                                      const double *__restrict__ val, const double *__restrict__ dinv,
                                                      const double *__restrict__ b, double *__restrict__ x,
                                                      double *__restrict__ r, double *__restrict__ p, double *__restrict__ q,
                                                      double rtol, int max_it, unsigned *counter, double *partials,
                                                      double *info) {
    const int T = blockDim.x, G = gridDim.x, tid = blockIdx.x * T + threadIdx.x, nt = G * T;
    const int lane = threadIdx.x & 63, wave_g = tid >> 6, n_waves = nt >> 6;
    double *pa = partials, *pb = partials + 2 * G;      // (p.q | r.z, r.r: two arrays, see the barriers)
    unsigned done = 0;
    bool fine = true;
    // x = 0, r = b, p = Dinv r; r.z and r.r
    double s0 = 0.0, s1 = 0.0;
    for (int i = tid; i < n; i += nt) {
        const double ri = b[i], zi = dinv[i] * ri;
        x[i] = 0.0;
        r[i] = ri;
        p[i] = zi;
        s0 += ri * zi;
        s1 += ri * ri;
    }
    cg_block_sums(s0, s1, pb);
    fine = cg_grid_barrier(counter, ++done * G) && fine;
    double rz, rr0;
    cg_grid_sums(pb, rz, rr0);
    double rr = rr0;
    int it = 0;
    if (rr0 > 0.0 && isfinite(rr0)) {
        for (; it < max_it && fine; ++it) {
            // q = M p (a wave per slice of 64 lanes, as ell_spmv_kernel<0>) and p.q
            s0 = 0.0;
            for (int slice = wave_g; slice < n_slices; slice += n_waves) {
                const int b0 = width > 0 ? slice * width : boff[slice], b1 = width > 0 ? b0 + width : boff[slice + 1];
                const size_t row = ((size_t)slice * SLICE + lane) >> log2_split;
                double acc = 0.0;
                for (int bc = b0; bc < b1; ++bc) {
                    const size_t k = (size_t)bc * SLICE + lane;
                    acc += val[k] * p[col[k]];
                }
                for (int off = (1 << log2_split) >> 1; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
                if ((lane & ((1 << log2_split) - 1)) == 0 && row < (size_t)n) {
                    q[row] = acc;
                    s0 += p[row] * acc;
                }
            }
            cg_block_sums(s0, 0.0, pa);
            fine = cg_grid_barrier(counter, ++done * G) && fine;
            double pq, unused;
            cg_grid_sums(pa, pq, unused);
            const double alpha = pq != 0.0 ? rz / pq : 0.0;
            // x += alpha p, r -= alpha q; r.z and r.r with z = Dinv r
            s0 = s1 = 0.0;
            for (int i = tid; i < n; i += nt) {
                const double ri = r[i] - alpha * q[i];
                x[i] += alpha * p[i];
                r[i] = ri;
                s0 += ri * dinv[i] * ri;
                s1 += ri * ri;
            }
            cg_block_sums(s0, s1, pb);
            fine = cg_grid_barrier(counter, ++done * G) && fine;
            double rz_new;
            cg_grid_sums(pb, rz_new, rr);
            if (!(rr > rtol * rtol * rr0) || !isfinite(rr)) {     // converged (or not a number: the host looks at rr)
                ++it;
                break;
            }
            const double beta = rz != 0.0 ? rz_new / rz : 0.0;
            rz = rz_new;
            for (int i = tid; i < n; i += nt) p[i] = dinv[i] * r[i] + beta * p[i];
            fine = cg_grid_barrier(counter, ++done * G) && fine;
        }
    }
    if (tid == 0) {
        info[0] = (double)it;
        info[1] = fine ? 0.0 : 1.0;
        info[2] = rr0;
        info[3] = rr;
    }
}

// The CG's updates with their scalars read from the device (s = {r.z, p.q, r.z of the new residual, r.r}): the host
// forms neither alpha nor beta and so has nothing to wait for inside an iteration
__global__ void saxpy2_dev_kernel(int n, const double *__restrict__ s, const double *__restrict__ p, double *__restrict__ x,
                                  const double *__restrict__ q, double *__restrict__ r) {
    const double alpha = s[1] != 0.0 ? s[0] / s[1] : 0.0;     // (p.q = 0: the residual is exactly zero already)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        x[i] += alpha * p[i];
        r[i] -= alpha * q[i];
    }
}
__global__ void sxpby_dev_kernel(int n, const double *__restrict__ s, const double *__restrict__ dinv,
                                 const double *__restrict__ r, double *__restrict__ p) {
    const double beta = s[0] != 0.0 ? s[2] / s[0] : 0.0;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = dinv[i] * r[i] + beta * p[i];
}
__global__ void cg_shift_kernel(double *s) { s[0] = s[2]; }

// M x = b by Jacobi-preconditioned CG, rtol on |r|.  An iteration is seven launches and no wait: its scalars stay on
// the device, its |r|^2 goes to the host mailbox, and the host reads the publication of iteration k - 1 after it has
// queued iteration k -- the iteration that runs while the convergence of the previous one is being looked at is the
// only one that may be superfluous (it improves x a little further).  (Until round 4 the host formed alpha and beta:
// two round trips per iteration, 26 per time step of the glow-discharge case, a tenth of its step.)
static int mass_solve_launches(Ctx &c, GdPrep &g, const double *b, double *x, double rtol, int max_it);

static int mass_solve(Ctx &c, GdPrep &g, const double *b, double *x, double rtol, int max_it) {
    // small systems: the whole CG in one launch (above); FEDM_GD_CG=launches (read at set-up) keeps one launch per operation
    if (g.cg_one_launch_ok && !c.capturing && c.nv <= 400000 && g.M.val && g.M.n_slices > 0) {
        const int G = std::max(8, std::min(32, (g.M.n_slices + 15) / 16));
        hipMemsetAsync(g.d_cg_counter, 0, sizeof(unsigned), c.stream);
        hipLaunchKernelGGL(mass_cg_kernel, dim3(G), dim3(512), 0, c.stream, c.nv, g.M.n_slices, g.M.log2_split, g.M.width,
                           g.M.boff, g.M.col, g.M.val, g.M.dinv, b, x, g.d_r, g.d_p, g.d_q, rtol, max_it, g.d_cg_counter,
                           g.d_cg_partials, g.d_cg + 4);
        wait_red_seq(c, publish_values(c, g.d_cg + 4, 4));
        const double its = c.h_red[0], gave_up = c.h_red[1], rr0 = c.h_red[2], rr = c.h_red[3];
        if (gave_up == 0.0) {
            if (!std::isfinite(rr0) || !std::isfinite(rr)) return FEDM_DIVERGED_NAN;
            if (rr0 == 0.0 || rr <= rtol * rtol * rr0) return 0;
            return its >= max_it ? FEDM_DIVERGED_LINEAR : 0;
        }
        g.cg_one_launch_ok = false;   // a barrier gave up (never seen): launch by launch from now on
    }
    return mass_solve_launches(c, g, b, x, rtol, max_it);
}

static int mass_solve_launches(Ctx &c, GdPrep &g, const double *b, double *x, double rtol, int max_it) {
    const int n = c.nv, np = g.nvp;
    const dim3 gv((np + 255) / 256), bv(256);
    int grid = (n + 255) / 256;
    if (grid > RED_BLOCKS) grid = RED_BLOCKS;
    double *s = g.d_cg;
    auto dots = [&](const double *a, const double *bb, const double *cc, const double *dd, double *out) {
        hipLaunchKernelGGL(sdot2_kernel, dim3(grid), dim3(256), 0, c.stream, n, a, bb, cc, dd, c.d_partials);
        hipLaunchKernelGGL(sreduce_kernel, dim3(2), dim3(64), 0, c.stream, c.d_partials, grid, out);
    };
    auto publish = [&](const double *src) { return publish_values(c, src, 2); };
    hipMemsetAsync(x, 0, sizeof(double) * np, c.stream);
    hipMemcpyAsync(g.d_r, b, sizeof(double) * np, hipMemcpyDeviceToDevice, c.stream);
    hipLaunchKernelGGL(smul_kernel, gv, bv, 0, c.stream, n, g.M.dinv, g.d_r, g.d_p);  // p = z = Dinv r
    dots(g.d_r, g.d_p, g.d_r, g.d_r, s);          // s[0] = r.z, s[1] = r.r (p.q of the first iteration overwrites it)
    wait_red_seq(c, publish(s));
    const double r0 = std::sqrt(c.h_red[1]);
    if (r0 == 0.0) return 0;
    if (!std::isfinite(r0)) return FEDM_DIVERGED_NAN;
    // one iteration: nine launches on vectors of 40 000 entries, 3 us of work and 5 us of launch each -- replayed as a
    // graph (recorded once per context; a capture that fails leaves the plain launches)
    auto iteration = [&]() {
        ell_apply(c, g.M, 0, g.d_p, nullptr, g.d_q, 0.0);
        dots(g.d_p, g.d_q, g.d_p, g.d_q, s + 1);                                                       // s[1] = p.q
        hipLaunchKernelGGL(saxpy2_dev_kernel, gv, bv, 0, c.stream, n, s, g.d_p, x, g.d_q, g.d_r);
        hipLaunchKernelGGL(smul_kernel, gv, bv, 0, c.stream, n, g.M.dinv, g.d_r, g.d_q);               // z in q
        dots(g.d_r, g.d_q, g.d_r, g.d_r, s + 2);                                                       // s[2] = r.z, s[3] = r.r
        publish_values_queued(c, s + 2, 2);              // (the host counts the publication itself: a replay has no host code)
        hipLaunchKernelGGL(sxpby_dev_kernel, gv, bv, 0, c.stream, n, s, g.M.dinv, g.d_r, g.d_p);
        hipLaunchKernelGGL(cg_shift_kernel, dim3(1), dim3(1), 0, c.stream, s);
    };
    if (g.cg_graph && g.cg_graph_x != x) {
        hipGraphExecDestroy(g.cg_graph);
        g.cg_graph = nullptr;
    }
    if (!g.cg_graph && g.cg_graph_ok && !c.capturing && !(c.prof.on && c.prof.all_kinds)) {
        hipGraph_t graph = nullptr;
        if (hipStreamBeginCapture(c.stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            c.capturing = true;
            iteration();
            c.capturing = false;
            const bool ok = hipStreamEndCapture(c.stream, &graph) == hipSuccess && graph &&
                            hipGraphInstantiate(&g.cg_graph, graph, nullptr, nullptr, 0) == hipSuccess;
            if (graph) hipGraphDestroy(graph);
            if (!ok) g.cg_graph = nullptr;
        }
        if (!g.cg_graph) {
            hipGetLastError();
            g.cg_graph_ok = false;
        }
        g.cg_graph_x = x;
    }
    unsigned long long previous = 0;
    for (int it = 0; it < max_it; ++it) {
        if (g.cg_graph && hipGraphLaunch(g.cg_graph, c.stream) != hipSuccess) {
            hipGetLastError();
            hipGraphExecDestroy(g.cg_graph);
            g.cg_graph = nullptr;
            g.cg_graph_ok = false;
            iteration();
        } else if (!g.cg_graph) {
            iteration();
        }
        const unsigned long long mine = ++c.mail_seq;
        if (previous) {
            wait_red_seq(c, previous);
            const double rn = std::sqrt(c.h_red[1]);
            if (!std::isfinite(rn)) return FEDM_DIVERGED_NAN;
            if (rn <= rtol * r0) return 0;
        }
        previous = mine;
    }
    wait_red_seq(c, previous);
    const double rn = std::sqrt(c.h_red[1]);
    if (!std::isfinite(rn)) return FEDM_DIVERGED_NAN;
    return rn <= rtol * r0 ? 0 : FEDM_DIVERGED_LINEAR;
}

// np.interp(x, xp, fp): clamped ends, slope*(x - xp[j]) + fp[j] inside
__device__ __forceinline__ double interp1(double x, const double *__restrict__ xp,
                                          const double *__restrict__ fp, int n) {
    if (x <= xp[0]) return fp[0];
    if (x >= xp[n - 1]) return fp[n - 1];
    int lo = 0, hi = n - 1;  // xp[lo] <= x < xp[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (xp[mid] <= x) lo = mid;
        else hi = mid;
    }
    if (x == xp[lo]) return fp[lo];
    const double slope = (fp[lo + 1] - fp[lo]) / (xp[lo + 1] - xp[lo]);
    return slope * (x - xp[lo]) + fp[lo];
}

__global__ void gd_fields_kernel(int nv, int n_rows, const fedm_gd_field_prog *__restrict__ progs,
                                 const int *__restrict__ tab_ptr, const double *__restrict__ tab_x,
                                 const double *__restrict__ tab_y, const double *__restrict__ redE,
                                 const double *__restrict__ uold, int neq, int row_me_old, int row_me,
                                 double *__restrict__ fields) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nv) return;
    // mean_energy_old <- mean_energy first: the tables are evaluated at the old mean energy
    const double me_old = fields[(size_t)row_me * nv + v];
    fields[(size_t)row_me_old * nv + v] = me_old;
    for (int r = 0; r < n_rows; ++r) {
        const fedm_gd_field_prog p = progs[r];
        if (p.kind == FEDM_GDP_TABLE) {
            const double x = (p.arg == FEDM_GDP_ARG_ENERGY) ? me_old : redE[v];
            const int t0 = tab_ptr[p.table], n = tab_ptr[p.table + 1] - t0;
            fields[(size_t)r * nv + v] = interp1(x, tab_x + t0, tab_y + t0, n) * p.scale;
        } else if (p.kind == FEDM_GDP_UE_OLD) {
            fields[(size_t)r * nv + v] = uold[(size_t)v * neq + (neq - 2)];
        }
    }
    for (int r = 0; r < n_rows; ++r) {  // rows derived from other rows (after the look-ups)
        const fedm_gd_field_prog p = progs[r];
        if (p.kind == FEDM_GDP_SCALED_ROW) fields[(size_t)r * nv + v] = p.scale * fields[(size_t)p.src_row * nv + v];
    }
}

__global__ void gd_mean_energy_kernel(int nv, int neq, const double *__restrict__ u, int row_me,
                                      double *__restrict__ fields) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < nv) fields[(size_t)row_me * nv + v] = exp(u[(size_t)v * neq] - u[(size_t)v * neq + (neq - 2)]);
}

int gd_prep_setup(Ctx &c, const fedm_csr *mass, int n_tables, const int32_t *tab_ptr,
                  const double *tab_x, const double *tab_y, const fedm_gd_field_prog *progs) {
    gd_prep_release(c);
    GdPrep *g = new GdPrep();
    if (g->M.from_csr(*mass, true, 0)) {
        set_error("mass matrix upload failed");
        delete g;
        return -1;
    }
    g->nvp = g->M.n_rows_p;
    g->n_tables = n_tables;
    g->n_rows = c.gd_n_fields;
    g->progs.assign(progs, progs + g->n_rows);
    const int ntab = tab_ptr[n_tables];
    FEDM_HIP_CHECK(hipMalloc((void **)&g->d_tab_ptr, sizeof(int) * (n_tables + 1)));
    FEDM_HIP_CHECK(hipMalloc((void **)&g->d_tab_x, sizeof(double) * std::max(ntab, 1)));
    FEDM_HIP_CHECK(hipMalloc((void **)&g->d_tab_y, sizeof(double) * std::max(ntab, 1)));
    FEDM_HIP_CHECK(hipMalloc((void **)&g->d_progs, sizeof(fedm_gd_field_prog) * g->n_rows));
    FEDM_HIP_CHECK(hipMemcpy(g->d_tab_ptr, tab_ptr, sizeof(int) * (n_tables + 1), hipMemcpyHostToDevice));
    FEDM_HIP_CHECK(hipMemcpy(g->d_tab_x, tab_x, sizeof(double) * ntab, hipMemcpyHostToDevice));
    FEDM_HIP_CHECK(hipMemcpy(g->d_tab_y, tab_y, sizeof(double) * ntab, hipMemcpyHostToDevice));
    FEDM_HIP_CHECK(hipMemcpy(g->d_progs, progs, sizeof(fedm_gd_field_prog) * g->n_rows, hipMemcpyHostToDevice));
    for (double **p : {&g->d_redE, &g->d_b, &g->d_r, &g->d_p, &g->d_q}) {
        FEDM_HIP_CHECK(hipMalloc((void **)p, sizeof(double) * g->nvp));
        FEDM_HIP_CHECK(hipMemset(*p, 0, sizeof(double) * g->nvp));
    }
    FEDM_HIP_CHECK(hipMalloc((void **)&g->d_cg, sizeof(double) * 8));
    FEDM_HIP_CHECK(hipMemset(g->d_cg, 0, sizeof(double) * 8));
    {
        const char *e = std::getenv("FEDM_GD_CG");     // (read when the pipeline is set up)
        g->cg_one_launch_ok = !(e && e[0] == 'l');
    }
    FEDM_HIP_CHECK(hipMalloc((void **)&g->d_cg_counter, sizeof(unsigned) * 4));
    FEDM_HIP_CHECK(hipMalloc((void **)&g->d_cg_partials, sizeof(double) * 4 * 64));
    c.gd_prep = g;
    return 0;
}

int gd_prep_step(Ctx &c) {
    GdPrep &g = *c.gd_prep;
    const int ns = c.gd.n_species, nr = c.gd.n_reactions;
    const int row_me_old = 4 * ns + 2 * nr, row_me = row_me_old + 1;
    // reduced field of the current potential: project(1e21*sqrt(dot(-grad Phi, -grad Phi))/N0)
    hipMemsetAsync(g.d_b, 0, sizeof(double) * g.nvp, c.stream);
    const int ncol = (int)c.pat.colour_ptr.size() - 1;
    for (int k = 0; k < ncol; ++k) {
        const int n = c.pat.colour_ptr[k + 1] - c.pat.colour_ptr[k];
        if (n == 0) continue;
        hipLaunchKernelGGL(redfield_rhs_kernel, dim3((n + 255) / 256), dim3(256), 0, c.stream, n,
                           c.d_colour_cells + c.pat.colour_ptr[k], c.d_cells, c.d_coords, c.d_u, c.neq,
                           c.gd.N0, g.d_b);
    }
    const int rc = mass_solve(c, g, g.d_b, g.d_redE, 1e-14, 500);
    if (rc) return rc;
    hipLaunchKernelGGL(gd_fields_kernel, dim3((c.nv + 255) / 256), dim3(256), 0, c.stream, c.nv, g.n_rows,
                       g.d_progs, g.d_tab_ptr, g.d_tab_x, g.d_tab_y, g.d_redE, c.d_uold, c.neq, row_me_old,
                       row_me, c.d_gd_fields);
    return 0;
}

void gd_update_mean_energy(Ctx &c) {
    const int row_me = 4 * c.gd.n_species + 2 * c.gd.n_reactions + 1;
    hipLaunchKernelGGL(gd_mean_energy_kernel, dim3((c.nv + 255) / 256), dim3(256), 0, c.stream, c.nv, c.neq,
                       c.d_u, row_me, c.d_gd_fields);
}


// ---- Expression sources on the device (fedm_ext_source_program / fedm_ext_source_eval) -----------
// One thread per (cell, lattice node): the node's coordinates from the cell's vertices, then the
// postfix program.  The node order is the lattice order of the host path: (i / d, j / d) for
// j = 0..d, i = 0..d - j, barycentric weights (1 - l1 - l2, l1, l2).
struct ExprParams {
    double p[FEDM_EXPR_MAX_PARAMS];
};

__global__ __launch_bounds__(256) void ext_source_eval_kernel(int nc, int nodes, int degree,
                                                              const int *__restrict__ cells,
                                                              const double *__restrict__ coords,
                                                              const int *__restrict__ ops, int n_ops,
                                                              const double *__restrict__ consts, ExprParams P,
                                                              double *__restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nc * nodes) return;
    const int cell = t / nodes;
    int m = t - cell * nodes, j = 0;
    while (m > degree - j) {   // row j of the lattice holds degree + 1 - j nodes
        m -= degree + 1 - j;
        ++j;
    }
    const double l1 = (double)m / degree, l2 = (double)j / degree, l0 = 1.0 - l1 - l2;
    const int v0 = cells[3 * cell], v1 = cells[3 * cell + 1], v2 = cells[3 * cell + 2];
    const double x[2] = {l0 * coords[2 * v0] + l1 * coords[2 * v1] + l2 * coords[2 * v2],
                         l0 * coords[2 * v0 + 1] + l1 * coords[2 * v1 + 1] + l2 * coords[2 * v2 + 1]};
    double st[FEDM_EXPR_STACK];
    int sp = 0;
    for (int k = 0; k < n_ops; ++k) {
        const int op = ops[2 * k], arg = ops[2 * k + 1];
        if (op <= FEDM_OP_PARAM) {
            st[sp++] = op == FEDM_OP_CONST ? consts[arg] : op == FEDM_OP_X ? x[arg] : P.p[arg];
        } else if (op <= FEDM_OP_POW) {
            const double b = st[--sp], a = st[sp - 1];
            st[sp - 1] = op == FEDM_OP_ADD ? a + b : op == FEDM_OP_SUB ? a - b : op == FEDM_OP_MUL ? a * b
                       : op == FEDM_OP_DIV ? a / b : pow(a, b);
        } else {
            const double a = st[sp - 1];
            double r;
            switch (op) {
                case FEDM_OP_NEG: r = -a; break;
                case FEDM_OP_EXP: r = exp(a); break;
                case FEDM_OP_LOG: r = log(a); break;
                case FEDM_OP_SQRT: r = sqrt(a); break;
                case FEDM_OP_SIN: r = sin(a); break;
                case FEDM_OP_COS: r = cos(a); break;
                case FEDM_OP_TAN: r = tan(a); break;
                case FEDM_OP_FABS: r = fabs(a); break;
                case FEDM_OP_TANH: r = tanh(a); break;
                default: r = atan(a); break;
            }
            st[sp - 1] = r;
        }
    }
    out[t] = st[0];
}

void launch_ext_source_eval(Ctx &c, int species, const double *params) {
    const int nodes = c.model.ext_nodes[species];
    const int degree = nodes == 3 ? 1 : nodes == 6 ? 2 : 3;
    ExprParams P;
    for (int i = 0; i < FEDM_EXPR_MAX_PARAMS; ++i) P.p[i] = i < c.expr_n_params[species] ? params[i] : 0.0;
    const int n = c.nc * nodes;
    hipLaunchKernelGGL(ext_source_eval_kernel, dim3((n + 255) / 256), dim3(256), 0, c.stream, c.nc, nodes, degree,
                       c.d_cells, c.d_coords, c.d_expr_ops[species], c.expr_n_ops[species],
                       c.d_expr_consts[species], P, c.d_ext[species]);
}

}  // namespace fedm
