// C ABI of libfedm_hip.so (include/fedm_hip.h): context, state, Newton / GMRES drivers.
// Host logic only; every flop of the hot path runs in kernels.hip.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>

#include "amg.hpp"
#include "comm.hpp"
#include "fedm_internal.hpp"

struct fedm_ctx {
    fedm::Ctx c;
};

namespace fedm {

static thread_local std::string g_error;
void set_error(const std::string &msg) { g_error = msg; }

void prof_collect(Ctx &c) {
    Prof &p = c.prof;
    if (p.used == 0) return;
    hipStreamSynchronize(c.stream);
    for (int i = 0; i + 1 < p.used; i += 2) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.ev[i], p.ev[i + 1]) == hipSuccess) {
            p.ms[p.kind[i / 2]] += ms;
            p.cnt[p.kind[i / 2]]++;
        }
    }
    p.used = 0;
}

void prof_begin(Ctx &c, int kind) {
    Prof &p = c.prof;
    p.recording = false;
    if (!p.on || c.capturing || ((kind == 1 || kind == 3) && !p.all_kinds)) return;
    // events cost a few microseconds of stream time each: short, frequent kernels are sampled
    p.recording = (p.seen[kind]++ % p.stride[kind]) == 0;
    if (!p.recording) return;
    if (p.used + 2 > (int)p.ev.size()) prof_collect(c);
    p.kind[p.used / 2] = kind;
    hipEventRecord(p.ev[p.used], c.stream);
}

void prof_end(Ctx &c) {
    Prof &p = c.prof;
    if (!p.on || !p.recording || c.capturing) return;
    hipEventRecord(p.ev[p.used + 1], c.stream);
    p.used += 2;
    p.recording = false;
}

template <class T>
static int upload(T *&dst, const T *src, size_t n) {
    FEDM_HIP_CHECK(hipMalloc((void **)&dst, sizeof(T) * std::max<size_t>(n, 1)));
    if (n) FEDM_HIP_CHECK(hipMemcpy(dst, src, sizeof(T) * n, hipMemcpyHostToDevice));
    return 0;
}

static int alloc_zero(double *&p, size_t n, hipStream_t st) {
    FEDM_HIP_CHECK(hipMalloc((void **)&p, sizeof(double) * std::max<size_t>(n, 1)));
    FEDM_HIP_CHECK(hipMemsetAsync(p, 0, sizeof(double) * std::max<size_t>(n, 1), st));
    return 0;
}

static int check_model(const fedm_model_desc &m) {
    if (m.n_species < 1 || m.n_species > FEDM_MAX_SPECIES) return 1;
    if (m.n_reactions < 0 || m.n_reactions > FEDM_MAX_REACTIONS) return 1;
    if (m.n_qp < 1 || m.n_qp > FEDM_MAX_QP || m.n_fqp < 0 || m.n_fqp > FEDM_MAX_FQP) return 1;
    if (m.n_tags < 0 || m.n_tags > FEDM_MAX_TAGS) return 1;
    const int ns = m.n_species, po = m.poisson ? 1 : 0;
    const bool ok = (ns == 1) || (ns == 2) || ((ns == 3 || ns == 4) && po);
    if (!ok) return 1;
    for (int s = 0; s < ns; ++s) {
        if (m.eq_type[s] < 0 || m.eq_type[s] > 2) return 1;
        if (m.mu[s].n_terms < 0 || m.mu[s].n_terms > FEDM_MAX_TERMS) return 1;
        if (m.D[s].n_terms < 0 || m.D[s].n_terms > FEDM_MAX_TERMS) return 1;
        if (m.ext_nodes[s] < 0 || m.ext_nodes[s] > FEDM_MAX_EXT_NODES) return 1;
    }
    for (int j = 0; j < m.n_reactions; ++j)
        if (m.k[j].n_terms < 0 || m.k[j].n_terms > FEDM_MAX_TERMS) return 1;
    return 0;
}

// the alternative set of species sweeps and (when installed) the alternative V-cycle, together
static void set_hard_mode(Ctx &c, bool hard) {
    if (c.fs_alt_active == hard) return;
    c.fs_skip_sample = true;   // (measured policy: the next solve captures its graphs anew)
    hipStreamSynchronize(c.stream);
    iter_graphs_clear(c);  // the captured steps contain the other sweeps / the other cycle
    c.fs_alt_active = hard;
    if (c.fs_alt_sweeps > 0) {
        c.fs_sweeps = hard ? c.fs_alt_sweeps : c.fs_main_sweeps;
        for (int i = 0; i < c.fs_sweeps; ++i) c.fs_w[i] = hard ? c.fs_alt_w[i] : c.fs_main_w[i];
    }
    if (c.amg_alt) std::swap(c.amg, c.amg_alt);
}

static int ensure_krylov(Ctx &c, int restart) {
    if (ensure_spmv_dots(c)) return -1;
    if (restart + 1 <= c.krylov_cap) return 0;
    iter_graphs_clear(c);  // they hold the old Krylov vectors' addresses
    if (c.d_V) hipFree(c.d_V);
    c.d_V = nullptr;
    FEDM_HIP_CHECK(hipMalloc((void **)&c.d_V, sizeof(double) * (size_t)c.np * (restart + 1)));
    if (c.d_Z) hipFree(c.d_Z);
    c.d_Z = nullptr;
    FEDM_HIP_CHECK(hipMalloc((void **)&c.d_Z, sizeof(double) * (size_t)c.np * restart));
    c.krylov_cap = restart + 1;
    return 0;
}

// w = Minv (J v): point-block Jacobi (fused in the SpMV) or field split with multigrid
static void apply_operator(Ctx &c, const double *v, double *w) {
    comm_halo(c, const_cast<double *>(v));  // ghost inputs from their owners (multi-GPU)
    if (c.amg && c.poisson) {
        fieldsplit_apply_operator(c, *c.amg, v, c.d_tmp, w, true);
    } else {
        prof_begin(c, 1);
        launch_spmv(c, v, w, true);
        prof_end(c);
    }
}

// the field split sits on the right of the operator (flexible GMRES)
// (across GPUs always: the left variant is kept for one GPU only -- its restarted cycles were never
// made to work over several ranks)
static bool right_preconditioned(const Ctx &c) { return (c.right_precond || c.comm) && c.amg && c.poisson; }

// rhs = -Minv F (preconditioner on the left) or -F (on the right), after the Jacobian has been assembled
static void prepare_preconditioner_and_rhs(Ctx &c) {
    if (right_preconditioned(c)) {
        fieldsplit_setup(c);  // the right-hand side is -F itself: gmres reads c.d_F
    } else if (c.amg && c.poisson) {
        fieldsplit_setup(c);
        fieldsplit_apply(c, *c.amg, c.d_F, c.d_rhs, -1.0);
    } else {
        launch_block_inverse(c);
        launch_apply_dinv(c, c.d_F, c.d_rhs, -1.0);
    }
}

void iter_graphs_clear(Ctx &c) {
    for (hipGraphExec_t g : c.iter_graph)
        if (g) hipGraphExecDestroy(g);
    for (hipGraphExec_t g : c.iter_graph_interior)
        if (g) hipGraphExecDestroy(g);
    for (hipGraphExec_t g : c.iter_graph_pre)
        if (g) hipGraphExecDestroy(g);
    for (auto *vec : {&c.iter_graph_pair, &c.iter_graph_last, &c.iter_graph_pair_last}) {
        for (hipGraphExec_t g : *vec)
            if (g) hipGraphExecDestroy(g);
        vec->clear();
    }
    c.iter_graph.clear();
    c.iter_graph_interior.clear();
    c.iter_graph_pre.clear();
}

// One Krylov step  w = Minv J v_j;  h = V^T w;  w <- (w - V h)/|.|  as a hipGraph, captured the
// first time index j is reached: ~30 kernels replayed back to back with no launch gaps and one
// host call.  All pointers are fixed for a given j; the mailbox tag is a device counter.
// Across GPUs the collectives stay outside the graphs, and the halo exchange of v_j runs on the
// communication stream while the compute stream already multiplies the interior matrix slices
// (those without ghost columns):
//   mark v_j complete -> graph I_j (interior SpMV)  ||  exchange  -> wait -> graph B_j (boundary
//   SpMV, preconditioner, local partial sums) -> all-reduce -> finish/publish -> update.
// Producers of Krylov vectors.  One GPU, field split on the right with species sweeps (Ctx::fs_first_by_producer, set by
// gmres): the kernel that completes v_j also forms the first stage of the preconditioner for it (cgs_update_fs_kernel).
static void krylov_vector_update(Ctx &c, int k, const double *const *vp, double *w) {
    if (c.fs_first_by_producer) {
        if (!launch_cgs_update_fs(c, k, vp, w, reinterpret_cast<float *>(c.d_fs_g), c.amg->levels[0].b)) {
            launch_cgs_update(c, k, vp, w);
            fieldsplit_first_stage(c, *c.amg, w);
        }
    } else {
        launch_cgs_update(c, k, vp, w);
    }
}

static void krylov_vector_scale(Ctx &c, double a, const double *x, double *y) {
    if (c.fs_first_by_producer) {
        if (!launch_scale_copy_fs(c, a, x, y, reinterpret_cast<float *>(c.d_fs_g), c.amg->levels[0].b)) {
            launch_scale_copy(c, a, x, y);
            fieldsplit_first_stage(c, *c.amg, y);
        }
    } else {
        launch_scale_copy(c, a, x, y);
    }
}

static bool capture_graph(Ctx &c, hipGraphExec_t *out, const std::function<void()> &body) {
    hipGraph_t graph = nullptr;
    if (hipStreamBeginCapture(c.stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        hipGetLastError();
        return false;
    }
    c.capturing = true;
    body();
    c.capturing = false;
    hipGraphExec_t exec = nullptr;
    const bool ok = hipStreamEndCapture(c.stream, &graph) == hipSuccess && graph &&
                    hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
    if (graph) hipGraphDestroy(graph);
    if (!ok) {
        hipGetLastError();
        return false;
    }
    *out = exec;
    return true;
}

static bool iter_graph_launch(Ctx &c, int j, const double *const *vp, double *w) {
    if (!c.iter_graphs_ok || !(c.amg && c.poisson) || (c.prof.on && c.prof.all_kinds)) return false;
    const bool multi = c.comm != nullptr;
    if ((int)c.iter_graph.size() <= j) {
        c.iter_graph.resize(j + 1, nullptr);
        c.iter_graph_interior.resize(j + 1, nullptr);
    }
    if (!c.iter_graph[j]) {
        std::vector<const double *> dotp(j + 2);
        for (int i = 0; i <= j; ++i) dotp[i] = vp[i];
        dotp[j + 1] = w;
        // the V-cycle's last sweep writes the potential component of w itself (V(nu,nu) with
        // more than one level); otherwise the reduction kernel scatters it
        const bool direct = c.amg->pre_smooth && c.amg->levels.size() > 1;
        const double *x0 = direct ? nullptr : c.amg->levels[0].x;
        auto with_direct_output = [&](const std::function<void()> &f) {
            if (direct) {
                c.amg->out = w;
                c.amg->out_stride = c.neq;
                c.amg->out_offset = c.neq - 1;
            }
            f();
            c.amg->out = nullptr;
        };
        bool ok = true;
        if (!multi) {
            ok = capture_graph(c, &c.iter_graph[j], [&] {
                with_direct_output([&] { fieldsplit_apply_operator(c, *c.amg, vp[j], c.d_tmp, w, false); });
                launch_dots_fused(c, dotp.data(), w, j + 2, x0, true);
                krylov_vector_update(c, j + 1, vp, w);
            });
        } else {
            Comm &cm = *c.comm;
            if (cm.n_interior)
                ok = capture_graph(c, &c.iter_graph_interior[j], [&] {
                    fieldsplit_apply_operator_part(c, *c.amg, vp[j], c.d_tmp, w, false, 0, cm.d_interior, cm.n_interior);
                });
            if (c.amg->global)  // the V-cycle contains collectives: the graph ends at its right-hand side
                ok = ok && capture_graph(c, &c.iter_graph[j], [&] {
                    fieldsplit_apply_operator_part(c, *c.amg, vp[j], c.d_tmp, w, false, 2, cm.d_boundary, cm.n_boundary);
                });
            else
                ok = ok && capture_graph(c, &c.iter_graph[j], [&] {
                    with_direct_output([&] {
                        fieldsplit_apply_operator_part(c, *c.amg, vp[j], c.d_tmp, w, false, 1, cm.d_boundary, cm.n_boundary);
                    });
                    launch_dots_fused(c, dotp.data(), w, j + 2, x0, false);
                });
        }
        if (!ok) {
            c.iter_graphs_ok = false;
            return false;
        }
    }
    if (!multi) {
        if (hipGraphLaunch(c.iter_graph[j], c.stream) != hipSuccess) {
            hipGetLastError();
            c.iter_graphs_ok = false;
            return false;
        }
        ++c.mail_seq;
        return true;
    }
    // FEDM_HALO_OVERLAP=0 keeps the exchange on the compute stream (diagnostics)
    static const bool overlap = [] {
        const char *e = std::getenv("FEDM_HALO_OVERLAP");
        return !(e && e[0] == '0');
    }();
    if (overlap) comm_halo_begin(c);
    else comm_halo(c, const_cast<double *>(vp[j]));
    bool ok = !c.iter_graph_interior[j] || hipGraphLaunch(c.iter_graph_interior[j], c.stream) == hipSuccess;
    if (overlap) comm_halo_exchange(c, const_cast<double *>(vp[j]));
    ok = ok && hipGraphLaunch(c.iter_graph[j], c.stream) == hipSuccess;
    if (!ok) {  // the exchange has happened: redo the whole step with plain launches (same result)
        hipGetLastError();
        c.iter_graphs_ok = false;
        std::vector<const double *> dotp(j + 2);
        for (int i = 0; i <= j; ++i) dotp[i] = vp[i];
        dotp[j + 1] = w;
        fieldsplit_apply_operator(c, *c.amg, vp[j], c.d_tmp, w, true);
        launch_dots(c, dotp.data(), w, j + 2, true);
        krylov_vector_update(c, j + 1, vp, w);
        return true;
    }
    if (c.amg->global) {  // V-cycle with its collectives, then scatter + local partial sums
        c.amg->run(c);
        std::vector<const double *> dotp(j + 2);
        for (int i = 0; i <= j; ++i) dotp[i] = vp[i];
        dotp[j + 1] = w;
        launch_dots_fused(c, dotp.data(), w, j + 2, c.amg->levels[0].x, false);
    }
    comm_allreduce(c, c.d_red, j + 2);
    launch_cgs_finish(c, j + 2);
    krylov_vector_update(c, j + 1, vp, w);
    return true;
}

// The same step with the field split on the right:  z_j = Minv v_j (kept),  w = J z_j,  h = V^T w,
// w <- (w - V h)/|.|.  One GPU: one graph.  Several GPUs: it is z_j whose ghost entries the
// product needs --
//   graph P_j (first stage, sweeps, coupling[, V-cycle]) [-> V-cycle with its collectives] ->
//   mark z_j complete -> graph I_j (interior rows of J z_j)  ||  exchange of z_j -> wait ->
//   graph B_j (boundary rows, local partial sums) -> all-reduce -> finish/publish -> update.
// The Krylov vectors keep zero ghost entries (ghost rows of the product are zero), so Minv sees
// the same inputs as on the left.
// (deep halos: the caller has exchanged v_j already)
static void right_step_plain_after_exchange(Ctx &c, int j, const double *const *vp, double *z, double *w) {
    fieldsplit_apply(c, *c.amg, vp[j], z, 1.0);
    if (!deep_halo_active(c)) comm_halo(c, z);
    prof_begin(c, 1);
    launch_spmv(c, z, w, false);
    prof_end(c);
    std::vector<const double *> dotp(j + 2);
    for (int i = 0; i <= j; ++i) dotp[i] = vp[i];
    dotp[j + 1] = w;
    launch_dots(c, dotp.data(), w, j + 2, true);
    krylov_vector_update(c, j + 1, vp, w);
}

static void right_step_plain(Ctx &c, int j, const double *const *vp, double *z, double *w) {
    const bool deep = deep_halo_active(c);
    if (deep) comm_halo(c, const_cast<double *>(vp[j]));   // the step's one exchange: its input on all ghost layers
    fieldsplit_apply(c, *c.amg, vp[j], z, 1.0);
    if (!deep) comm_halo(c, z);
    prof_begin(c, 1);
    launch_spmv(c, z, w, false);
    prof_end(c);
    std::vector<const double *> dotp(j + 2);
    for (int i = 0; i <= j; ++i) dotp[i] = vp[i];
    dotp[j + 1] = w;
    launch_dots(c, dotp.data(), w, j + 2, true);
    krylov_vector_update(c, j + 1, vp, w);
}

// One GPU, field split on the right: Krylov steps j and j + 1 as ONE graph (a step needs nothing from the host, and
// when the previous solve says that both will be needed they are launched together anyway; between two graph
// launches the GPU idles for 8 us, tools/step_sequence.py).  Publishes twice: mail_seq advances by two.
static bool iter_graph_launch_right_pair(Ctx &c, int j, const double *const *vp, double *z0, double *w0, double *z1,
                                         double *w1, bool skip_last_update) {
    static const bool off = [] {
        const char *e = std::getenv("FEDM_KRYLOV_PAIRS");
        return e && e[0] == '0';
    }();
    if (off || c.comm || !c.iter_graphs_ok || (c.prof.on && c.prof.all_kinds) || fieldsplit_upper(c)) return false;
    std::vector<hipGraphExec_t> &cache = skip_last_update ? c.iter_graph_pair_last : c.iter_graph_pair;
    if ((int)cache.size() <= j) cache.resize(j + 1, nullptr);
    if (!cache[j]) {
        const bool direct = c.amg->pre_smooth && c.amg->levels.size() > 1;
        auto step = [&](int jj, double *z, double *w, bool update) {
            std::vector<const double *> dotp(jj + 2);
            for (int i = 0; i <= jj; ++i) dotp[i] = vp[i];
            dotp[jj + 1] = w;
            if (direct) {
                c.amg->out = z;
                c.amg->out_stride = c.neq;
                c.amg->out_offset = c.neq - 1;
            }
            fieldsplit_apply(c, *c.amg, vp[jj], z, 1.0, !direct);
            c.amg->out = nullptr;
            if (!launch_spmv_dots(c, z, w, dotp.data(), jj + 2)) {
                launch_spmv(c, z, w, false);
                launch_dots_fused(c, dotp.data(), w, jj + 2, nullptr, true);
            }
            if (update) krylov_vector_update(c, jj + 1, vp, w);
        };
        if (!capture_graph(c, &cache[j], [&] {
                step(j, z0, w0, true);
                step(j + 1, z1, w1, !skip_last_update);
            })) {
            c.iter_graphs_ok = false;
            return false;
        }
    }
    if (hipGraphLaunch(cache[j], c.stream) != hipSuccess) {
        hipGetLastError();
        c.iter_graphs_ok = false;
        return false;
    }
    c.mail_seq += 2;
    return true;
}

static bool iter_graph_launch_right(Ctx &c, int j, const double *const *vp, double *z, double *w,
                                    bool skip_update = false) {
    if (!c.iter_graphs_ok || (c.prof.on && c.prof.all_kinds)) return false;
    const bool multi = c.comm != nullptr;
    if (skip_update && !multi) {   // one GPU: the step expected to end the solve, without its update
        if ((int)c.iter_graph_last.size() <= j) c.iter_graph_last.resize(j + 1, nullptr);
        if (!c.iter_graph_last[j]) {
            const bool direct_ = !fieldsplit_upper(c) && c.amg->pre_smooth && c.amg->levels.size() > 1;
            std::vector<const double *> dotp(j + 2);
            for (int i = 0; i <= j; ++i) dotp[i] = vp[i];
            dotp[j + 1] = w;
            if (!capture_graph(c, &c.iter_graph_last[j], [&] {
                    if (direct_) {
                        c.amg->out = z;
                        c.amg->out_stride = c.neq;
                        c.amg->out_offset = c.neq - 1;
                    }
                    fieldsplit_apply(c, *c.amg, vp[j], z, 1.0, !direct_);
                    c.amg->out = nullptr;
                    if (!launch_spmv_dots(c, z, w, dotp.data(), j + 2)) {
                        launch_spmv(c, z, w, false);
                        launch_dots_fused(c, dotp.data(), w, j + 2, nullptr, true);
                    }
                })) {
                c.iter_graphs_ok = false;
                return false;
            }
        }
        if (hipGraphLaunch(c.iter_graph_last[j], c.stream) != hipSuccess) {
            hipGetLastError();
            c.iter_graphs_ok = false;
            return false;
        }
        ++c.mail_seq;
        return true;
    }
    if ((int)c.iter_graph.size() <= j) {
        c.iter_graph.resize(j + 1, nullptr);
        c.iter_graph_interior.resize(j + 1, nullptr);
        c.iter_graph_pre.resize(j + 1, nullptr);
    }
    // the V-cycle's last sweep writes the potential component of z itself (V(nu,nu) with more
    // than one level); otherwise a scatter kernel does.  Upper-triangular order: the V-cycle comes
    // first and the last species sweep writes the whole of z.
    const bool upper = fieldsplit_upper(c);
    const bool direct = !upper && c.amg->pre_smooth && c.amg->levels.size() > 1;
    auto with_direct_output = [&](const std::function<void()> &f) {
        if (direct) {
            c.amg->out = z;
            c.amg->out_stride = c.neq;
            c.amg->out_offset = c.neq - 1;
        }
        f();
        c.amg->out = nullptr;
    };
    if (!c.iter_graph[j]) {
        std::vector<const double *> dotp(j + 2);
        for (int i = 0; i <= j; ++i) dotp[i] = vp[i];
        dotp[j + 1] = w;
        bool ok = true;
        if (!multi) {
            ok = capture_graph(c, &c.iter_graph[j], [&] {
                with_direct_output([&] { fieldsplit_apply(c, *c.amg, vp[j], z, 1.0, !direct); });
                if (!launch_spmv_dots(c, z, w, dotp.data(), j + 2)) {
                    launch_spmv(c, z, w, false);
                    launch_dots_fused(c, dotp.data(), w, j + 2, nullptr, true);
                }
                krylov_vector_update(c, j + 1, vp, w);
            });
        } else if (deep_halo_active(c) && !upper) {
            // deep halos: nothing is exchanged inside the step, so it is cut at its two all-reduces only --
            // [field split + the cycle's leg down] | level-1 all-reduce | [the cycle's leg up + the whole
            // Krylov product + local dot products] | all-reduce
            ok = capture_graph(c, &c.iter_graph_pre[j], [&] {
                with_direct_output([&] { fieldsplit_apply(c, *c.amg, vp[j], z, 1.0, !direct, false); });
                c.amg->vcycle(c, 0, 1);
            });
            ok = ok && capture_graph(c, &c.iter_graph[j], [&] {
                with_direct_output([&] { c.amg->vcycle(c, 0, 2); });
                if (!direct) fieldsplit_scatter(c, *c.amg, z);
                if (!launch_spmv_dots(c, z, w, dotp.data(), j + 2, false)) {
                    launch_spmv(c, z, w, false);
                    launch_dots_fused(c, dotp.data(), w, j + 2, nullptr, false);
                }
            });
        } else {
            Comm &cm = *c.comm;
            const bool cycle_inside = !c.amg->global;  // no collectives in the V-cycle
            ok = capture_graph(c, &c.iter_graph_pre[j], [&] {
                if (upper) fieldsplit_upper_species(c, *c.amg, vp[j], z, 1.0);  // the V-cycle and its halo come before
                else with_direct_output([&] { fieldsplit_apply(c, *c.amg, vp[j], z, 1.0, !direct, cycle_inside); });
            });
            if (cm.n_interior)
                ok = ok && capture_graph(c, &c.iter_graph_interior[j], [&] {
                    launch_spmv(c, z, w, false, cm.d_interior, cm.n_interior);
                });
            ok = ok && capture_graph(c, &c.iter_graph[j], [&] {
                launch_spmv(c, z, w, false, cm.d_boundary, cm.n_boundary);
                launch_dots_fused(c, dotp.data(), w, j + 2, nullptr, false);
            });
        }
        if (!ok) {
            c.iter_graphs_ok = false;
            return false;
        }
    }
    if (!multi) {
        if (hipGraphLaunch(c.iter_graph[j], c.stream) != hipSuccess) {
            hipGetLastError();
            c.iter_graphs_ok = false;
            return false;
        }
        ++c.mail_seq;
        return true;
    }
    const bool deep = deep_halo_active(c);
    // deep halos: v_j on all ghost layers, then sweeps, smoothings and the product without an exchange
    if (deep) comm_halo(c, const_cast<double *>(vp[j]));
    if (deep && !upper) {
        if (hipGraphLaunch(c.iter_graph_pre[j], c.stream) != hipSuccess) {
            hipGetLastError();
            c.iter_graphs_ok = false;
            right_step_plain_after_exchange(c, j, vp, z, w);
            return true;
        }
        c.amg->allreduce_level1(c);
        if (hipGraphLaunch(c.iter_graph[j], c.stream) != hipSuccess) {   // the same leg up and product, plainly
            hipGetLastError();
            c.iter_graphs_ok = false;
            with_direct_output([&] { c.amg->vcycle(c, 0, 2); });
            if (!direct) fieldsplit_scatter(c, *c.amg, z);
            std::vector<const double *> dotp(j + 2);
            for (int i = 0; i <= j; ++i) dotp[i] = vp[i];
            dotp[j + 1] = w;
            launch_spmv(c, z, w, false);
            launch_dots(c, dotp.data(), w, j + 2, true);
            krylov_vector_update(c, j + 1, vp, w);
            return true;
        }
        comm_allreduce(c, c.d_red, j + 2);
        launch_cgs_finish(c, j + 2);
        krylov_vector_update(c, j + 1, vp, w);
        return true;
    }
    if (upper) {
        // potential first (the V-cycle with its collectives, then the ghost entries of its result),
        // then the species part: plain launches with the exchanges between the sweeps, or its graph
        fieldsplit_upper_potential(c, *c.amg, vp[j], 1.0);
        if (c.fs_halo) {
            fieldsplit_upper_species(c, *c.amg, vp[j], z, 1.0);
        } else if (hipGraphLaunch(c.iter_graph_pre[j], c.stream) != hipSuccess) {
            hipGetLastError();
            c.iter_graphs_ok = false;
            fieldsplit_upper_species(c, *c.amg, vp[j], z, 1.0);  // same result with plain launches
        }
    } else if (c.fs_halo && !deep) {
        // exchanges between the sweeps: the rank-local part of the preconditioner is not one graph
        with_direct_output([&] { fieldsplit_apply(c, *c.amg, vp[j], z, 1.0, !direct, !c.amg->global); });
    } else if (hipGraphLaunch(c.iter_graph_pre[j], c.stream) != hipSuccess) {
        hipGetLastError();
        c.iter_graphs_ok = false;
        return false;  // nothing has been communicated yet: the caller repeats the step plainly
    }
    if (!upper && c.amg->global) {  // V-cycle with its collectives
        with_direct_output([&] { c.amg->run(c); });
        if (!direct) fieldsplit_scatter(c, *c.amg, z);
    }
    static const bool overlap = [] {
        const char *e = std::getenv("FEDM_HALO_OVERLAP");
        return !(e && e[0] == '0');
    }();
    if (deep) {
        // z_j is exact on the first ghost layer already
    } else if (overlap) {
        comm_halo_begin(c);
    } else {
        comm_halo(c, z);
    }
    bool ok = !c.iter_graph_interior[j] || hipGraphLaunch(c.iter_graph_interior[j], c.stream) == hipSuccess;
    if (overlap && !deep) comm_halo_exchange(c, z);
    ok = ok && hipGraphLaunch(c.iter_graph[j], c.stream) == hipSuccess;
    if (!ok) {  // z_j is complete on every rank: redo the product with plain launches (same result)
        hipGetLastError();
        c.iter_graphs_ok = false;
        std::vector<const double *> dotp(j + 2);
        for (int i = 0; i <= j; ++i) dotp[i] = vp[i];
        dotp[j + 1] = w;
        launch_spmv(c, z, w, false);
        launch_dots(c, dotp.data(), w, j + 2, true);
        krylov_vector_update(c, j + 1, vp, w);
        return true;
    }
    comm_allreduce(c, c.d_red, j + 2);
    launch_cgs_finish(c, j + 2);
    krylov_vector_update(c, j + 1, vp, w);
    return true;
}

static const bool skip_last_ok = [] {
    const char *e = std::getenv("FEDM_KRYLOV_SKIP_LAST_UPDATE");
    return !(e && e[0] == '0');
}();

// ---- GMRES(m) ---------------------------------------------------------------------------------
// Preconditioner on the left (point-block Jacobi; field split across GPUs): solves
// Minv J delta = Minv rhs (rhs in c.d_rhs, already scaled), convergence on the preconditioned
// residual norm |r| <= max(rtol*|r0|, atol).  Field split on one GPU: on the right, flexible
// (z_j = Minv v_j kept, delta = Z y): rhs is -F itself, one preconditioner application less per
// solve, and the norm tested is that of the true residual.  delta starts at 0; classical
// Gram-Schmidt (PETSc's KSPGMRES default).
static int gmres(Ctx &c, int restart, double rtol, double atol, int max_it, int *its_out,
                 double *rnorm_out, const double *bvec, double bscale, double bnorm_known, double *u_update,
                 bool *u_updated) {
    if (restart < 1 || restart > RED_K - 10) {
        set_error("GMRES restart must be between 1 and 30");
        return -2;
    }
    if (ensure_krylov(c, restart)) return -1;
    const int m = restart;
    std::vector<double> H((size_t)(m + 1) * m, 0.0), cs(m), sn(m), gvec(m + 1), yv(m);
    std::vector<const double *> vp(m + 1), dotp(m + 2), zp(m);
    for (int i = 0; i <= m; ++i) vp[i] = c.d_V + (size_t)i * c.np;
    for (int i = 0; i < m; ++i) zp[i] = c.d_Z + (size_t)i * c.np;
    const bool right = right_preconditioned(c);
    // one GPU, species sweeps, lower-triangular order: the producers of the Krylov vectors form the preconditioner's
    // first stage (krylov_vector_update); for the duration of this solve
    static const bool first_by_producer_ok = [] {
        const char *e = std::getenv("FEDM_FS_FIRST_BY_PRODUCER");
        return !(e && e[0] == '0');
    }();
    struct ProducerMode {
        Ctx &c;
        ~ProducerMode() { c.fs_first_by_producer = false; }
    } producer_mode{c};
    c.fs_first_by_producer = first_by_producer_ok && right && !c.comm && !fieldsplit_upper(c) && c.fs_sweeps > 1 &&
                             c.d_fs_g != nullptr;
    // the (unpreconditioned) operator of the right-preconditioned variant
    auto plain_operator = [&](double *v, double *w) {
        comm_halo(c, v);  // ghost inputs from their owners (multi-GPU)
        prof_begin(c, 1);
        launch_spmv(c, v, w, false);
        prof_end(c);
    };
    // The system is  J delta = bscale * bvec  (bvec = c.d_rhs, already preconditioned, on the left;
    // bvec = F, bscale = -1 on the right, where |bvec| is the |F| the Newton loop has just read:
    // bnorm_known >= 0).  u_update != nullptr: when the solve converges within its first cycle the
    // Newton update u += delta and the norms |delta|^2, |u|^2 (slots 1, 2) are formed by the kernel
    // that forms delta (*u_updated = true); c.d_delta is zeroed only if a generic update needs it.
    bool delta_zeroed = false;
    auto zero_delta = [&] {
        if (!delta_zeroed) hipMemsetAsync(c.d_delta, 0, sizeof(double) * c.np, c.stream);
        delta_zeroed = true;
    };
    if (u_updated) *u_updated = false;
    int its = 0, cycle = 0;
    double r0 = -1.0, rnorm = 0.0;
    bool first = true;
    while (true) {
        // r = rhs - A delta  (delta == 0 on the first cycle)
        double *v0 = c.d_V;
        // (vector copies are kernels of ours: the runtime's blit copy runs at a tenth of the
        // memory bandwidth for these sizes)
        if (!first) {
            if (right) plain_operator(c.d_delta, c.d_w);
            else apply_operator(c, c.d_delta, c.d_w);
            launch_scale_copy(c, bscale, bvec, v0);
            launch_axpy(c, -1.0, c.d_w, v0);
        }
        // First cycle: |rhs| is not waited for -- v0 is normalised on the device and the norm
        // rides along with the first Krylov step's publication (slot RED_SPARE).
        const bool deferred = first && bnorm_known < 0.0;
        double beta = 0.0, tol = 0.0;
        if (first && !deferred) {
            beta = bnorm_known;
            r0 = rnorm = beta;
            tol = std::max(rtol * r0, atol);
            first = false;
            if (beta <= tol) {  // nothing to solve
                zero_delta();
                break;
            }
            krylov_vector_scale(c, bscale / beta, bvec, v0);
        } else if (deferred) {
            launch_norm2(c, bvec, RED_SPARE);
            launch_normalise_copy(c, RED_SPARE, bvec, v0);  // v0 = b / |b|  (bscale is 1 on this path)
            if (c.fs_first_by_producer) fieldsplit_first_stage(c, *c.amg, v0);
            first = false;
        } else {
            launch_norm2(c, v0, 0);
            read_red(c, 1);
            beta = std::sqrt(c.h_red[0]);
            if (!std::isfinite(beta)) {
                *its_out = its;
                *rnorm_out = beta;
                return FEDM_DIVERGED_NAN;
            }
            rnorm = beta;
            tol = std::max(rtol * r0, atol);
            if (beta <= tol || its >= max_it) break;
            krylov_vector_scale(c, 1.0 / beta, v0, v0);
        }
        std::fill(gvec.begin(), gvec.end(), 0.0);
        gvec[0] = beta;
        int j = 0;
        bool done = false;
        std::deque<unsigned long long> queued;
        int update_skipped_for = -1;   // step whose vector has not been orthonormalised yet (launched as 'the last one')
        for (; j < m && its < max_it; ++j) {
            double *w = c.d_V + (size_t)(j + 1) * c.np;
            // classical Gram-Schmidt with ONE reduction and ONE host wait per iteration:
            // h_i = v_i.w and ww = w.w together; |w - V h|^2 = ww - |h|^2 on the device;
            // the update and the normalisation read their coefficients from device memory.
            auto launch_step = [&](int jj, bool skip_update = false) {
                double *ww = c.d_V + (size_t)(jj + 1) * c.np;
                if (right) {
                    double *z = c.d_Z + (size_t)jj * c.np;
                    if (skip_update && iter_graph_launch_right(c, jj, vp.data(), z, ww, true)) {
                        update_skipped_for = jj;
                    } else if (!iter_graph_launch_right(c, jj, vp.data(), z, ww)) {
                        right_step_plain(c, jj, vp.data(), z, ww);
                    }
                } else if (!iter_graph_launch(c, jj, vp.data(), ww)) {
                    apply_operator(c, vp[jj], ww);
                    for (int i = 0; i <= jj; ++i) dotp[i] = vp[i];
                    dotp[jj + 1] = ww;
                    launch_dots(c, dotp.data(), ww, jj + 2, true);
                    krylov_vector_update(c, jj + 1, vp.data(), ww);
                }
                return c.mail_seq;  // the sequence number of this step's publication
            };
            // A Krylov step needs nothing from the host (coefficients and normalisation stay on the
            // device), so the next one can be queued before this one's numbers arrive: the GPU does
            // not idle through the host's round trip and graph launch.  Done while the previous solve
            // says that step will be needed (one GPU: no collectives in between); a step launched in
            // vain only writes vectors nobody reads.
            // `queued`: publications of the steps j, j + 1, ... that are in the queue already.  Two steps go in as one
            // graph whenever two are wanted (between two graph launches the GPU idles for 8 us); a new pair is
            // launched when the queue has run empty, BEFORE this step's numbers are waited for -- so up to three
            // publications may be unread (MAIL_SLOTS).
            auto wanted = [&](int q) {   // step q is expected to be needed: launch it without waiting for step q - 1
                return right && !c.comm && q < m && its + (q - j) < max_it && q < c.krylov_steps_hint;
            };
            // The step expected to end the solve (the previous solve's count; early in a run: the second one) goes in
            // WITHOUT the update that would orthonormalise its vector for a next step: 8 us of kernel nobody needs when
            // the guess is right; when it is wrong the update is launched by itself before the solve goes on.
            auto ends_here = [&](int q) {
                return skip_last_ok && right && !c.comm && c.krylov_steps_hint >= 1 && c.krylov_steps_hint <= 4 &&
                       q + 1 == c.krylov_steps_hint && cycle == 0;
            };
            auto launch_from = [&](int q, bool first_is_needed) {
                if (q > 0 && update_skipped_for == q - 1 && (first_is_needed || wanted(q))) {   // (the guess was wrong)
                    krylov_vector_update(c, q, vp.data(), c.d_V + (size_t)q * c.np);
                    update_skipped_for = -1;
                }
                if ((first_is_needed || wanted(q)) && wanted(q + 1) &&
                    iter_graph_launch_right_pair(c, q, vp.data(), c.d_Z + (size_t)q * c.np, c.d_V + (size_t)(q + 1) * c.np,
                                                 c.d_Z + (size_t)(q + 1) * c.np, c.d_V + (size_t)(q + 2) * c.np,
                                                 ends_here(q + 1))) {
                    if (ends_here(q + 1)) update_skipped_for = q + 1;
                    queued.push_back(c.mail_seq - 1);
                    queued.push_back(c.mail_seq);
                } else if (first_is_needed || wanted(q)) {
                    queued.push_back(launch_step(q, ends_here(q)));
                }
            };
            if (queued.empty()) launch_from(j, true);
            const unsigned long long seq_j = queued.front();
            queued.pop_front();
            if (queued.empty()) launch_from(j + 1, false);
            wait_red_seq(c, seq_j);  // published by the finish kernel: the host works while the update runs
            if (comm_failed(c)) {  // a lost peer is an error, not a NaN
                *its_out = its;
                *rnorm_out = rnorm;
                return -1;
            }
            if (deferred && j == 0) {
                beta = std::sqrt(c.h_red[RED_SPARE]);
                if (!std::isfinite(beta)) {
                    *its_out = its;
                    *rnorm_out = beta;
                    return FEDM_DIVERGED_NAN;
                }
                r0 = rnorm = beta;
                tol = std::max(rtol * r0, atol);
                gvec[0] = beta;
                if (beta <= tol) {  // nothing to solve: delta = 0 (j == 0: no update below)
                    zero_delta();
                    break;
                }
            }
            for (int i = 0; i <= j; ++i) H[(size_t)i * m + j] = c.h_red[i];
            double hn2 = c.h_red[j + 1];
            const double ww = c.h_red[RED_K - 2];
            double hn;
            if (!(hn2 > 1e-8 * ww && hn2 > 0.0) && std::isfinite(ww) && ww > 0.0) {
                // strong cancellation: w was left unscaled; refine (second CGS pass) and
                // take the norm explicitly (a step launched ahead used the unrefined vector: let it
                // finish, its results are dropped and the step is repeated)
                if (!queued.empty()) {
                    wait_red_seq(c, queued.back());
                    queued.clear();
                    // a dropped step that went in 'as the last one' is launched again from scratch: its skipped
                    // update must not be made up for a second time behind the relaunch
                    if (update_skipped_for > j) update_skipped_for = -1;
                }
                if (update_skipped_for == j) {   // the first Gram-Schmidt pass has not been applied to w yet
                    krylov_vector_update(c, j + 1, vp.data(), w);
                    update_skipped_for = -1;
                }
                launch_dots(c, vp.data(), w, j + 1, false);
                read_red(c, j + 1);
                for (int i = 0; i <= j; ++i) H[(size_t)i * m + j] += c.h_red[i];
                launch_multi_axpy(c, c.h_red, j + 1, vp.data(), w, -1.0);
                launch_norm2(c, w, 0);
                read_red(c, 1);
                hn = std::sqrt(c.h_red[0]);
                if (hn > 0.0 && std::isfinite(hn)) krylov_vector_scale(c, 1.0 / hn, w, w);
            } else {
                hn = std::sqrt(hn2);
            }
            H[(size_t)(j + 1) * m + j] = hn;
            if (!std::isfinite(hn)) {
                *its_out = its;
                *rnorm_out = hn;
                return FEDM_DIVERGED_NAN;
            }
            // Givens rotations on column j
            for (int i = 0; i < j; ++i) {
                const double t = cs[i] * H[(size_t)i * m + j] + sn[i] * H[(size_t)(i + 1) * m + j];
                H[(size_t)(i + 1) * m + j] = -sn[i] * H[(size_t)i * m + j] + cs[i] * H[(size_t)(i + 1) * m + j];
                H[(size_t)i * m + j] = t;
            }
            const double a = H[(size_t)j * m + j], b = H[(size_t)(j + 1) * m + j];
            const double d = std::hypot(a, b);
            cs[j] = d > 0.0 ? a / d : 1.0;
            sn[j] = d > 0.0 ? b / d : 0.0;
            H[(size_t)j * m + j] = d;
            H[(size_t)(j + 1) * m + j] = 0.0;
            gvec[j + 1] = -sn[j] * gvec[j];
            gvec[j] = cs[j] * gvec[j];
            ++its;
            rnorm = std::fabs(gvec[j + 1]);
            if (rnorm <= tol || hn == 0.0) {
                ++j;
                done = true;
                break;
            }
        }
        // back substitution, delta += V y
        const int k = j;
        for (int i = k - 1; i >= 0; --i) {
            double s = gvec[i];
            for (int l = i + 1; l < k; ++l) s -= H[(size_t)i * m + l] * yv[l];
            yv[i] = s / H[(size_t)i * m + i];
        }
        if (k > 0) {
            if (u_update && done && cycle == 0 && k <= 8) {
                launch_newton_update(c, yv.data(), k, right ? zp.data() : vp.data(), u_update, nullptr);
                *u_updated = true;
            } else {
                zero_delta();
                launch_multi_axpy(c, yv.data(), k, right ? zp.data() : vp.data(), c.d_delta, 1.0);
            }
        }
        ++cycle;
        if (done || its >= max_it) {
            if (!done) {  // recompute the true (preconditioned, on the left) residual for the report
                if (right) plain_operator(c.d_delta, c.d_w);
                else apply_operator(c, c.d_delta, c.d_w);
                launch_axpy(c, -bscale, bvec, c.d_w);
                launch_norm2(c, c.d_w, 0);
                read_red(c, 1);
                rnorm = std::sqrt(c.h_red[0]);
            }
            break;
        }
    }
    c.krylov_steps_hint = its;
    *its_out = its;
    *rnorm_out = rnorm;
    const double tol = std::max(rtol * r0, atol);
    return rnorm <= tol ? 0 : FEDM_DIVERGED_LINEAR;
}

static int eval_residual(Ctx &c, int mode, double *fnorm) {
    launch_assemble(c, false, mode);
    launch_finalize(c, false, mode);
    launch_norm2(c, c.d_F, 0);
    read_red(c, 1);
    *fnorm = std::sqrt(c.h_red[0]);
    return 0;
}

static void eval_jacobian(Ctx &c, int mode) {
    launch_assemble(c, true, mode);
    launch_finalize(c, true, mode);
}

// ---- host-side sparse algebra for the composite multigrid levels (amg.hpp) ------------------
namespace {
struct HostCsr {
    int n_rows = 0, n_cols = 0;
    std::vector<int64_t> indptr;
    std::vector<int32_t> indices;
    std::vector<double> values;
    fedm_csr view() const { return fedm_csr{n_rows, n_cols, indptr.data(), indices.data(), values.data()}; }
};

// alpha * A * diag(dc) * B  (dc may be null), rows merged with a dense accumulator, columns sorted
HostCsr csr_product(const fedm_csr &A, const double *dc, const fedm_csr &B, double alpha) {
    HostCsr out;
    out.n_rows = A.n_rows;
    out.n_cols = B.n_cols;
    out.indptr.assign((size_t)A.n_rows + 1, 0);
    std::vector<double> acc((size_t)B.n_cols, 0.0);
    std::vector<char> seen((size_t)B.n_cols, 0);
    std::vector<int32_t> cols;
    for (int i = 0; i < A.n_rows; ++i) {
        cols.clear();
        for (int64_t k = A.indptr[i]; k < A.indptr[i + 1]; ++k) {
            const int j = A.indices[k];
            const double a = alpha * A.values[k] * (dc ? dc[j] : 1.0);
            for (int64_t q = B.indptr[j]; q < B.indptr[j + 1]; ++q) {
                const int cidx = B.indices[q];
                if (!seen[cidx]) {
                    seen[cidx] = 1;
                    cols.push_back(cidx);
                }
                acc[cidx] += a * B.values[q];
            }
        }
        std::sort(cols.begin(), cols.end());
        for (int32_t cidx : cols) {
            out.indices.push_back(cidx);
            out.values.push_back(acc[cidx]);
            acc[cidx] = 0.0;
            seen[cidx] = 0;
        }
        out.indptr[i + 1] = (int64_t)out.indices.size();
    }
    return out;
}

// diag(dl) * (alpha * A + beta * B) with B's columns shifted by `shift` into a matrix of n_cols
// columns (A and B may overlap in pattern when shift == 0); dl may be null
HostCsr csr_combine(const fedm_csr &A, double alpha, const fedm_csr &B, double beta, int shift, int n_cols,
                    const double *dl) {
    HostCsr out;
    out.n_rows = A.n_rows;
    out.n_cols = n_cols;
    out.indptr.assign((size_t)A.n_rows + 1, 0);
    std::vector<std::pair<int32_t, double>> row;
    for (int i = 0; i < A.n_rows; ++i) {
        row.clear();
        const double s = dl ? dl[i] : 1.0;
        for (int64_t k = A.indptr[i]; k < A.indptr[i + 1]; ++k) row.emplace_back(A.indices[k], s * alpha * A.values[k]);
        for (int64_t k = B.indptr[i]; k < B.indptr[i + 1]; ++k)
            row.emplace_back(B.indices[k] + shift, s * beta * B.values[k]);
        std::sort(row.begin(), row.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
        for (size_t k = 0; k < row.size(); ++k) {
            if (!out.indices.empty() && (int64_t)out.indices.size() > out.indptr[i] && out.indices.back() == row[k].first)
                out.values.back() += row[k].second;
            else {
                out.indices.push_back(row[k].first);
                out.values.push_back(row[k].second);
            }
        }
        out.indptr[i + 1] = (int64_t)out.indices.size();
    }
    return out;
}

HostCsr csr_identity(int n) {
    HostCsr out;
    out.n_rows = out.n_cols = n;
    out.indptr.resize((size_t)n + 1);
    out.indices.resize(n);
    out.values.assign(n, 1.0);
    for (int i = 0; i <= n; ++i) out.indptr[i] = i;
    for (int i = 0; i < n; ++i) out.indices[i] = i;
    return out;
}
}  // namespace

}  // namespace fedm

using namespace fedm;

extern "C" {

const char *fedm_last_error(void) { return g_error.c_str(); }
int fedm_abi_version(void) { return FEDM_ABI_VERSION; }

static int ctx_create_impl(const fedm_mesh_desc *mesh, const fedm_model_desc *model,
                           const fedm_gd_desc *gd, int device, fedm_ctx **out);

// every error exit of the set-up releases what had been allocated so far (context, stream, device
// and pinned memory): a caller that retries after an out-of-memory must not accumulate leaked HBM
static int ctx_create_guarded(const fedm_mesh_desc *mesh, const fedm_model_desc *model,
                              const fedm_gd_desc *gd, int device, fedm_ctx **out) {
    *out = nullptr;
    fedm_ctx *h = nullptr;
    const int rc = ctx_create_impl(mesh, model, gd, device, &h);
    if (rc != 0) {
        const std::string msg = g_error;  // destroy may overwrite it
        if (h) fedm_ctx_destroy(h);
        g_error = msg;
        return rc;
    }
    *out = h;
    return 0;
}

int fedm_ctx_create(const fedm_mesh_desc *mesh, const fedm_model_desc *model, int device,
                    fedm_ctx **out) {
    if (!mesh || !model || !out) {
        set_error("null argument");
        return -2;
    }
    if (check_model(*model)) {
        // refused, never truncated: the LFA kernels are instantiated for 1-2 species without and 1-4 species
        // with a Poisson equation (fedm_model_desc's arrays hold FEDM_MAX_SPECIES = 4 and FEDM_MAX_REACTIONS = 8)
        set_error("unsupported LFA model: " + std::to_string(model->n_species) + " species" +
                  (model->poisson ? " + Poisson" : "") + ", " + std::to_string(model->n_reactions) +
                  " reactions, " + std::to_string(model->n_qp) + " quadrature points (supported: 1-2 species, or 1-4 "
                  "with a Poisson equation; at most " + std::to_string(FEDM_MAX_REACTIONS) + " reactions, " +
                  std::to_string(FEDM_MAX_TERMS) + " terms per coefficient, " + std::to_string(FEDM_MAX_QP) +
                  " quadrature points)");
        return -2;
    }
    return ctx_create_guarded(mesh, model, nullptr, device, out);
}

int fedm_ctx_create_gd(const fedm_mesh_desc *mesh, const fedm_gd_desc *gd, int device,
                       fedm_ctx **out) {
    if (!mesh || !gd || !out) {
        set_error("null argument");
        return -2;
    }
    if (gd->n_species < 2 || gd->n_species > FEDM_GD_MAX_SPECIES - 1 || gd->n_reactions < 0 ||
        gd->n_reactions > FEDM_GD_MAX_REACTIONS || gd->n_qp < 1 || gd->n_qp > FEDM_MAX_QP ||
        gd->n_fqp < 0 || gd->n_fqp > FEDM_MAX_FQP || gd->n_tags < 0 || gd->n_tags > FEDM_MAX_TAGS) {
        set_error("unsupported LMEA model descriptor");
        return -2;
    }
    for (int j = 0; j < gd->n_reactions; ++j)
        for (int i = 0; i < gd->n_species; ++i)
            if (gd->power[j][i] < 0 || gd->power[j][i] > 15) {   // the kernels pack them 4 bits each
                set_error("LMEA reaction powers must be between 0 and 15");
                return -2;
            }
    return ctx_create_guarded(mesh, nullptr, gd, device, out);
}

int fedm_gd_prep_setup(fedm_ctx *h, const fedm_csr *mass, int n_tables, const int32_t *tab_ptr,
                       const double *tab_x, const double *tab_y, const fedm_gd_field_prog *progs) {
    Ctx &c = h->c;
    if (c.model_kind != 1 || !mass || !tab_ptr || !progs || n_tables < 0 || mass->n_rows != c.nv) {
        set_error("bad LMEA field-refresh description");
        return -2;
    }
    for (int r = 0; r < c.gd_n_fields; ++r)
        if ((progs[r].kind == FEDM_GDP_TABLE && (progs[r].table < 0 || progs[r].table >= n_tables)) ||
            (progs[r].kind == FEDM_GDP_SCALED_ROW && (progs[r].src_row < 0 || progs[r].src_row >= c.gd_n_fields))) {
            set_error("field program refers to a missing table or row");
            return -2;
        }
    FEDM_HIP_CHECK(hipSetDevice(c.device));
    return gd_prep_setup(c, mass, n_tables, tab_ptr, tab_x, tab_y, progs);
}

int fedm_gd_prep_step(fedm_ctx *h) {
    Ctx &c = h->c;
    if (c.model_kind != 1 || !c.gd_prep) {
        set_error("fedm_gd_prep_setup has not been called");
        return -2;
    }
    FEDM_HIP_CHECK(hipSetDevice(c.device));
    const int rc = gd_prep_step(c);
    FEDM_HIP_CHECK(hipStreamSynchronize(c.stream));
    return rc;
}

int fedm_gd_update_mean_energy(fedm_ctx *h) {
    Ctx &c = h->c;
    if (c.model_kind != 1) {
        set_error("not an LMEA context");
        return -2;
    }
    FEDM_HIP_CHECK(hipSetDevice(c.device));
    gd_update_mean_energy(c);
    return 0;
}

int fedm_gd_get_fields(fedm_ctx *h, double *out) {
    Ctx &c = h->c;
    if (c.model_kind != 1 || !out) {
        set_error("not an LMEA context");
        return -2;
    }
    FEDM_HIP_CHECK(hipSetDevice(c.device));
    FEDM_HIP_CHECK(hipStreamSynchronize(c.stream));
    FEDM_HIP_CHECK(hipMemcpy(out, c.d_gd_fields, sizeof(double) * (size_t)c.gd_n_fields * c.nv, hipMemcpyDeviceToHost));
    return 0;
}

int fedm_gd_set_fields(fedm_ctx *h, const double *fields) {
    Ctx &c = h->c;
    if (c.model_kind != 1 || !fields) {
        set_error("not an LMEA context");
        return -2;
    }
    FEDM_HIP_CHECK(hipSetDevice(c.device));
    FEDM_HIP_CHECK(hipMemcpyAsync(c.d_gd_fields, fields, sizeof(double) * (size_t)c.gd_n_fields * c.nv,
                                  hipMemcpyHostToDevice, c.stream));
    FEDM_HIP_CHECK(hipStreamSynchronize(c.stream));
    return 0;
}

static int ctx_create_impl(const fedm_mesh_desc *mesh, const fedm_model_desc *model,
                           const fedm_gd_desc *gd, int device, fedm_ctx **out) {
    if (mesh->n_vertices < 3 || mesh->n_cells < 1) {
        set_error("empty mesh");
        return -2;
    }
    for (int i = 0; i < 3 * mesh->n_cells; ++i)
        if (mesh->cells[i] < 0 || mesh->cells[i] >= mesh->n_vertices) {
            set_error("cell vertex index out of range");
            return -2;
        }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("no HIP device available: libfedm_hip has no CPU fallback");
        return -3;
    }
    FEDM_HIP_CHECK(hipSetDevice(device));
    fedm_ctx *h = new fedm_ctx();
    *out = h;  // from here on the caller (ctx_create_guarded) owns it, error exits included
    Ctx &c = h->c;
    c.device = device;
    const int n_tags_model = model ? model->n_tags : gd->n_tags;
    if (model) {
        c.model = *model;
        c.ns = model->n_species;
        c.poisson = model->poisson != 0;
    } else {  // LMEA: energy + (n_species - 1) particle equations + potential
        c.model_kind = 1;
        c.gd = *gd;
        c.ns = gd->n_species;
        c.poisson = true;
        c.assembly_kind = 0;
    }
    c.neq = c.ns + (c.poisson ? 1 : 0);
    c.nv = mesh->n_vertices;
    c.nc = mesh->n_cells;
    {
        // (Expression sources are tables indexed by (cell, local node): their cells keep their vertex order)
        bool tables = false;
        if (model)
            for (int s_ = 0; s_ < model->n_species; ++s_) tables = tables || model->ext_nodes[s_] > 0;
        build_pattern(*mesh, c.pat, !tables);
    }
    c.nvp = c.pat.nvp;
    c.n_owned = (mesh->n_owned_vertices > 0 && mesh->n_owned_vertices <= c.nv) ? mesh->n_owned_vertices : c.nv;
    c.n_dot = (int64_t)c.n_owned * c.neq;
    c.halo_depth = 1;
    if (mesh->halo_depth > 1 && (mesh->identity_vertices || mesh->n_identity_vertices == 0)) {
        for (int i = 0; i < mesh->n_identity_vertices; ++i)
            if (mesh->identity_vertices[i] < c.n_owned || mesh->identity_vertices[i] >= c.nv) {
                set_error("identity_vertices must be ghost vertices");
                return -2;
            }
        c.halo_depth = mesh->halo_depth;
        c.n_identity = mesh->n_identity_vertices;
        if (upload(c.d_identity, mesh->identity_vertices, (size_t)c.n_identity)) return -1;
    }
    c.n = (int64_t)c.nv * c.neq;
    c.np = (int64_t)c.nvp * c.neq;
    for (int i = 0; i < mesh->n_dirichlet; ++i)
        if (mesh->dirichlet_dofs[i] < 0 || mesh->dirichlet_dofs[i] >= c.n) {
            set_error("Dirichlet dof out of range");
            return -2;
        }
    FEDM_HIP_CHECK(hipStreamCreate(&c.stream));
    if (upload(c.d_coords, mesh->coords, (size_t)2 * c.nv)) return -1;
    if (upload(c.d_cells, mesh->cells, (size_t)3 * c.nc)) return -1;
    std::vector<int8_t> zero_tags;
    const int8_t *tags = mesh->facet_tags;
    if (!tags) {
        zero_tags.assign((size_t)3 * c.nc, 0);
        tags = zero_tags.data();
    }
    for (size_t i = 0; i < (size_t)3 * c.nc; ++i)
        if (tags[i] < 0 || tags[i] > n_tags_model) {
            set_error("facet tag out of range");
            return -2;
        }
    if (upload(c.d_ftags, tags, (size_t)3 * c.nc)) return -1;
    {
        // tagged boundary facets, greedily coloured so that facets of one colour share no
        // vertex of their cells: the boundary kernel can then add without atomics, in a
        // fixed order (bitwise reproducible)
        std::vector<int> fcell, floc, ftag, fcol;
        std::vector<uint32_t> used(c.nv, 0);
        int ncol = 0;
        for (int cell = 0; cell < c.nc; ++cell)
            for (int i = 0; i < 3; ++i)
                if (tags[3 * cell + i] > 0) {
                    const int32_t *v = mesh->cells + 3 * cell;
                    const uint32_t m = used[v[0]] | used[v[1]] | used[v[2]];
                    int k = 0;
                    while (k < 31 && ((m >> k) & 1u)) ++k;
                    for (int a = 0; a < 3; ++a) used[v[a]] |= (1u << k);
                    ncol = std::max(ncol, k + 1);
                    fcell.push_back(cell);
                    floc.push_back(i);
                    ftag.push_back(tags[3 * cell + i]);
                    fcol.push_back(k);
                }
        std::vector<int> bf;
        c.bfacet_colour_ptr.assign(ncol + 1, 0);
        for (int k = 0; k < ncol; ++k) {
            for (size_t f = 0; f < fcell.size(); ++f)
                if (fcol[f] == k) {
                    bf.push_back(fcell[f]);
                    bf.push_back(floc[f]);
                    bf.push_back(ftag[f]);
                }
            c.bfacet_colour_ptr[k + 1] = (int)bf.size() / 3;
        }
        c.n_bfacets = (int)bf.size() / 3;
        if (upload(c.d_bfacets, bf.data(), bf.size())) return -1;
        // Do the boundary facets (species rows of their cells' vertices) and the Dirichlet dofs meet in a row?  If not
        // -- the scripts put Dirichlet values on the potential only -- the two can share a launch (launch_finalize)
        std::vector<char> is_dirichlet((size_t)c.nv * c.neq, 0);
        for (int k = 0; k < mesh->n_dirichlet; ++k)
            if (mesh->dirichlet_dofs[k] >= 0 && (int64_t)mesh->dirichlet_dofs[k] < (int64_t)c.nv * c.neq)
                is_dirichlet[mesh->dirichlet_dofs[k]] = 1;
        c.boundary_rows_disjoint = true;
        for (size_t f = 0; f < fcell.size() && c.boundary_rows_disjoint; ++f)
            for (int a = 0; a < 3; ++a)
                for (int sp = 0; sp < c.ns; ++sp)
                    if (is_dirichlet[(size_t)mesh->cells[3 * fcell[f] + a] * c.neq + sp]) c.boundary_rows_disjoint = false;
        // the rows that change behind the volume assembly (Ctx::planes_fused): vertices of boundary-facet cells, of
        // Dirichlet values, and the padding of the last slice
        {
            std::vector<char> touched((size_t)c.nvp, 0);
            for (size_t f = 0; f < fcell.size(); ++f)
                for (int a = 0; a < 3; ++a) touched[mesh->cells[3 * fcell[f] + a]] = 1;
            for (int k = 0; k < mesh->n_dirichlet; ++k)
                if (mesh->dirichlet_dofs[k] >= 0 && (int64_t)mesh->dirichlet_dofs[k] < (int64_t)c.nv * c.neq)
                    touched[mesh->dirichlet_dofs[k] / c.neq] = 1;
            for (int v = c.nv; v < c.nvp; ++v) touched[v] = 1;
            std::vector<int> rows;
            for (int v = 0; v < c.nvp; ++v)
                if (touched[v]) rows.push_back(v);
            c.n_planes_rows = (int)rows.size();
            if (c.n_planes_rows && upload(c.d_planes_rows, rows.data(), rows.size())) return -1;
            const char *e = getenv("FEDM_PLANES_FUSED");
            // one GPU, whole-mesh launches: no identity rows of ghost layers, no listed part launches
            // (opt-in, FEDM_PLANES_FUSED=1: measured -9 us per assembly back to back, nothing in the bench's step, and the
            // dominant kernel 75 -> 89 us: DESIGN.md Appendix A)
            c.planes_fuse_ok = (e && e[0] == '1') && mesh->n_identity_vertices == 0 && c.n_owned == c.nv;
        }
    }
    if (upload(c.d_cell_slots, c.pat.cell_slots.data(), c.pat.cell_slots.size())) return -1;
    if (upload(c.d_colour_cells, c.pat.colour_cells.data(), c.pat.colour_cells.size())) return -1;
    if (model) {
        if (upload(c.d_model, model, 1)) return -1;
    } else {
        if (upload(c.d_gd, gd, 1)) return -1;
        c.gd_n_fields = FEDM_GD_N_FIELDS(gd->n_species, gd->n_reactions);
        if (alloc_zero(c.d_gd_fields, (size_t)c.gd_n_fields * c.nv, c.stream)) return -1;
    }
    if (upload(c.d_patch_cell_ptr, c.pat.patch_cell_ptr.data(), c.pat.patch_cell_ptr.size())) return -1;
    if (upload(c.d_patch_halo_ptr, c.pat.patch_halo_ptr.data(), c.pat.patch_halo_ptr.size())) return -1;
    if (upload(c.d_patch_halo, c.pat.patch_halo.data(), c.pat.patch_halo.size())) return -1;
    if (upload(c.d_patch_cells, c.pat.patch_cells.data(), c.pat.patch_cells.size())) return -1;
    {
        // LDS patches need 8-bit local indices and <= 160 KiB of LDS per workgroup; the
        // globally coloured kernel is the (deterministic, slower) alternative.
        const char *env = getenv("FEDM_ASSEMBLY");
        c.assembly_kind = (env && std::string(env) == "colour") ? 0 : 1;
        // Side of the field split: right for the LFA systems (3 instead of 5.25 Krylov steps per
        // Newton iteration on the streamer case); left for LMEA, whose rows (energy balance next to
        // densities) are scaled so differently that the true residual norm is the harder target
        // (glow discharge, 402k DOFs: 70 against 100 steps per time step).  FEDM_PRECOND_SIDE or
        // fedm_set_preconditioner_side override.
        c.right_precond = c.model_kind == 0;
        if (const char *lean = getenv("FEDM_ASSEMBLY_LEAN")) c.assembly_lean = lean[0] == '0' ? 0 : (lean[0] == '3' ? 3 : 2);
        if (const char *e = getenv("FEDM_XCD_REMAP")) c.xcd_remap = e[0] != '0';
        if (const char *e = getenv("FEDM_ASSEMBLY_OVERLAP")) c.assembly_overlap = e[0] != '0';
        if (const char *e = getenv("FEDM_SKIP_CONST_PLANES")) c.skip_const_planes = e[0] != '0';
        if (model && c.poisson) {
            // potential-potential: geometry only.  Species (s, i), i != s: zero unless a reaction that
            // changes species s has species i among its reactants (fedm/functions.py:835-843).
            c.const_plane_mask = 1u << (c.ns * c.neq + c.ns);
            for (int s_ = 0; s_ < c.ns; ++s_)
                for (int i = 0; i < c.ns; ++i) {
                    if (i == s_) continue;
                    bool coupled = false;
                    for (int j = 0; j < model->n_reactions; ++j)
                        coupled = coupled || (model->net[j][s_] != 0 && model->power[j][i] > 0);
                    if (!coupled) c.const_plane_mask |= 1u << (s_ * c.neq + i);
                }
            c.zero_plane_mask = c.const_plane_mask & ~(1u << (c.ns * c.neq + c.ns));
            if (const char *e = getenv("FEDM_SPMV_SKIP_ZERO_PLANES"))
                if (e[0] == '0') c.zero_plane_mask = 0;
        }
        if (const char *e = getenv("FEDM_FS_HALO")) c.fs_halo = e[0] != '0';
        if (const char *e = getenv("FEDM_DEEP_HALO")) c.deep_halo = e[0] != '0';
        if (const char *e = getenv("FEDM_FS_POLICY")) c.fs_measured_policy = std::string(e) != "counts";
        if (const char *e = getenv("FEDM_FS_LAGGED_COUPLING")) c.fs_lagged_coupling = e[0] != '0';
        if (const char *e = getenv("FEDM_GD_HAND"))
            if (e[0] == '0' || (e[0] >= '2' && e[0] <= '5')) c.gd_hand_mode = e[0] - '0';
        if (const char *e = getenv("FEDM_FS_ORDER")) c.fs_upper = std::string(e) == "upper";
        const char *side = getenv("FEDM_PRECOND_SIDE");
        if (side && std::string(side) == "left") c.right_precond = false;
        if (side && std::string(side) == "right") c.right_precond = true;
        if (!c.pat.patch_ok || patch_lds_bytes(c) > 160 * 1024 || c.model_kind == 1) c.assembly_kind = 0;
    }
    if (upload(c.d_slice_boff, c.pat.slice_boff.data(), c.pat.slice_boff.size())) return -1;
    if (upload(c.d_colidx, c.pat.colidx.data(), c.pat.colidx.size())) return -1;
    if (upload(c.d_diag_slot, c.pat.diag_slot.data(), c.pat.diag_slot.size())) return -1;
    c.n_dir = mesh->n_dirichlet;
    if (upload(c.d_dir_dofs, mesh->dirichlet_dofs, (size_t)c.n_dir)) return -1;
    if (upload(c.d_dir_vals, mesh->dirichlet_vals, (size_t)c.n_dir)) return -1;
    const size_t nval = (size_t)c.pat.total_bc * SLICE * c.neq * c.neq;
    if (alloc_zero(c.d_val, nval, c.stream)) return -1;
    if (alloc_zero(c.d_dinv, (size_t)c.nvp * c.neq * c.neq, c.stream)) return -1;
    double **vecs[] = {&c.d_u, &c.d_uold, &c.d_uold1, &c.d_F, &c.d_delta, &c.d_w, &c.d_rhs, &c.d_tmp, &c.d_fs, &c.d_fs_g};
    for (auto v : vecs)
        if (alloc_zero(*v, (size_t)c.np, c.stream)) return -1;
    if (alloc_zero(c.d_partials, (size_t)RED_BLOCKS * RED_K, c.stream)) return -1;
    if (alloc_zero(c.d_red, RED_K, c.stream)) return -1;
    FEDM_HIP_CHECK(hipHostMalloc((void **)&c.h_mail, sizeof(double) * MAIL_SLOTS * (RED_K + 1), hipHostMallocDefault));
    std::memset(c.h_mail, 0, sizeof(double) * MAIL_SLOTS * (RED_K + 1));
    c.h_red = c.h_mail;
    FEDM_HIP_CHECK(hipMalloc((void **)&c.d_mail_seq, sizeof(unsigned long long)));
    FEDM_HIP_CHECK(hipMemset(c.d_mail_seq, 0, sizeof(unsigned long long)));

    FEDM_HIP_CHECK(hipHostMalloc((void **)&c.h_stage, sizeof(double) * (size_t)c.np));
    for (int s = 0; model && s < c.ns; ++s)
        if (model->ext_nodes[s] > 0)
            if (alloc_zero(c.d_ext[s], (size_t)c.nc * model->ext_nodes[s], c.stream)) return -1;
    FEDM_HIP_CHECK(hipStreamSynchronize(c.stream));
    return 0;
}

void fedm_ctx_destroy(fedm_ctx *h) {
    if (!h) return;
    Ctx &c = h->c;
    hipSetDevice(c.device);
    // a failed transport first: its communicator is aborted before anything below waits for the device
    // (comm.hip, Comm::release_communicator); the error stays readable through fedm_last_error
    if (c.comm) {
        comm_poll_async_error(c);
        if (c.comm->failed && c.comm->release_communicator())
            set_error(c.comm->error + " -- communicator aborted at teardown");
    }
    if (c.stream) hipStreamSynchronize(c.stream);
    void *ptrs[] = {c.d_coords, c.d_cells, c.d_ftags, c.d_cell_slots, c.d_colour_cells, c.d_model,
                    c.d_slice_boff, c.d_colidx, c.d_diag_slot, c.d_val, c.d_dinv, c.d_dir_dofs,
                    c.d_dir_vals, c.d_identity, c.d_u, c.d_uold, c.d_uold1, c.d_F, c.d_delta, c.d_w, c.d_rhs,
                    c.d_tmp, c.d_fs, c.d_fs_g, c.d_V, c.d_partials, c.d_partials_wide, c.d_red, c.d_ext[0], c.d_ext[1], c.d_ext[2],
                    c.d_ext[3], c.d_patch_cell_ptr, c.d_patch_halo_ptr, c.d_patch_halo,
                    c.d_patch_cells, c.d_bfacets, c.d_gd, c.d_gd_fields, c.d_gd_elem, c.d_gd_inv_ptr,
                    c.d_gd_inv_idx, c.d_gd_elemF, c.d_gd_vinv_ptr, c.d_gd_vinv_idx, c.d_gd_kpos};
    for (void *p : ptrs)
        if (p) hipFree(p);
    for (Amg *a : {c.amg, c.amg_alt})
        if (a) {
            a->release();
            delete a;
        }
    if (c.comm) {
        c.comm->release();
        delete c.comm;
    }
    gd_prep_release(c);
    fs_tiles_release(c);
    for (auto &e : c.prof.ev) hipEventDestroy(e);
    iter_graphs_clear(c);
    if (c.d_mail_seq) hipFree(c.d_mail_seq);
    if (c.h_mail) hipHostFree(c.h_mail);
    if (c.d_val32) hipFree(c.d_val32);
    if (c.d_s16) hipFree(c.d_s16);
    if (c.d_planes_rows) hipFree(c.d_planes_rows);
    if (c.d_Z) hipFree(c.d_Z);
    if (c.h_stage) hipHostFree(c.h_stage);
    if (c.d_snapshot) hipFree(c.d_snapshot);
    lean3_release(c);
    for (int s_ = 0; s_ < FEDM_MAX_SPECIES; ++s_) {
        if (c.d_expr_ops[s_]) hipFree(c.d_expr_ops[s_]);
        if (c.d_expr_consts[s_]) hipFree(c.d_expr_consts[s_]);
    }
    if (c.stream) hipStreamDestroy(c.stream);
    delete h;
}

static int put_vec(Ctx &c, double *dst, const double *src) {
    if (!src) return 0;
    std::memcpy(c.h_stage, src, sizeof(double) * c.n);
    FEDM_HIP_CHECK(hipMemcpyAsync(dst, c.h_stage, sizeof(double) * c.n, hipMemcpyHostToDevice, c.stream));
    FEDM_HIP_CHECK(hipStreamSynchronize(c.stream));
    return 0;
}

static int get_vec(Ctx &c, double *dst, const double *src) {
    FEDM_HIP_CHECK(hipMemcpyAsync(c.h_stage, src, sizeof(double) * c.n, hipMemcpyDeviceToHost, c.stream));
    FEDM_HIP_CHECK(hipStreamSynchronize(c.stream));
    std::memcpy(dst, c.h_stage, sizeof(double) * c.n);
    return 0;
}

int fedm_set_state(fedm_ctx *h, const double *u_new, const double *u_old, const double *u_old1) {
    Ctx &c = h->c;
    c.err_cache_comp = -1;   // the state changes: the kept error norm is stale
    FEDM_HIP_CHECK(hipSetDevice(c.device));
    if (u_new) c.halo_pending = false;  // the caller's vector carries its own ghost values
    if (put_vec(c, c.d_u, u_new) || put_vec(c, c.d_uold, u_old) || put_vec(c, c.d_uold1, u_old1)) return -1;
    return 0;
}

int fedm_get_state(fedm_ctx *h, double *u_new) {
    Ctx &c = h->c;
    FEDM_HIP_CHECK(hipSetDevice(c.device));
    return get_vec(c, u_new, c.d_u);
}

int fedm_get_state_old(fedm_ctx *h, double *u_old) {
    Ctx &c = h->c;
    FEDM_HIP_CHECK(hipSetDevice(c.device));
    return get_vec(c, u_old, c.d_uold);
}

int fedm_shift_state(fedm_ctx *h) {
    Ctx &c = h->c;
    c.err_cache_comp = -1;   // the state changes: the kept error norm is stale
    FEDM_HIP_CHECK(hipSetDevice(c.device));
    std::swap(c.d_uold1, c.d_uold);  // old1 <- old (by rotation), then old <- new
    launch_scale_copy(c, 1.0, c.d_u, c.d_uold);
    return 0;
}

int fedm_reset_state(fedm_ctx *h) {
    Ctx &c = h->c;
    c.err_cache_comp = -1;   // the state changes: the kept error norm is stale
    FEDM_HIP_CHECK(hipSetDevice(c.device));
    launch_scale_copy(c, 1.0, c.d_uold, c.d_u);
    return 0;
}

int fedm_state_snapshot(fedm_ctx *h) {
    Ctx &c = h->c;
    FEDM_HIP_CHECK(hipSetDevice(c.device));
    if (c.halo_pending) {   // (several GPUs: the ghost entries of the state are exchanged lazily)
        comm_halo(c, c.d_u);
        c.halo_pending = false;
    }
    if (!c.d_snapshot) FEDM_HIP_CHECK(hipMalloc((void **)&c.d_snapshot, sizeof(double) * 3 * c.np));
    const double *src[3] = {c.d_u, c.d_uold, c.d_uold1};
    for (int k = 0; k < 3; ++k)
        FEDM_HIP_CHECK(hipMemcpyAsync(c.d_snapshot + (size_t)k * c.np, src[k], sizeof(double) * c.np,
                                      hipMemcpyDeviceToDevice, c.stream));
    FEDM_HIP_CHECK(hipStreamSynchronize(c.stream));
    // ... and what the solver has learnt from the steps before (how many Krylov steps to queue ahead, at which Newton
    // iteration the final check is expected): steps repeated from the checkpoint then run as they did the first time
    c.snap_krylov_steps_hint = c.krylov_steps_hint;
    c.snap_newton_its_hint = c.newton_its_hint;
    return 0;
}

int fedm_state_restore(fedm_ctx *h) {
    Ctx &c = h->c;
    if (!c.d_snapshot) {
        set_error("fedm_state_restore: no snapshot taken");
        return -2;
    }
    c.err_cache_comp = -1;
    c.krylov_steps_hint = c.snap_krylov_steps_hint;
    c.newton_its_hint = c.snap_newton_its_hint;
    FEDM_HIP_CHECK(hipSetDevice(c.device));
    c.halo_pending = false;   // the snapshot was taken with exchanged ghosts
    double *dst[3] = {c.d_u, c.d_uold, c.d_uold1};
    for (int k = 0; k < 3; ++k)
        FEDM_HIP_CHECK(hipMemcpyAsync(dst[k], c.d_snapshot + (size_t)k * c.np, sizeof(double) * c.np,
                                      hipMemcpyDeviceToDevice, c.stream));
    FEDM_HIP_CHECK(hipStreamSynchronize(c.stream));
    return 0;
}

int fedm_set_step(fedm_ctx *h, double dt, double dt_old) {
    h->c.dt = dt;
    h->c.dt_old = dt_old;
    return 0;
}

int fedm_set_dirichlet_values(fedm_ctx *h, const double *vals) {
    Ctx &c = h->c;
    FEDM_HIP_CHECK(hipSetDevice(c.device));
    if (c.n_dir)
        FEDM_HIP_CHECK(hipMemcpy(c.d_dir_vals, vals, sizeof(double) * c.n_dir, hipMemcpyHostToDevice));
    return 0;
}

int fedm_set_ext_source(fedm_ctx *h, int species, const double *nodal) {
    Ctx &c = h->c;
    if (species < 0 || species >= c.ns || !c.d_ext[species]) {
        set_error("species has no Expression source");
        return -2;
    }
    FEDM_HIP_CHECK(hipSetDevice(c.device));
    FEDM_HIP_CHECK(hipMemcpyAsync(c.d_ext[species], nodal,
                                  sizeof(double) * (size_t)c.nc * c.model.ext_nodes[species],
                                  hipMemcpyHostToDevice, c.stream));
    FEDM_HIP_CHECK(hipStreamSynchronize(c.stream));
    return 0;
}

// the program is checked here once (stack discipline, index ranges): the kernel trusts it
int fedm_ext_source_program(fedm_ctx *h, int species, int n_ops, const int32_t *ops, int n_consts,
                            const double *consts, int n_params) {
    Ctx &c = h->c;
    if (species < 0 || species >= c.ns || !c.d_ext[species]) {
        set_error("species has no Expression source");
        return -2;
    }
    const int nodes = c.model.ext_nodes[species];
    if (nodes != 3 && nodes != 6 && nodes != 10) {
        set_error("device evaluation of Expression sources: degree 1, 2 or 3");
        return -2;
    }
    if (!ops || n_ops < 1 || n_ops > FEDM_EXPR_MAX_OPS || n_consts < 0 || (n_consts > 0 && !consts) ||
        n_params < 0 || n_params > FEDM_EXPR_MAX_PARAMS) {
        set_error("bad expression program");
        return -2;
    }
    int depth = 0;
    for (int k = 0; k < n_ops; ++k) {
        const int op = ops[2 * k], arg = ops[2 * k + 1];
        bool ok = true;
        if (op == FEDM_OP_CONST) ok = arg >= 0 && arg < n_consts, ++depth;
        else if (op == FEDM_OP_X) ok = arg == 0 || arg == 1, ++depth;
        else if (op == FEDM_OP_PARAM) ok = arg >= 0 && arg < n_params, ++depth;
        else if (op >= FEDM_OP_ADD && op <= FEDM_OP_POW) ok = depth >= 2, --depth;
        else if (op >= FEDM_OP_NEG && op <= FEDM_OP_ATAN) ok = depth >= 1;
        else ok = false;
        if (!ok || depth > FEDM_EXPR_STACK) {
            set_error("bad expression program (opcode, operand index or stack depth)");
            return -2;
        }
    }
    if (depth != 1) {
        set_error("bad expression program (it must leave one value)");
        return -2;
    }
    FEDM_HIP_CHECK(hipSetDevice(c.device));
    FEDM_HIP_CHECK(hipStreamSynchronize(c.stream));
    if (c.d_expr_ops[species]) hipFree(c.d_expr_ops[species]);
    if (c.d_expr_consts[species]) hipFree(c.d_expr_consts[species]);
    c.d_expr_ops[species] = nullptr;
    c.d_expr_consts[species] = nullptr;
    c.expr_n_ops[species] = 0;
    FEDM_HIP_CHECK(hipMalloc((void **)&c.d_expr_ops[species], sizeof(int) * 2 * n_ops));
    FEDM_HIP_CHECK(hipMalloc((void **)&c.d_expr_consts[species], sizeof(double) * (n_consts > 0 ? n_consts : 1)));
    FEDM_HIP_CHECK(hipMemcpy(c.d_expr_ops[species], ops, sizeof(int) * 2 * n_ops, hipMemcpyHostToDevice));
    if (n_consts > 0)
        FEDM_HIP_CHECK(hipMemcpy(c.d_expr_consts[species], consts, sizeof(double) * n_consts, hipMemcpyHostToDevice));
    c.expr_n_ops[species] = n_ops;
    c.expr_n_params[species] = n_params;
    return 0;
}

int fedm_ext_source_eval(fedm_ctx *h, int species, const double *params) {
    Ctx &c = h->c;
    if (species < 0 || species >= c.ns || !c.d_ext[species] || c.expr_n_ops[species] == 0) {
        set_error("species has no expression program");
        return -2;
    }
    if (c.expr_n_params[species] > 0 && !params) {
        set_error("null argument");
        return -2;
    }
    FEDM_HIP_CHECK(hipSetDevice(c.device));
    launch_ext_source_eval(c, species, params);
    FEDM_HIP_CHECK(hipGetLastError());
    return 0;
}

int fedm_residual(fedm_ctx *h, double *F_out, double *fnorm) {
    Ctx &c = h->c;
    FEDM_HIP_CHECK(hipSetDevice(c.device));
    double fn = 0.0;
    eval_residual(c, 0, &fn);
    if (fnorm) *fnorm = fn;
    if (F_out) return get_vec(c, F_out, c.d_F);
    FEDM_HIP_CHECK(hipGetLastError());
    return 0;
}

int fedm_jacobian(fedm_ctx *h) {
    Ctx &c = h->c;
    FEDM_HIP_CHECK(hipSetDevice(c.device));
    eval_jacobian(c, 0);
    FEDM_HIP_CHECK(hipStreamSynchronize(c.stream));
    FEDM_HIP_CHECK(hipGetLastError());
    return 0;
}

int64_t fedm_jacobian_nnz(fedm_ctx *h) {
    return h->c.pat.nnz_blocks * h->c.neq * h->c.neq;
}

int fedm_jacobian_csr(fedm_ctx *h, int64_t *indptr, int32_t *indices, double *values) {
    Ctx &c = h->c;
    FEDM_HIP_CHECK(hipSetDevice(c.device));
    const int neq = c.neq, neq2 = neq * neq;
    const size_t nval = (size_t)c.pat.total_bc * SLICE * neq2;
    std::vector<double> val(nval);
    FEDM_HIP_CHECK(hipStreamSynchronize(c.stream));
    FEDM_HIP_CHECK(hipMemcpy(val.data(), c.d_val, sizeof(double) * nval, hipMemcpyDeviceToHost));
    int64_t pos = 0;
    indptr[0] = 0;
    for (int v = 0; v < c.nv; ++v) {
        const int s = v / SLICE, l = v % SLICE;
        const int len = c.pat.row_len[v];
        for (int cr = 0; cr < neq; ++cr) {
            for (int j = 0; j < len; ++j) {
                const size_t bc = (size_t)c.pat.slice_boff[s] + j;
                const int col = c.pat.colidx[bc * SLICE + l];
                for (int cc = 0; cc < neq; ++cc) {
                    indices[pos] = col * neq + cc;
                    values[pos] = val[(bc * neq2 + cr * neq + cc) * SLICE + l];
                    ++pos;
                }
            }
            indptr[(size_t)v * neq + cr + 1] = pos;
        }
    }
    return 0;
}

int fedm_spmv(fedm_ctx *h, const double *x, double *y) {
    Ctx &c = h->c;
    FEDM_HIP_CHECK(hipSetDevice(c.device));
    FEDM_HIP_CHECK(hipMemsetAsync(c.d_tmp, 0, sizeof(double) * c.np, c.stream));
    if (put_vec(c, c.d_tmp, x)) return -1;
    launch_spmv(c, c.d_tmp, c.d_w, false);
    return get_vec(c, y, c.d_w);
}

int fedm_newton_solve(fedm_ctx *h, const fedm_newton_opts *o, fedm_newton_report *rep) {
    Ctx &c = h->c;
    FEDM_HIP_CHECK(hipSetDevice(c.device));
    fedm_newton_report r{};
    c.err_cache_comp = -1;
    const auto t_begin = std::chrono::steady_clock::now();
    int it = 0, lin_total = 0, rc = 0;
    double fnorm = 0.0, fnorm0 = 0.0, snorm = 0.0, xnorm = 0.0;
    while (true) {
        // one fused F + J assembly per iteration: the residual norm that decides convergence
        // comes from the same pass (a J assembly is wasted only on the final check)
        // The iteration at which the previous solve converged is expected to be the final check
        // again: assemble the residual only there (a wrong guess costs one extra F+J assembly).
        const bool residual_only = it > 0 && it == c.newton_its_hint;
        if (residual_only) {
            launch_assemble(c, false, 0);
            launch_finalize(c, false, 0);
        } else {
            eval_jacobian(c, 0);
        }
        // |F|, and |dx|, |x| of the previous update (slots 1, 2) in one publication.  After a Jacobian
        // assembly the field split's planes are formed while those numbers travel to the host (an
        // iteration that turns out to be the last one has formed them for nothing: the expected last
        // one assembles no Jacobian at all).
        // at the expected last iteration the watched component's change (adaptive_solver's error norm)
        // rides along: slots 3, 4
        // several GPUs: the update's sums and the error sums are rank-local until this publication's
        // all-reduce carries them with |F|^2 (one collective instead of three)
        const bool sums_local = c.comm && it > 0 && c.red12_local;
        const bool with_error = residual_only && (!c.comm || sums_local) && o->watch_component > 0 &&
                                o->watch_component <= c.neq;
        if (with_error) launch_field_error_slots34(c, o->watch_component - 1);
        norm2_publish(c, c.d_F, 0, with_error ? 5 : 3, sums_local ? (with_error ? 5 : 3) : 1);
        c.red12_local = false;
        bool planes_done = false;
        if (!residual_only && right_preconditioned(c)) {
            prepare_preconditioner_and_rhs(c);
            planes_done = true;
        }
        wait_red(c);
        fnorm = std::sqrt(c.h_red[0]);
        if (it > 0) {
            snorm = std::sqrt(c.h_red[1]);
            xnorm = std::sqrt(c.h_red[2]);
        }
        if (!std::isfinite(fnorm)) {
            rc = FEDM_DIVERGED_NAN;
            break;
        }
        const bool done = it == 0 ? fnorm < o->atol
                                  : (fnorm < o->atol || fnorm <= o->rtol * fnorm0 || snorm < o->stol * xnorm);
        if (it == 0) fnorm0 = fnorm;
        if (done) {
            if (with_error) {   // the state is final: keep the error norm for fedm_field_error
                c.err_cache = std::sqrt(c.h_red[3]) / std::sqrt(c.h_red[4]);
                c.err_cache_comp = o->watch_component - 1;
            }
            break;
        }
        if (it >= o->max_it) {
            rc = FEDM_DIVERGED_MAX_IT;
            break;
        }
        if (residual_only) eval_jacobian(c, 0);  // not converged after all: the Jacobian is needed
        if (!planes_done) prepare_preconditioner_and_rhs(c);
        int lits = 0;
        double lres = 0.0;
        const bool right = right_preconditioned(c);
        bool updated = false;
        const int lrc = gmres(c, o->ksp_restart, o->ksp_rtol, o->ksp_atol, o->ksp_max_it, &lits, &lres,
                              right ? c.d_F : c.d_rhs, right ? -1.0 : 1.0, right ? fnorm : -1.0, c.d_u, &updated);
        lin_total += lits;
        if (lrc != 0 || comm_failed(c)) {
            rc = (lrc < 0 || comm_failed(c)) ? -1 : (lrc == FEDM_DIVERGED_NAN ? FEDM_DIVERGED_NAN : FEDM_DIVERGED_LINEAR);
            break;
        }
        // |dx| and |x| for the stol test (slots 1, 2) are read with the next |F|
        if (!updated) {
            launch_axpy(c, 1.0, c.d_delta, c.d_u);
            launch_norm2(c, c.d_delta, 1);
            launch_norm2(c, c.d_u, 2);
            c.red12_local = false;   // (all-reduced by launch_norm2)
        }
        // ghost entries of the new state: exchanged by the next assembly, behind its interior patches
        if (c.comm && c.assembly_overlap) c.halo_pending = true;
        else comm_halo(c, c.d_u);
        if (comm_failed(c)) {
            rc = -1;
            break;
        }
        ++it;
    }
    if (comm_failed(c)) rc = -1;  // the message is in fedm_last_error (Comm::error)
    if (rc == 0) c.newton_its_hint = it;
    ++c.fs_solves[c.fs_alt_active ? 1 : 0];
    if ((c.fs_alt_sweeps > 0 || c.amg_alt) && it > 0) {
        // same counts on every rank, so every rank takes the same decision
        const double per_solve = (double)lin_total / it;
        if (!c.fs_measured_policy || (c.comm && c.comm->nranks > 1)) {
            const bool to_alt = !c.fs_alt_active && per_solve >= c.fs_switch_above;
            const bool to_main = c.fs_alt_active && per_solve <= c.fs_back_below;
            if (to_alt || to_main) set_hard_mode(c, to_alt);
        } else if (rc == 0) {
            // measured policy (one GPU): this solve's wall time per Newton iteration goes to the set it ran
            // with; in the hard regime the cheaper set is used, the other one is looked at again now and then
            const int cur = c.fs_alt_active ? 1 : 0;
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count() / it;
            ++c.fs_age[0];
            ++c.fs_age[1];
            if (c.fs_skip_sample) {
                c.fs_skip_sample = false;
            } else {
                c.fs_cost[cur] = c.fs_cost[cur] > 0.0 ? 0.6 * c.fs_cost[cur] + 0.4 * ms : ms;
                c.fs_age[cur] = 0;
            }
            const bool hard_regime = per_solve >= c.fs_switch_above || (c.fs_alt_active && per_solve > c.fs_back_below);
            if (c.fs_probe_left > 0) {
                if (--c.fs_probe_left == 0 && c.fs_cost[1 - cur] > 0.0 && c.fs_cost[1 - cur] <= c.fs_cost[cur])
                    set_hard_mode(c, cur == 0);   // the probed set lost: back to the other one
            } else if (!hard_regime) {
                if (c.fs_alt_active) set_hard_mode(c, false);
            } else if (c.fs_cost[1 - cur] == 0.0 || c.fs_age[1 - cur] >= c.fs_probe_every) {
                c.fs_probe_left = 2;              // (the first solve after the switch does not count)
                set_hard_mode(c, cur == 0);
            } else if (c.fs_cost[1 - cur] < 0.95 * c.fs_cost[cur]) {
                set_hard_mode(c, cur == 0);
            }
        }
    }
    r.iterations = it;
    r.linear_iterations = lin_total;
    r.fnorm0 = fnorm0;
    r.fnorm = fnorm;
    r.reason = rc > 0 ? rc : 0;
    r.converged = rc == 0 ? 1 : 0;
    if (rep) *rep = r;
    if (comm_failed(c)) {
        set_error(c.comm->error);
        return -1;
    }
    if (hipGetLastError() != hipSuccess) {
        set_error("HIP error during Newton solve");
        return -1;
    }
    return rc;
}

// Jacobi-preconditioned CG on the potential rows with the species frozen.  The matrix has
// identity rows for species / Dirichlet / padding dofs and their residual is zero once the
// state satisfies the boundary values, so CG runs on the symmetric positive definite
// remainder.  Replaces assemble(a), assemble(L), solve() of fedm-streamer.py:205-215.
int fedm_poisson_solve(fedm_ctx *h, double rtol, int max_it, int *iterations) {
    Ctx &c = h->c;
    c.err_cache_comp = -1;
    FEDM_HIP_CHECK(hipSetDevice(c.device));
    if (!c.poisson) {
        set_error("model has no Poisson row");
        return -2;
    }
    launch_set_dirichlet_state(c);
    launch_assemble(c, true, 1);
    launch_finalize(c, true, 1);
    launch_block_inverse(c);
    double *r = c.d_rhs, *z = c.d_tmp, *p = c.d_delta, *q = c.d_w;
    auto precondition = [&](const double *rr, double *zz) {
        if (c.amg) poisson_precondition(c, *c.amg, rr, zz);
        else launch_apply_dinv(c, rr, zz, 1.0);
    };
    // r = -F, x = 0 (correction), z = Minv r, p = z
    launch_scale_copy(c, -1.0, c.d_F, r);
    precondition(r, z);
    launch_scale_copy(c, 1.0, z, p);
    const double *rz_ptr[1] = {r};
    launch_dots(c, rz_ptr, z, 1);
    read_red(c, 1);
    double rz = c.h_red[0];
    launch_norm2(c, r, 0);
    read_red(c, 1);
    const double r0 = std::sqrt(c.h_red[0]);
    double rn = r0;
    int it = 0;
    if (ensure_krylov(c, 1)) return -1;
    double *x = c.d_V;  // accumulated correction
    hipMemsetAsync(x, 0, sizeof(double) * c.np, c.stream);
    while (rn > rtol * r0 && it < max_it && r0 > 0.0) {
        comm_halo(c, p);
        launch_spmv(c, p, q, false);
        const double *pp[1] = {p};
        launch_dots(c, pp, q, 1);
        read_red(c, 1);
        const double alpha = rz / c.h_red[0];
        launch_axpy(c, alpha, p, x);
        launch_axpy(c, -alpha, q, r);
        precondition(r, z);
        launch_dots(c, rz_ptr, z, 1);
        launch_norm2(c, r, 1);
        read_red(c, 2);
        const double rz_new = c.h_red[0];
        rn = std::sqrt(c.h_red[1]);
        if (!std::isfinite(rn)) break;
        const double beta = rz_new / rz;
        rz = rz_new;
        // p = z + beta p
        launch_scale_copy(c, beta, p, p);
        launch_axpy(c, 1.0, z, p);
        ++it;
    }
    launch_axpy(c, 1.0, x, c.d_u);
    comm_halo(c, c.d_u);
    FEDM_HIP_CHECK(hipStreamSynchronize(c.stream));
    if (iterations) *iterations = it;
    if (comm_failed(c)) {
        set_error(c.comm->error);
        return -1;
    }
    if (!std::isfinite(rn)) return FEDM_DIVERGED_NAN;
    return rn <= rtol * r0 || r0 == 0.0 ? 0 : FEDM_DIVERGED_LINEAR;
}

int fedm_field_error(fedm_ctx *h, int component, double *rel_err) {
    Ctx &c = h->c;
    if (component < 0 || component >= c.neq) {
        set_error("component out of range");
        return -2;
    }
    FEDM_HIP_CHECK(hipSetDevice(c.device));
    if (component == c.err_cache_comp) {   // computed with the last solve's final residual check
        *rel_err = c.err_cache;
        return 0;
    }
    launch_field_error(c, component);
    read_red(c, 2);
    *rel_err = std::sqrt(c.h_red[0]) / std::sqrt(c.h_red[1]);
    if (comm_failed(c)) {
        set_error(c.comm->error);
        return -1;
    }
    return 0;
}

int fedm_debug_species_planes_check(fedm_ctx *h, double *out) {
    Ctx &c = h->c;
    FEDM_HIP_CHECK(hipSetDevice(c.device));
    if (!out) {
        set_error("fedm_debug_species_planes_check: null argument");
        return -2;
    }
    return fieldsplit_planes_check(c, out, c.planes_last_fused);
}

int fedm_time_kernel(fedm_ctx *h, int kind, int repeats, double *ms_per_launch) {
    Ctx &c = h->c;
    FEDM_HIP_CHECK(hipSetDevice(c.device));
    hipEvent_t e0, e1;
    FEDM_HIP_CHECK(hipEventCreate(&e0));
    FEDM_HIP_CHECK(hipEventCreate(&e1));
    auto run = [&]() {
        if (kind == 0) {
            launch_assemble(c, true, 0);
        } else if (kind == 1) {
            launch_spmv(c, c.d_u, c.d_w, false);
        } else if (kind == 3) {
            if (c.amg) c.amg->run(c);  // one multigrid cycle on the potential block (its replayed graph)
        } else if (kind == 4) {
            if (c.amg && c.poisson) fieldsplit_setup(c);   // the preconditioner's species planes from the assembled Jacobian
        } else if (kind == 5) {
            launch_assemble(c, true, 0);                   // ... behind the assembly, as in a Newton iteration
            c.boundary_pending = 0;
            if (c.amg && c.poisson) fieldsplit_setup(c);
        } else {
            launch_assemble(c, false, 0);
        }
    };
    run();  // warm-up
    c.boundary_pending = 0;   // (the volume kernel alone is timed: no launch_finalize follows)
    FEDM_HIP_CHECK(hipStreamSynchronize(c.stream));
    FEDM_HIP_CHECK(hipEventRecord(e0, c.stream));
    for (int i = 0; i < repeats; ++i) run();
    FEDM_HIP_CHECK(hipEventRecord(e1, c.stream));
    FEDM_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    FEDM_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    *ms_per_launch = (double)ms / repeats;
    c.boundary_pending = 0;
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return 0;
}

int fedm_copy_bandwidth(int device, int64_t bytes, int repeats, double *gbs) {
    if (bytes < (1 << 20) || repeats < 1 || !gbs) {
        set_error("fedm_copy_bandwidth: at least 1 MiB and one repeat");
        return -2;
    }
    return copy_bandwidth(device, bytes, repeats, gbs);
}

// latency of the multi-GPU primitives on this context's transport, back to back on the compute
// stream: kind 0 = halo exchange of a block vector, 1 = of a scalar vector, 2 = all-reduce of 32
// doubles (what a Krylov step's dot products need)
int fedm_time_comm(fedm_ctx *h, int kind, int repeats, double *ms_per_op) {
    Ctx &c = h->c;
    if (!c.comm || kind < 0 || kind > 2 || repeats < 1 || !ms_per_op) {
        set_error("fedm_time_comm: no transport installed, or bad arguments");
        return -2;
    }
    FEDM_HIP_CHECK(hipSetDevice(c.device));
    hipEvent_t e0, e1;
    FEDM_HIP_CHECK(hipEventCreate(&e0));
    FEDM_HIP_CHECK(hipEventCreate(&e1));
    auto run = [&]() {
        if (kind == 0) comm_halo(c, c.d_w);
        else if (kind == 1) comm_halo_scalar(c, c.d_w);
        else comm_allreduce(c, c.d_red, 32);
    };
    FEDM_HIP_CHECK(hipMemsetAsync(c.d_w, 0, sizeof(double) * c.np, c.stream));
    run();
    FEDM_HIP_CHECK(hipStreamSynchronize(c.stream));
    FEDM_HIP_CHECK(hipEventRecord(e0, c.stream));
    for (int i = 0; i < repeats; ++i) run();
    FEDM_HIP_CHECK(hipEventRecord(e1, c.stream));
    FEDM_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    FEDM_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    *ms_per_op = (double)ms / repeats;
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    if (comm_failed(c)) {
        set_error(c.comm->error);
        return -1;
    }
    return 0;
}

int fedm_comm_unique_id(void *out128) { return comm_unique_id(out128); }

static int comm_common(Ctx &c, int n_nb, const int32_t *nb_rank, const int32_t *send_ptr,
                       const int32_t *send_idx, const int32_t *recv_ptr) {
    if (n_nb < 0 || (n_nb > 0 && (!nb_rank || !send_ptr || !recv_ptr))) {
        set_error("bad halo plan");
        return -2;
    }
    hipStreamSynchronize(c.stream);
    iter_graphs_clear(c);  // captured for the previous transport (or for none)
    if (c.comm) {
        c.comm->release();
        delete c.comm;
        c.comm = nullptr;
    }
    Comm *cm = new Comm();
    static const int32_t zero2[2] = {0, 0};
    const int rc = comm_setup_plan(c, *cm, n_nb, nb_rank, n_nb ? send_ptr : zero2, send_idx,
                                   n_nb ? recv_ptr : zero2);
    if (rc) {
        cm->release();
        delete cm;
        return rc;
    }
    c.comm = cm;
    return 0;
}

int fedm_comm_init_rccl(fedm_ctx *h, int n_nb, const int32_t *nb_rank, const int32_t *send_ptr,
                        const int32_t *send_idx, const int32_t *recv_ptr, const void *unique_id,
                        int rank, int n_ranks) {
    Ctx &c = h->c;
    FEDM_HIP_CHECK(hipSetDevice(c.device));
    if (int rc = comm_common(c, n_nb, nb_rank, send_ptr, send_idx, recv_ptr)) return rc;
    return comm_init_rccl(c, *c.comm, unique_id, rank, n_ranks);
}

int fedm_comm_init_callbacks(fedm_ctx *h, int n_nb, const int32_t *nb_rank, const int32_t *send_ptr,
                             const int32_t *send_idx, const int32_t *recv_ptr,
                             fedm_allreduce_fn allreduce, fedm_exchange_fn exchange, void *user,
                             int rank, int n_ranks) {
    Ctx &c = h->c;
    if (!allreduce || !exchange) {
        set_error("null transport callback");
        return -2;
    }
    FEDM_HIP_CHECK(hipSetDevice(c.device));
    if (int rc = comm_common(c, n_nb, nb_rank, send_ptr, send_idx, recv_ptr)) return rc;
    c.comm->kind = 1;
    c.comm->rank = rank;
    c.comm->nranks = n_ranks;
    c.comm->allreduce_cb = allreduce;
    c.comm->exchange_cb = exchange;
    c.comm->user = user;
    return 0;
}

int fedm_comm_stats(fedm_ctx *h, int64_t out[10]) {
    Ctx &c = h->c;
    for (int i = 0; i < 10; ++i) out[i] = 0;
    if (!c.comm) return 0;
    const Comm &cm = *c.comm;
    out[0] = cm.kind;
    out[1] = cm.nranks;
    out[2] = cm.n_exchanges;
    out[3] = cm.n_allreduces;
    out[4] = cm.failed ? 1 : 0;
    out[5] = cm.n_nb;
    out[6] = cm.n_patch_interior;
    out[7] = cm.n_patch_boundary;
    out[8] = cm.halo_bytes;
    out[9] = cm.allreduce_bytes;
    return 0;
}

int fedm_debug_comm_fault(int fail_at, int64_t out[6]) { return comm_fault_selftest(fail_at, out); }

int fedm_debug_comm_roundtrip(fedm_ctx *h, double *vec, double *red, int k) {
    Ctx &c = h->c;
    if (!c.comm || !vec || k < 0 || k > RED_K || (k > 0 && !red)) {
        set_error("no transport on this context, or bad arguments");
        return -2;
    }
    FEDM_HIP_CHECK(hipSetDevice(c.device));
    FEDM_HIP_CHECK(hipMemsetAsync(c.d_tmp, 0, sizeof(double) * c.np, c.stream));
    if (put_vec(c, c.d_tmp, vec)) return -1;
    comm_halo(c, c.d_tmp);          // on the compute stream ...
    comm_halo_begin(c);             // ... and once more the way the Krylov loop overlaps it: on the
    comm_halo_exchange(c, c.d_tmp); // communication stream, fenced by events (same values)
    if (k > 0) {
        FEDM_HIP_CHECK(hipMemcpyAsync(c.d_red, red, sizeof(double) * k, hipMemcpyHostToDevice, c.stream));
        comm_allreduce(c, c.d_red, k);
        comm_allreduce_f32_payload(c, c.d_red, k);   // the multigrid's single-precision payload (rounds to fp32)
        FEDM_HIP_CHECK(hipMemcpyAsync(red, c.d_red, sizeof(double) * k, hipMemcpyDeviceToHost, c.stream));
    }
    if (get_vec(c, vec, c.d_tmp)) return -1;
    if (comm_failed(c)) {
        set_error(c.comm->error);
        return -1;
    }
    return 0;
}

int fedm_debug_fieldsplit_tiles(fedm_ctx *h, int mode, int tile_slices, int depth, int threads) {
    if (!h) return -2;
    Ctx &c = h->c;
    FEDM_HIP_CHECK(hipSetDevice(c.device));
    FEDM_HIP_CHECK(hipStreamSynchronize(c.stream));
    iter_graphs_clear(c);   // captured Krylov steps hold the kernels of the old setting
    fs_tiles_configure(c, mode & 1, tile_slices, depth, threads);
    c.mg_tiles_off = (mode & 2) != 0;   // mode 3: species sweeps on tiles, the multigrid's finest-level sweeps not
    // the cycles' own graphs hold the kernels (and the tile tables) of the old setting: both hierarchies, the
    // one in use and the alternative for hard systems
    for (Amg *a : {c.amg, c.amg_alt})
        if (a && a->graph_exec) {
            hipGraphExecDestroy(a->graph_exec);
            a->graph_exec = nullptr;
        }
    return 0;
}

int fedm_debug_fieldsplit_apply(fedm_ctx *h, const double *t, double *z) {
    Ctx &c = h->c;
    if (!t || !z || !(c.amg && c.poisson)) {
        set_error("the field split needs a model with a Poisson row and a multigrid hierarchy (fedm_amg_setup)");
        return -2;
    }
    FEDM_HIP_CHECK(hipSetDevice(c.device));
    if (put_vec(c, c.d_rhs, t)) return -1;
    fieldsplit_setup(c);
    fieldsplit_apply(c, *c.amg, c.d_rhs, c.d_w, 1.0);
    FEDM_HIP_CHECK(hipGetLastError());
    return get_vec(c, z, c.d_w);
}

int fedm_pattern_stats(const fedm_mesh_desc *mesh, int64_t out[12]) {
    if (!mesh || !out || mesh->n_vertices < 3 || mesh->n_cells < 1) {
        set_error("null or empty mesh");
        return -2;
    }
    for (int i = 0; i < 3 * mesh->n_cells; ++i)
        if (mesh->cells[i] < 0 || mesh->cells[i] >= mesh->n_vertices) {
            set_error("cell vertex index out of range");
            return -2;
        }
    Pattern pat;
    build_pattern(*mesh, pat);
    // LDS-atomic clashes of the patch cell order: for each lane group of 16 cells and each local
    // vertex index a, owned a-vertices that fall into an accumulator bank class (id mod 16) another
    // cell of the group already uses
    int64_t pairs = 0, clashes = 0;
    for (int s = 0; s < pat.n_slices; ++s) {
        const int c0 = pat.patch_cell_ptr[s], c1 = pat.patch_cell_ptr[s + 1];
        for (int g0 = c0; g0 < c1; g0 += 16)
            for (int a = 0; a < 3; ++a) {
                uint32_t used = 0;
                for (int k = g0; k < std::min(g0 + 16, c1); ++k) {
                    const int lv = pat.patch_cells[k].lv[a];
                    if (lv >= SLICE) continue;
                    ++pairs;
                    if ((used >> (lv & 15)) & 1u) ++clashes;
                    used |= 1u << (lv & 15);
                }
            }
    }
    out[0] = pat.n_slices;
    out[1] = pat.max_patch_cells;
    out[2] = pat.max_patch_width;
    out[3] = pat.max_patch_verts;
    out[4] = (int64_t)pat.patch_cells.size();
    out[5] = pairs;
    out[6] = clashes;
    out[7] = pat.nnz_blocks;
    out[8] = pat.total_bc * SLICE;
    out[9] = (int64_t)pat.patch_halo.size();
    out[10] = (int64_t)pat.colour_ptr.size() - 1;
    // emission blocks: (wave of 64 cells, local row a) pairs in which some cell owns its a-th vertex -- the
    // assembly kernels skip the others (at most 3 per wave)
    int64_t blocks = 0;
    for (int s = 0; s < pat.n_slices; ++s) {
        const int c0 = pat.patch_cell_ptr[s], c1 = pat.patch_cell_ptr[s + 1];
        for (int w0 = c0; w0 < c1; w0 += SLICE)
            for (int a = 0; a < 3; ++a) {
                bool any = false;
                for (int k = w0; k < std::min(w0 + SLICE, c1); ++k) any = any || pat.patch_cells[k].lv[a] < SLICE;
                blocks += any;
            }
    }
    out[11] = blocks;
    return 0;
}

int fedm_sync_ghosts(fedm_ctx *h) {
    Ctx &c = h->c;
    FEDM_HIP_CHECK(hipSetDevice(c.device));
    comm_halo(c, c.d_u);
    comm_halo(c, c.d_uold);
    comm_halo(c, c.d_uold1);
    c.halo_pending = false;
    FEDM_HIP_CHECK(hipStreamSynchronize(c.stream));
    if (comm_failed(c)) {
        set_error(c.comm->error);
        return -1;
    }
    return 0;
}

int fedm_profile(fedm_ctx *h, int enable) {
    Ctx &c = h->c;
    FEDM_HIP_CHECK(hipSetDevice(c.device));
    Prof &p = c.prof;
    prof_collect(c);
    if (enable && p.ev.empty()) {
        p.ev.resize(16384);
        p.kind.resize(8192);
        for (auto &e : p.ev) FEDM_HIP_CHECK(hipEventCreate(&e));
    }
    p.on = enable != 0;
    p.all_kinds = enable == 2;  // SpMV / V-cycle events need plain launches (no iteration graphs)
    if (enable)
        for (int k = 0; k < 8; ++k) {
            p.ms[k] = 0.0;
            p.cnt[k] = 0;
            p.seen[k] = 0;
        }
    return 0;
}

int fedm_profile_read(fedm_ctx *h, int kind, double *ms_total, int64_t *count) {
    Ctx &c = h->c;
    if (kind < 0 || kind >= 8) return -2;
    prof_collect(c);
    // kinds that are sampled (Prof::stride) report the sampled mean times the launches seen
    const Prof &p = c.prof;
    const double mean = p.cnt[kind] ? p.ms[kind] / (double)p.cnt[kind] : 0.0;
    if (ms_total) *ms_total = mean * (double)p.seen[kind];
    if (count) *count = p.seen[kind];
    return 0;
}

int fedm_set_fieldsplit(fedm_ctx *h, int sweeps, const double *weights) {
    if (sweeps < 1 || sweeps > 16 || !weights) {
        set_error("field-split sweeps must be 1..16 with one weight each");
        return -2;
    }
    for (int i = 0; i < sweeps; ++i)
        if (!(weights[i] > 0.0 && weights[i] < 4.0)) {
            set_error("field-split weights must be positive");
            return -2;
        }
    hipSetDevice(h->c.device);
    hipStreamSynchronize(h->c.stream);
    iter_graphs_clear(h->c);
    Ctx &c = h->c;
    set_hard_mode(c, false);
    c.fs_sweeps = c.fs_main_sweeps = sweeps;
    for (int i = 0; i < sweeps; ++i) c.fs_w[i] = c.fs_main_w[i] = weights[i];
    c.fs_alt_sweeps = 0;
    return 0;
}

int fedm_set_fieldsplit_alternative(fedm_ctx *h, int alt_sweeps, const double *alt_weights,
                                    double switch_above, double back_below) {
    Ctx &c = h->c;
    if (alt_sweeps < 0 || alt_sweeps > 16 || (alt_sweeps > 0 && !alt_weights) || !(back_below < switch_above)) {
        set_error("alternative field-split sweeps must be 0..16 with one weight each, back_below < switch_above");
        return -2;
    }
    for (int i = 0; i < alt_sweeps; ++i)
        if (!(alt_weights[i] > 0.0 && alt_weights[i] < 4.0)) {
            set_error("field-split weights must be positive");
            return -2;
        }
    hipSetDevice(c.device);
    set_hard_mode(c, false);  // back to the main set first
    c.fs_alt_sweeps = alt_sweeps;
    for (int i = 0; i < alt_sweeps; ++i) c.fs_alt_w[i] = alt_weights[i];
    c.fs_switch_above = switch_above;
    c.fs_back_below = back_below;
    return 0;
}

int fedm_fieldsplit_policy(fedm_ctx *h, int64_t out[4]) {
    if (!h || !out) return -2;
    const Ctx &c = h->c;
    out[0] = (c.fs_measured_policy && !(c.comm && c.comm->nranks > 1)) ? 1 : 0;
    out[1] = c.fs_alt_active ? 1 : 0;
    out[2] = c.fs_solves[0];
    out[3] = c.fs_solves[1];
    return 0;
}

int fedm_set_assembly(fedm_ctx *h, int kind) {
    Ctx &c = h->c;
    if (kind == 1 && (!c.pat.patch_ok || patch_lds_bytes(c) > 160 * 1024 || c.model_kind == 1)) {
        set_error("LDS patch assembly unavailable for this mesh");
        return -2;
    }
    if (kind != 0 && kind != 1) {
        set_error("assembly kind must be 0 (colouring) or 1 (LDS patches)");
        return -2;
    }
    c.assembly_kind = kind;
    return 0;
}

int fedm_set_preconditioner_side(fedm_ctx *h, int right) {
    Ctx &c = h->c;
    if (right != 0 && right != 1) {
        set_error("preconditioner side must be 0 (left) or 1 (right)");
        return -2;
    }
    if (c.right_precond != (right == 1)) {
        hipStreamSynchronize(c.stream);
        iter_graphs_clear(c);  // captured for the other variant
        c.right_precond = right == 1;
    }
    return 0;
}

int fedm_set_fieldsplit_order(fedm_ctx *h, int upper) {
    Ctx &c = h->c;
    if (upper != 0 && upper != 1) {
        set_error("field-split order must be 0 (lower) or 1 (upper)");
        return -2;
    }
    if (c.fs_upper != (upper == 1)) {
        hipStreamSynchronize(c.stream);
        iter_graphs_clear(c);  // captured for the other order
        c.fs_upper = upper == 1;
    }
    return 0;
}

int fedm_plane_masks(fedm_ctx *h, uint32_t *kept_planes, uint32_t *zero_planes) {
    Ctx &c = h->c;
    if (kept_planes) *kept_planes = (c.skip_const_planes && c.assembly_kind == 1 && c.assembly_lean >= 2) ? c.const_plane_mask : 0u;
    if (zero_planes) *zero_planes = c.neq == 3 ? (c.zero_plane_mask & 10u) : 0u;
    return 0;
}

int fedm_sizes(fedm_ctx *h, int64_t *n_vertices, int64_t *n_cells, int64_t *n_eq,
               int64_t *nnz_blocks, int64_t *stored_blocks, int64_t *n_colours) {
    Ctx &c = h->c;
    if (n_vertices) *n_vertices = c.nv;
    if (n_cells) *n_cells = c.nc;
    if (n_eq) *n_eq = c.neq;
    if (nnz_blocks) *nnz_blocks = c.pat.nnz_blocks;
    if (stored_blocks) *stored_blocks = c.pat.total_bc * SLICE;
    if (n_colours) *n_colours = (int64_t)c.pat.colour_ptr.size() - 1;
    return 0;
}

int fedm_pattern_info(fedm_ctx *h, int64_t out[9]) {
    if (!h || !out) return -2;
    Ctx &c = h->c;
    out[0] = c.pat.n_slices;
    out[1] = c.pat.max_patch_cells;
    out[2] = c.pat.max_patch_width;
    out[3] = c.pat.max_patch_verts;
    out[4] = (int64_t)c.pat.patch_cells.size();
    out[5] = (int64_t)c.pat.patch_halo.size();
    // the volume assembly a fedm_jacobian call runs: 0 global colouring, 1 LDS patches with the
    // unrolled element routine, 2 LDS patches one equation row at a time (lean2 kernels)
    bool ext = false;
    for (int s = 0; s < c.ns; ++s) ext = ext || (c.model_kind == 0 && c.model.ext_nodes[s] > 0);
    const bool lean_model = c.assembly_kind == 1 && c.assembly_lean >= 2 && c.poisson && c.model_kind == 0 && !ext &&
                            c.model.n_qp == 3 && !c.model.linear_representation;
    const bool lean3 = lean_model && c.assembly_lean >= 3 && lean3_applies(c);
    const bool lean2 = lean_model && c.pat.max_patch_cells <= 256;
    // 3: LDS patches, one pass over the cells (lean3 kernels, assemble3.hip)
    out[6] = c.assembly_kind == 0 ? 0 : (lean3 ? 3 : lean2 ? 2 : 1);
    out[7] = c.assembly_kind == 0 ? 0 : (c.pat.max_patch_cells <= 192 || out[6] == 3 ? 192 : (lean2 ? 256 : 320));
    // the one-pass kernels with the model's structure compiled in (assemble3.hip, Lean3SigBenchmark)
    out[8] = lean3 ? lean3_signature(c) : 0;
    return 0;
}

int fedm_fieldsplit_tiles_stats(const fedm_mesh_desc *mesh, int tile_slices, int depth, int64_t out[10]) {
    if (!mesh || !out || mesh->n_vertices < 3 || mesh->n_cells < 1) {
        set_error("null or empty mesh");
        return -2;
    }
    for (int i = 0; i < 3 * mesh->n_cells; ++i)
        if (mesh->cells[i] < 0 || mesh->cells[i] >= mesh->n_vertices) {
            set_error("cell vertex index out of range");
            return -2;
        }
    long long v[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int rc = fs_tiles_host_stats(*mesh, tile_slices, depth, v);
    if (rc) {
        set_error("tile parameters out of range (1..8 slices, 1..8 layers) or a tile too large for 16-bit local indices");
        return rc;
    }
    for (int i = 0; i < 10; ++i) out[i] = v[i];
    return 0;
}

int fedm_fieldsplit_tiles_info(fedm_ctx *h, int64_t out[10]) {
    if (!h || !out) return -2;
    long long v[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int in_use = fs_tiles_info(h->c, v);
    for (int i = 0; i < 10; ++i) out[i] = v[i];
    return in_use;
}

int64_t fedm_block_nnz(fedm_ctx *h) { return h->c.pat.nnz_blocks; }

int fedm_block_csr(fedm_ctx *h, int cr, int cc, int64_t *indptr, int32_t *indices, double *values) {
    Ctx &c = h->c;
    if (cr < 0 || cr >= c.neq || cc < 0 || cc >= c.neq) {
        set_error("block component out of range");
        return -2;
    }
    FEDM_HIP_CHECK(hipSetDevice(c.device));
    const int neq2 = c.neq * c.neq;
    const size_t nplane = (size_t)c.pat.total_bc * SLICE;
    std::vector<double> plane(nplane);
    FEDM_HIP_CHECK(hipStreamSynchronize(c.stream));
    // one strided 2-D copy: plane e of every block column
    FEDM_HIP_CHECK(hipMemcpy2D(plane.data(), sizeof(double) * SLICE,
                               c.d_val + (size_t)(cr * c.neq + cc) * SLICE,
                               sizeof(double) * SLICE * neq2, sizeof(double) * SLICE,
                               (size_t)c.pat.total_bc, hipMemcpyDeviceToHost));
    int64_t pos = 0;
    indptr[0] = 0;
    for (int v = 0; v < c.nv; ++v) {
        const int s = v / SLICE, l = v % SLICE;
        for (int j = 0; j < c.pat.row_len[v]; ++j) {
            const size_t bc = (size_t)c.pat.slice_boff[s] + j;
            indices[pos] = c.pat.colidx[bc * SLICE + l];
            values[pos] = plane[bc * SLICE + l];
            ++pos;
        }
        indptr[v + 1] = pos;
    }
    return 0;
}

int fedm_jacobian_poisson_only(fedm_ctx *h) {
    Ctx &c = h->c;
    if (!c.poisson) {
        set_error("model has no Poisson row");
        return -2;
    }
    FEDM_HIP_CHECK(hipSetDevice(c.device));
    launch_set_dirichlet_state(c);
    eval_jacobian(c, 1);
    FEDM_HIP_CHECK(hipStreamSynchronize(c.stream));
    return 0;
}

int fedm_amg_clear(fedm_ctx *h) {
    Ctx &c = h->c;
    if (c.amg || c.amg_alt) {
        hipSetDevice(c.device);
        set_hard_mode(c, false);
        hipStreamSynchronize(c.stream);
        iter_graphs_clear(c);
        for (Amg **a : {&c.amg, &c.amg_alt})
            if (*a) {
                (*a)->release();
                delete *a;
                *a = nullptr;
            }
    }
    return 0;
}

// shared by the rank-local hierarchy and the replicated global one (several GPUs)
static int build_amg(Ctx &c, int n_first_rows, int n_levels, const fedm_csr *A, const fedm_csr *P,
                     const fedm_csr *R, const double *coarse_inverse, int nu, double omega, Amg **out,
                     int composite_from, const double *poly_w = nullptr /* [n_levels - 1][nu] */) {
    // the hierarchy's matrices in single precision (FEDM_MG_F32=0: double); see EllMat::single
    const char *mg_env = std::getenv("FEDM_MG_F32");
    const bool mg_single = !(mg_env && mg_env[0] == '0');
    if (n_levels < 1 || !A || (n_levels > 1 && (!P || !R)) || nu == 0) {
        set_error("bad multigrid description");
        return -2;
    }
    if (A[0].n_rows != n_first_rows || A[0].n_cols != n_first_rows) {
        set_error("finest multigrid operator has the wrong size");
        return -2;
    }
    for (int l = 0; l + 1 < n_levels; ++l)
        if (P[l].n_rows != A[l].n_rows || P[l].n_cols != A[l + 1].n_rows ||
            R[l].n_rows != A[l + 1].n_rows || R[l].n_cols != A[l].n_rows) {
            set_error("inconsistent multigrid level shapes");
            return -2;
        }
    Amg *amg = new Amg();
    amg->nu = nu < 0 ? -nu : nu;  // nu < 0 selects V(0,|nu|) cycles
    amg->pre_smooth = nu > 0;
    amg->omega = omega;
    amg->poly = poly_w != nullptr;
    amg->levels.resize(n_levels);
    auto fail = [&](int rc) {
        amg->release();
        delete amg;
        return rc;
    };
    for (int l = 0; l < n_levels; ++l) {
        Amg::Level &L = amg->levels[l];
        int rc = 0;
        const char *composite_env = std::getenv("FEDM_AMG_COMPOSITE");  // "0": four kernels per level everywhere
        const bool composite_ok = !(composite_env && composite_env[0] == '0');
        L.composite = composite_ok && l + 1 < n_levels && l >= composite_from && (nu == 1 || poly_w);
        if (poly_w && l + 1 < n_levels) {
            // polynomial smoother: S_pre (weights in order), S_post (backwards), built by the
            // recurrence S <- S + w Dinv (I - A S) from S = w_0 Dinv
            const int n = A[l].n_rows, np = ((n + SLICE - 1) / SLICE) * SLICE;
            std::vector<double> dinv((size_t)n, 1.0);
            for (int i = 0; i < n; ++i)
                for (int64_t k = A[l].indptr[i]; k < A[l].indptr[i + 1]; ++k)
                    if (A[l].indices[k] == i && A[l].values[k] != 0.0) dinv[i] = 1.0 / A[l].values[k];
            const HostCsr I = csr_identity(n);
            const fedm_csr Iv = I.view();
            L.w.assign(poly_w + (size_t)l * nu, poly_w + (size_t)(l + 1) * nu);
            auto smoother = [&](const std::vector<double> &ws) {
                HostCsr S = csr_identity(n);
                for (int i = 0; i < n; ++i) S.values[i] = ws[0] * dinv[i];
                std::vector<double> wd((size_t)n);
                for (size_t k = 1; k < ws.size(); ++k) {
                    const HostCsr AS = csr_product(A[l], nullptr, S.view(), 1.0);
                    const HostCsr T = csr_combine(Iv, 1.0, AS.view(), -1.0, 0, n, nullptr);  // I - A S
                    for (int i = 0; i < n; ++i) wd[i] = ws[k] * dinv[i];
                    const HostCsr WT = csr_combine(T.view(), 1.0, T.view(), 0.0, 0, n, wd.data());
                    S = csr_combine(S.view(), 1.0, WT.view(), 1.0, 0, n, nullptr);
                }
                return S;
            };
            const HostCsr Spre = smoother(L.w);
            const HostCsr ASp = csr_product(A[l], nullptr, Spre.view(), 1.0);
            const HostCsr Tpre = csr_combine(Iv, 1.0, ASp.view(), -1.0, 0, n, nullptr);         // I - A S_pre
            const HostCsr Cm = csr_product(R[l], nullptr, Tpre.view(), 1.0);                     // R (I - A S_pre)
            L.C.single = mg_single;
            rc |= L.C.from_csr(Cm.view(), false);
            if (L.composite) {
                const HostCsr Spost = smoother(std::vector<double>(L.w.rbegin(), L.w.rend()));
                const HostCsr SA = csr_product(Spost.view(), nullptr, A[l], 1.0);
                const HostCsr E = csr_combine(Iv, 1.0, SA.view(), -1.0, 0, n, nullptr);          // I - S_post A
                const HostCsr ES = csr_product(E.view(), nullptr, Spre.view(), 1.0);
                const HostCsr G = csr_combine(ES.view(), 1.0, Spost.view(), 1.0, 0, n, nullptr);  // E S_pre + S_post
                const HostCsr Q = csr_product(E.view(), nullptr, P[l], 1.0);                       // E P
                const HostCsr GQ = csr_combine(G.view(), 1.0, Q.view(), 1.0, np, np + P[l].n_cols, nullptr);
                L.GQ.single = mg_single;
            rc |= L.GQ.from_csr(GQ.view(), false);
                L.A.n_rows = n;
                L.A.n_rows_p = np;
            } else {
                // leg up as one product on the concatenated vector [b ; x_c]: x = S b + P x_c
                const HostCsr SP = csr_combine(Spre.view(), 1.0, P[l], 1.0, np, np + P[l].n_cols, nullptr);
                L.A.single = mg_single;
            rc |= L.A.from_csr(A[l], true);
                L.S.single = mg_single;
            rc |= L.S.from_csr(SP.view(), false);
                L.down_composite = true;
            }
        } else if (L.composite) {
            const int n = A[l].n_rows, np = ((n + SLICE - 1) / SLICE) * SLICE;
            std::vector<double> wd((size_t)n, omega);  // w / A_ii (EllMat::from_csr's rule for dinv)
            for (int i = 0; i < n; ++i)
                for (int64_t k = A[l].indptr[i]; k < A[l].indptr[i + 1]; ++k)
                    if (A[l].indices[k] == i && A[l].values[k] != 0.0) wd[i] = omega / A[l].values[k];
            const HostCsr I = csr_identity(n);
            const fedm_csr Iv = I.view();
            const HostCsr M1 = csr_product(A[l], wd.data(), Iv, 1.0);                         // w A Dinv
            const HostCsr T = csr_combine(Iv, 1.0, M1.view(), -1.0, 0, n, nullptr);            // I - w A Dinv
            const HostCsr Cm = csr_product(R[l], nullptr, T.view(), 1.0);                      // R (I - w A Dinv)
            const HostCsr G = csr_combine(Iv, 2.0, M1.view(), -1.0, 0, n, wd.data());          // w Dinv (2I - w A Dinv)
            HostCsr WAP = csr_product(A[l], nullptr, P[l], 1.0);                               // w Dinv A P
            for (int i = 0; i < n; ++i)
                for (int64_t k = WAP.indptr[i]; k < WAP.indptr[i + 1]; ++k) WAP.values[k] *= wd[i];
            const HostCsr Q = csr_combine(P[l], 1.0, WAP.view(), -1.0, 0, P[l].n_cols, nullptr);  // (I - w Dinv A) P
            const HostCsr GQ = csr_combine(G.view(), 1.0, Q.view(), 1.0, np, np + P[l].n_cols, nullptr);
            L.C.single = mg_single;
            rc |= L.C.from_csr(Cm.view(), false);
            L.GQ.single = mg_single;
            rc |= L.GQ.from_csr(GQ.view(), false);
            L.A.n_rows = n;
            L.A.n_rows_p = np;
        } else if (l + 1 < n_levels) {
            L.A.single = mg_single;
            rc |= L.A.from_csr(A[l], true);
            L.P.single = mg_single;
            rc |= L.P.from_csr(P[l], false);
            // single-GPU hierarchy (its coarsest level is solved here): the finest level's leg down
            // as one product; across GPUs that level is a distributed operator with halo exchanges
            L.down_composite = composite_ok && l == 0 && composite_from == 1 && coarse_inverse && nu == 1;
            if (L.down_composite) {
                const int n = A[l].n_rows;
                std::vector<double> wd((size_t)n, omega);
                for (int i = 0; i < n; ++i)
                    for (int64_t k = A[l].indptr[i]; k < A[l].indptr[i + 1]; ++k)
                        if (A[l].indices[k] == i && A[l].values[k] != 0.0) wd[i] = omega / A[l].values[k];
                const HostCsr I = csr_identity(n);
                const HostCsr M1 = csr_product(A[l], wd.data(), I.view(), 1.0);
                const HostCsr T = csr_combine(I.view(), 1.0, M1.view(), -1.0, 0, n, nullptr);
                const HostCsr Cm = csr_product(R[l], nullptr, T.view(), 1.0);
                L.C.single = mg_single;
            rc |= L.C.from_csr(Cm.view(), false);
            } else {
                L.R.single = mg_single;
            rc |= L.R.from_csr(R[l], false);
            }
        } else {
            L.A.n_rows = A[l].n_rows;
            L.A.n_rows_p = ((A[l].n_rows + SLICE - 1) / SLICE) * SLICE;
        }
        if (rc) {
            set_error("multigrid level upload failed (bad CSR or out of memory)");
            return fail(rc < -1 ? -2 : -1);
        }
        const size_t n = (size_t)L.A.n_rows_p;
        const bool tail = L.composite || (poly_w && l + 1 < n_levels);  // b = [b ; next level's x]
        const size_t n_next = tail ? (size_t)(((A[l + 1].n_rows + SLICE - 1) / SLICE) * SLICE) : 0;
        L.x_is_alias = l > 0 && (amg->levels[l - 1].composite || poly_w);
        if (L.x_is_alias) L.x = amg->levels[l - 1].b + amg->levels[l - 1].A.n_rows_p;
        for (double **p : {&L.x, &L.x2, &L.b, &L.r}) {
            if (p == &L.x && L.x_is_alias) continue;
            const size_t len = n + (p == &L.b ? n_next : 0);
            if (hipMalloc((void **)p, sizeof(double) * len) != hipSuccess ||
                hipMemset(*p, 0, sizeof(double) * len) != hipSuccess) {
                set_error("out of device memory for the multigrid vectors");
                return fail(-1);
            }
        }
    }
    if (!coarse_inverse && composite_from == 1 && n_levels > 1 && c.n_owned < c.nv) {
        // several GPUs: which slices of the finest operator touch ghost columns
        const EllMat &A0 = amg->levels[0].A;
        const int l2s = A0.log2_split;
        std::vector<int> in_list, bd_list;
        for (int sl = 0; sl < A0.n_slices; ++sl) {
            bool ghost = false;
            for (int lane = 0; lane < SLICE && !ghost; lane += (1 << l2s)) {
                const int64_t r = ((int64_t)sl * SLICE + lane) >> l2s;
                if (r >= A[0].n_rows) break;
                for (int64_t k = A[0].indptr[r]; k < A[0].indptr[r + 1]; ++k)
                    if (A[0].indices[k] >= c.n_owned) {
                        ghost = true;
                        break;
                    }
            }
            (ghost ? bd_list : in_list).push_back(sl);
        }
        amg->n_interior0 = (int)in_list.size();
        amg->n_boundary0 = (int)bd_list.size();
        if (hipMalloc((void **)&amg->d_interior0, sizeof(int) * std::max<size_t>(in_list.size(), 1)) != hipSuccess ||
            hipMalloc((void **)&amg->d_boundary0, sizeof(int) * std::max<size_t>(bd_list.size(), 1)) != hipSuccess ||
            hipMemcpy(amg->d_interior0, in_list.data(), sizeof(int) * in_list.size(), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(amg->d_boundary0, bd_list.data(), sizeof(int) * bd_list.size(), hipMemcpyHostToDevice) != hipSuccess) {
            set_error("out of device memory for the multigrid slice lists");
            return fail(-1);
        }
    }
    amg->n_coarse = A[n_levels - 1].n_rows;
    amg->coarse_ld = ((amg->n_coarse + 255) / 256) * 256;
    if (coarse_inverse) {  // nullptr: the coarsest problem will be handed to a global hierarchy
        if (amg->n_coarse > 8192) {
            set_error("dense coarsest multigrid level above 8192 unknowns");
            return fail(-2);
        }
        std::vector<double> inv((size_t)amg->n_coarse * amg->coarse_ld, 0.0);
        for (int i = 0; i < amg->n_coarse; ++i)
            for (int j = 0; j < amg->n_coarse; ++j)
                inv[(size_t)i * amg->coarse_ld + j] = coarse_inverse[(size_t)i * amg->n_coarse + j];
        if (hipMalloc((void **)&amg->coarse_inv, sizeof(double) * inv.size()) != hipSuccess ||
            hipMemcpy(amg->coarse_inv, inv.data(), sizeof(double) * inv.size(), hipMemcpyHostToDevice) != hipSuccess) {
            set_error("out of device memory for the coarse inverse");
            return fail(-1);
        }
    }
    *out = amg;
    return 0;
}

int fedm_amg_setup(fedm_ctx *h, int n_levels, const fedm_csr *A, const fedm_csr *P,
                   const fedm_csr *R, const double *coarse_inverse, int nu, double omega) {
    Ctx &c = h->c;
    FEDM_HIP_CHECK(hipSetDevice(c.device));
    fedm_amg_clear(h);
    Amg *amg = nullptr;
    if (int rc = build_amg(c, c.nv, n_levels, A, P, R, coarse_inverse, nu, omega, &amg, 1)) return rc;
    c.amg = amg;
    FEDM_HIP_CHECK(hipStreamSynchronize(c.stream));
    if (coarse_inverse && amg->capture(c) != 0) {
        hipGetLastError();  // graph capture unavailable: fall back to plain launches
    }
    return 0;
}

int fedm_amg_setup_poly(fedm_ctx *h, int n_levels, const fedm_csr *A, const fedm_csr *P, const fedm_csr *R,
                        const double *coarse_inverse, int degree, const double *weights, int as_alternative) {
    Ctx &c = h->c;
    FEDM_HIP_CHECK(hipSetDevice(c.device));
    if (degree < 1 || degree > 4 || !weights || !coarse_inverse || n_levels < 2) {
        set_error("polynomial-smoother hierarchy: degree 1..4, one weight per sweep and level, dense coarsest level");
        return -2;
    }
    for (int i = 0; i < (n_levels - 1) * degree; ++i)
        if (!(weights[i] > 0.0 && weights[i] < 8.0)) {
            set_error("polynomial-smoother weights must be positive");
            return -2;
        }
    if (c.comm) {
        set_error("polynomial-smoother hierarchy is for one GPU (the distributed finest level keeps V(1,1))");
        return -2;
    }
    if (as_alternative && !c.amg) {
        set_error("install the main hierarchy (fedm_amg_setup) before its alternative");
        return -2;
    }
    if (as_alternative) {
        set_hard_mode(c, false);
        if (c.amg_alt) {
            c.amg_alt->release();
            delete c.amg_alt;
            c.amg_alt = nullptr;
        }
    } else {
        fedm_amg_clear(h);
    }
    Amg *amg = nullptr;
    if (int rc = build_amg(c, c.nv, n_levels, A, P, R, coarse_inverse, degree, 1.0, &amg, 1, weights)) return rc;
    FEDM_HIP_CHECK(hipStreamSynchronize(c.stream));
    (as_alternative ? c.amg_alt : c.amg) = amg;
    if (amg->capture(c) != 0) hipGetLastError();  // graph capture unavailable: plain launches
    return 0;
}

int fedm_amg_set_global_hierarchy(fedm_ctx *h, int n_global, int offset, int n_levels, const fedm_csr *A,
                                  const fedm_csr *P, const fedm_csr *R, const double *coarse_inverse,
                                  int nu, double omega) {
    Ctx &c = h->c;
    if (!c.amg || !coarse_inverse || offset < 0 || offset + c.amg->n_coarse > n_global) {
        set_error("bad global hierarchy description (install the local hierarchy first)");
        return -2;
    }
    FEDM_HIP_CHECK(hipSetDevice(c.device));
    FEDM_HIP_CHECK(hipStreamSynchronize(c.stream));
    iter_graphs_clear(c);
    Amg &a = *c.amg;
    if (a.global) {
        a.global->release();
        delete a.global;
        a.global = nullptr;
    }
    if (a.graph_exec) {  // a captured cycle would solve the coarsest problem locally
        hipGraphExecDestroy(a.graph_exec);
        a.graph_exec = nullptr;
    }
    Amg *g = nullptr;
    if (int rc = build_amg(c, n_global, n_levels, A, P, R, coarse_inverse, nu, omega, &g, 0)) return rc;
    FEDM_HIP_CHECK(hipStreamSynchronize(c.stream));
    if (g->capture(c) != 0) hipGetLastError();
    a.global = g;
    a.n_global = n_global;
    a.g_offset = offset;
    a.d_gb = g->levels[0].b;  // the global right-hand side IS the replicated hierarchy's input
    if (c.comm && comm_reserve_reduction(c, n_global)) return -1;
    return 0;
}

}  // extern "C"
