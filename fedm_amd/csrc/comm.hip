#include "comm.hpp"

#include <dlfcn.h>

#include <cstring>
#include <string>
#include <vector>

namespace fedm {

// ---- RCCL, resolved at run time so that the library also loads where RCCL is absent ----------
struct Id128 {  // ncclUniqueId, passed by value
    char internal[128];
};
namespace {
struct NcclApi {
    void *lib = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, Id128, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*CommAbort)(void *) = nullptr;  // optional; what a failed communicator is torn down with
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*CommGetAsyncError)(void *, int *) = nullptr;  // optional
};
NcclApi g_nccl;
constexpr int kNcclDouble = 8;  // ncclFloat64
constexpr int kNcclFloat = 7;   // ncclFloat32
constexpr int kNcclSum = 0;

int load_nccl() {
    if (g_nccl.lib) return 0;
    // RCCL must sit on the HIP/HSA runtime this library itself runs on.  A process may hold two ROCm
    // installations (PyTorch's wheel carries its own libamdhip64 / libhsa-runtime64 / librccl next to
    // /opt/rocm's): a bare dlopen("librccl.so.1") returns whichever RCCL was loaded first, and an RCCL
    // that opens the *other* installation's HSA finds it uninitialised ("no ROCm-capable device is
    // detected").  So: first the RCCL in the directory of the HIP runtime in use, then the search path.
    std::vector<std::string> names;
    Dl_info info;
    if (dladdr(reinterpret_cast<const void *>(&hipGetDeviceCount), &info) && info.dli_fname) {
        const std::string hip_path = info.dli_fname;
        const size_t slash = hip_path.rfind('/');
        if (slash != std::string::npos) {
            names.push_back(hip_path.substr(0, slash) + "/librccl.so.1");
            names.push_back(hip_path.substr(0, slash) + "/librccl.so");
        }
    }
    names.insert(names.end(), {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"});
    for (const std::string &n : names) {
        g_nccl.lib = dlopen(n.c_str(), RTLD_NOW | RTLD_GLOBAL);
        if (g_nccl.lib) break;
    }
    if (!g_nccl.lib) {
        set_error("cannot load librccl.so");
        return -1;
    }
#define FEDM_SYM(field, name)                                        \
    *(void **)(&g_nccl.field) = dlsym(g_nccl.lib, name);             \
    if (!g_nccl.field) {                                             \
        set_error(std::string("RCCL symbol missing: ") + name);      \
        return -1;                                                   \
    }
    FEDM_SYM(GetUniqueId, "ncclGetUniqueId")
    FEDM_SYM(CommInitRank, "ncclCommInitRank")
    FEDM_SYM(CommDestroy, "ncclCommDestroy")
    FEDM_SYM(AllReduce, "ncclAllReduce")
    FEDM_SYM(Send, "ncclSend")
    FEDM_SYM(Recv, "ncclRecv")
    FEDM_SYM(GroupStart, "ncclGroupStart")
    FEDM_SYM(GroupEnd, "ncclGroupEnd")
    FEDM_SYM(GetErrorString, "ncclGetErrorString")
#undef FEDM_SYM
    *(void **)(&g_nccl.CommGetAsyncError) = dlsym(g_nccl.lib, "ncclCommGetAsyncError");
    *(void **)(&g_nccl.CommAbort) = dlsym(g_nccl.lib, "ncclCommAbort");
    return 0;
}

// every RCCL return code goes through here: the first failure is latched in the Comm
bool nccl_ok(Comm *cm, int rc, const char *what) {
    if (rc == 0) return true;
    if (!cm->failed) {
        cm->failed = true;
        const char *msg = g_nccl.GetErrorString ? g_nccl.GetErrorString(rc) : "unknown error";
        cm->error = std::string("RCCL transport failed in ") + what + ": " + (msg ? msg : "unknown error") +
                    " (rank " + std::to_string(cm->rank) + " of " + std::to_string(cm->nranks) + ")";
        set_error(cm->error);
    }
    return false;
}
}  // namespace

bool comm_failed(const Ctx &c) { return c.comm && c.comm->failed; }

// ---- fault injection for the error path (tests/test_host_logic.py; runs without a GPU) --------
// A stub API table whose `fail_at`-th call returns ncclSystemError drives the same
// exchange_packed / comm_allreduce code the RCCL transport uses (host pointers, no kernels).
namespace {
int g_stub_calls = 0, g_stub_fail_at = -1, g_stub_after_failure = 0;
bool g_stub_failed = false;
int stub_rc() {
    if (g_stub_failed) ++g_stub_after_failure;
    if (g_stub_calls++ == g_stub_fail_at) {
        g_stub_failed = true;
        return 2;  // ncclSystemError
    }
    return 0;
}
int stub_allreduce(const void *, void *, size_t, int, int, void *, hipStream_t) { return stub_rc(); }
int stub_send(const void *, size_t, int, int, void *, hipStream_t) { return stub_rc(); }
int stub_recv(void *, size_t, int, int, void *, hipStream_t) { return stub_rc(); }
int stub_group() { return stub_rc(); }
const char *stub_errstr(int) { return "stub system error"; }
int g_stub_aborts = 0, g_stub_destroys = 0;
int stub_abort(void *) { return ++g_stub_aborts, 0; }
int stub_destroy(void *) { return ++g_stub_destroys, 0; }
}  // namespace
static void exchange_packed(Ctx &c, Comm *cm, double *recv_dst, hipStream_t st, int w);

int comm_fault_selftest(int fail_at, int64_t out[6]) {
    const NcclApi saved = g_nccl;
    g_nccl = NcclApi{};
    g_nccl.CommAbort = stub_abort;
    g_nccl.CommDestroy = stub_destroy;
    g_stub_aborts = g_stub_destroys = 0;
    g_nccl.AllReduce = stub_allreduce;
    g_nccl.Send = stub_send;
    g_nccl.Recv = stub_recv;
    g_nccl.GroupStart = stub_group;
    g_nccl.GroupEnd = stub_group;
    g_nccl.GetErrorString = stub_errstr;
    g_stub_calls = 0;
    g_stub_fail_at = fail_at;
    g_stub_after_failure = 0;
    g_stub_failed = false;
    Ctx c;
    Comm cm;
    cm.kind = 2;
    cm.rank = 1;
    cm.nranks = 2;
    cm.n_nb = 1;
    cm.nb_rank = {0};
    cm.send_ptr = {0, 1};
    cm.recv_ptr = {0, 1};
    cm.n_send = cm.n_ghost = 1;
    double sendbuf[4] = {0}, recvbuf[4] = {0}, red[4] = {0};
    cm.d_sendbuf = sendbuf;
    cm.nccl = &cm;  // never dereferenced by the stubs
    c.comm = &cm;
    set_error("");
    // two rounds of what a Krylov step issues: halo group (4 calls), all-reduce (1 call)
    for (int round = 0; round < 2; ++round) {
        exchange_packed(c, &cm, recvbuf, nullptr, 1);
        comm_allreduce(c, red, 3);
    }
    out[0] = cm.failed ? 1 : 0;
    out[1] = g_stub_calls;           // API calls that reached the transport
    out[2] = g_stub_after_failure;   // ... of which after the failing one (only the group's close may follow)
    out[3] = comm_failed(c) ? 1 : 0;
    // teardown: a failed communicator is aborted (its unfinished collective would make ncclCommDestroy,
    // and every stream synchronisation behind it, wait for ever), a healthy one destroyed
    const bool failed = cm.failed;
    cm.release_communicator();
    out[4] = g_stub_aborts;
    out[5] = g_stub_destroys;
    cm.failed = failed;
    c.comm = nullptr;
    cm.d_sendbuf = nullptr;
    cm.nccl = nullptr;
    cm.kind = 0;
    g_nccl = saved;
    return cm.failed ? 1 : 0;
}

bool comm_poll_async_error(Ctx &c) {
    Comm *cm = c.comm;
    if (!cm || cm->kind != 2 || !cm->nccl || !g_nccl.CommGetAsyncError) return cm && cm->failed;
    int async = 0;
    if (!nccl_ok(cm, g_nccl.CommGetAsyncError(cm->nccl, &async), "ncclCommGetAsyncError")) return true;
    // ncclInProgress (7) is not an error
    if (async != 0 && async != 7) nccl_ok(cm, async, "an asynchronous operation (ncclCommGetAsyncError)");
    return cm->failed;
}

int comm_unique_id(void *out128) {
    if (load_nccl()) return -1;
    Id128 id;
    const int rc = g_nccl.GetUniqueId(&id);
    if (rc != 0) {
        set_error(std::string("ncclGetUniqueId: ") + g_nccl.GetErrorString(rc));
        return -1;
    }
    std::memcpy(out128, &id, 128);
    return 0;
}

// Ends the RCCL communicator.  After a latched transport failure (a lost peer, a failing call) the
// kernel of the unfinished collective may still spin on one of the streams: ncclCommDestroy would wait
// for it, and so would every later stream synchronisation and hipFree of the context -- the run would
// hang at teardown instead of ending with the promised error.  ncclCommAbort stops those kernels; it
// is therefore what a failed communicator gets, BEFORE anything synchronises with the device.
bool Comm::release_communicator() {
    if (!nccl) return false;
    bool aborted = false;
    if (failed && g_nccl.CommAbort) {
        g_nccl.CommAbort(nccl);
        aborted = true;
    } else if (g_nccl.CommDestroy) {
        g_nccl.CommDestroy(nccl);
    }
    nccl = nullptr;
    return aborted;
}

void Comm::release() {
    release_communicator();
    if (d_interior) hipFree(d_interior);
    if (d_boundary) hipFree(d_boundary);
    if (d_patch_interior) hipFree(d_patch_interior);
    if (d_patch_boundary) hipFree(d_patch_boundary);
    d_patch_interior = d_patch_boundary = nullptr;
    if (ev_ready) hipEventDestroy(ev_ready);
    if (ev_halo) hipEventDestroy(ev_halo);
    if (stream) hipStreamDestroy(stream);
    d_interior = d_boundary = nullptr;
    ev_ready = ev_halo = nullptr;
    stream = nullptr;
    if (d_send_idx) hipFree(d_send_idx);
    if (d_sendbuf) hipFree(d_sendbuf);
    if (d_recvtmp) hipFree(d_recvtmp);
    if (h_send) hipHostFree(h_send);
    if (h_recv) hipHostFree(h_recv);
    if (h_red) hipHostFree(h_red);
    if (d_red32) hipFree(d_red32);
    d_red32 = nullptr;
    red32_cap = 0;
    d_send_idx = nullptr;
    d_sendbuf = d_recvtmp = nullptr;
    h_send = h_recv = h_red = nullptr;
    kind = 0;
}

int comm_setup_plan(Ctx &c, Comm &cm, int n_nb, const int32_t *nb_rank, const int32_t *send_ptr,
                    const int32_t *send_idx, const int32_t *recv_ptr) {
    cm.n_nb = n_nb;
    cm.nb_rank.assign(nb_rank, nb_rank + n_nb);
    cm.send_ptr.assign(send_ptr, send_ptr + n_nb + 1);
    cm.recv_ptr.assign(recv_ptr, recv_ptr + n_nb + 1);
    cm.n_send = cm.send_ptr[n_nb];
    cm.n_ghost = cm.recv_ptr[n_nb];
    if (cm.n_ghost != c.nv - c.n_owned) {
        set_error("halo plan does not cover the ghost vertices of the local mesh");
        return -2;
    }
    for (int i = 0; i < cm.n_send; ++i)
        if (send_idx[i] < 0 || send_idx[i] >= c.n_owned) {
            set_error("halo send index is not an owned vertex");
            return -2;
        }
    const size_t ns = std::max(cm.n_send, 1), ng = std::max(cm.n_ghost, 1);
    FEDM_HIP_CHECK(hipMalloc((void **)&cm.d_send_idx, sizeof(int) * ns));
    if (cm.n_send)
        FEDM_HIP_CHECK(hipMemcpy(cm.d_send_idx, send_idx, sizeof(int) * cm.n_send, hipMemcpyHostToDevice));
    FEDM_HIP_CHECK(hipMalloc((void **)&cm.d_sendbuf, sizeof(double) * ns * c.neq));
    FEDM_HIP_CHECK(hipMalloc((void **)&cm.d_recvtmp, sizeof(double) * ng * c.neq));
    FEDM_HIP_CHECK(hipHostMalloc((void **)&cm.h_send, sizeof(double) * ns * c.neq));
    FEDM_HIP_CHECK(hipHostMalloc((void **)&cm.h_recv, sizeof(double) * ng * c.neq));
    FEDM_HIP_CHECK(hipHostMalloc((void **)&cm.h_red, sizeof(double) * 16384));
    cm.h_red_cap = 16384;
    // interior / boundary matrix slices: a slice is interior when none of its stored columns is
    // a ghost vertex (padding entries point at the row itself)
    std::vector<int> interior, boundary;
    for (int sl = 0; sl < c.pat.n_slices; ++sl) {
        bool ghost = false;
        for (int bc = c.pat.slice_boff[sl]; bc < c.pat.slice_boff[sl + 1] && !ghost; ++bc)
            for (int l = 0; l < SLICE; ++l)
                if (c.pat.colidx[(size_t)bc * SLICE + l] >= c.n_owned) {
                    ghost = true;
                    break;
                }
        (ghost ? boundary : interior).push_back(sl);
    }
    cm.n_interior = (int)interior.size();
    cm.n_boundary = (int)boundary.size();
    FEDM_HIP_CHECK(hipMalloc((void **)&cm.d_interior, sizeof(int) * std::max<size_t>(interior.size(), 1)));
    FEDM_HIP_CHECK(hipMalloc((void **)&cm.d_boundary, sizeof(int) * std::max<size_t>(boundary.size(), 1)));
    if (cm.n_interior)
        FEDM_HIP_CHECK(hipMemcpy(cm.d_interior, interior.data(), sizeof(int) * interior.size(), hipMemcpyHostToDevice));
    if (cm.n_boundary)
        FEDM_HIP_CHECK(hipMemcpy(cm.d_boundary, boundary.data(), sizeof(int) * boundary.size(), hipMemcpyHostToDevice));
    // assembly patches (= slices) that stage no ghost vertex: their owned lanes and halo vertices
    // are all owned.  The state halo travels while they are assembled (kernels.hip, lean2 launch).
    {
        std::vector<int> pin, pbd;
        for (int sl = 0; sl < c.pat.n_slices; ++sl) {
            bool ghost = (sl + 1) * SLICE > c.n_owned;  // ghost (or padding) lanes
            for (int k = c.pat.patch_halo_ptr[sl]; k < c.pat.patch_halo_ptr[sl + 1] && !ghost; ++k)
                ghost = c.pat.patch_halo[k] >= c.n_owned;
            (ghost ? pbd : pin).push_back(sl);
        }
        cm.n_patch_interior = (int)pin.size();
        cm.n_patch_boundary = (int)pbd.size();
        FEDM_HIP_CHECK(hipMalloc((void **)&cm.d_patch_interior, sizeof(int) * std::max<size_t>(pin.size(), 1)));
        FEDM_HIP_CHECK(hipMalloc((void **)&cm.d_patch_boundary, sizeof(int) * std::max<size_t>(pbd.size(), 1)));
        if (!pin.empty())
            FEDM_HIP_CHECK(hipMemcpy(cm.d_patch_interior, pin.data(), sizeof(int) * pin.size(), hipMemcpyHostToDevice));
        if (!pbd.empty())
            FEDM_HIP_CHECK(hipMemcpy(cm.d_patch_boundary, pbd.data(), sizeof(int) * pbd.size(), hipMemcpyHostToDevice));
    }
    FEDM_HIP_CHECK(hipStreamCreateWithFlags(&cm.stream, hipStreamNonBlocking));
    FEDM_HIP_CHECK(hipEventCreateWithFlags(&cm.ev_ready, hipEventDisableTiming));
    FEDM_HIP_CHECK(hipEventCreateWithFlags(&cm.ev_halo, hipEventDisableTiming));
    return 0;
}

int comm_init_rccl(Ctx &c, Comm &cm, const void *unique_id, int rank, int nranks) {
    if (load_nccl()) return -1;
    Id128 id;
    std::memcpy(&id, unique_id, 128);
    const int rc = g_nccl.CommInitRank(&cm.nccl, nranks, id, rank);
    if (rc != 0) {
        set_error(std::string("ncclCommInitRank: ") + g_nccl.GetErrorString(rc));
        return -1;
    }
    cm.kind = 2;
    cm.rank = rank;
    cm.nranks = nranks;
    return 0;
}

int comm_reserve_reduction(Ctx &c, int n) {
    Comm *cm = c.comm;
    if (!cm || n <= cm->h_red_cap) return 0;
    if (cm->h_red) hipHostFree(cm->h_red);
    cm->h_red = nullptr;
    FEDM_HIP_CHECK(hipHostMalloc((void **)&cm->h_red, sizeof(double) * n));
    cm->h_red_cap = n;
    return 0;
}

void comm_allreduce(Ctx &c, double *d_buf, int n) {
    Comm *cm = c.comm;
    if (!cm || cm->kind == 0 || cm->failed) return;
    ++cm->n_allreduces;
    cm->allreduce_bytes += (long long)n * (long long)(cm->f32_payload_pending ? sizeof(float) : sizeof(double));
    cm->f32_payload_pending = false;
    if (cm->kind == 2) {
        nccl_ok(cm, g_nccl.AllReduce(d_buf, d_buf, (size_t)n, kNcclDouble, kNcclSum, cm->nccl, c.stream),
                "ncclAllReduce");
        return;
    }
    hipMemcpyAsync(cm->h_red, d_buf, sizeof(double) * n, hipMemcpyDeviceToHost, c.stream);
    hipStreamSynchronize(c.stream);
    cm->allreduce_cb(cm->h_red, n, cm->user);
    hipMemcpyAsync(d_buf, cm->h_red, sizeof(double) * n, hipMemcpyHostToDevice, c.stream);
}

__global__ void to_f32_kernel(int n, const double *__restrict__ x, float *__restrict__ y) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = (float)x[i];
}
__global__ void from_f32_kernel(int n, const float *__restrict__ x, double *__restrict__ y) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = (double)x[i];
}

// The same sum with a single-precision payload: for vectors that only a preconditioner reads (the
// level-1 right-hand side of the replicated multigrid hierarchy: n_global values per V-cycle, the
// largest message of a Krylov step).  RCCL: half the bytes on the links, summed in fp32.  Host-staged
// transport: the contributions are rounded to fp32 and summed in fp64 (same quantisation, so the
// one-GPU rehearsals see what the RCCL path sees up to the order of the sum).
void comm_allreduce_f32_payload(Ctx &c, double *d_buf, int n) {
    Comm *cm = c.comm;
    if (!cm || cm->kind == 0 || cm->failed) return;
    if (n > cm->red32_cap) {
        if (cm->d_red32) hipFree(cm->d_red32);
        cm->d_red32 = nullptr;
        cm->red32_cap = 0;
        if (hipMalloc((void **)&cm->d_red32, sizeof(float) * (size_t)n) != hipSuccess) {
            hipGetLastError();
            comm_allreduce(c, d_buf, n);   // no scratch: the double-precision path
            return;
        }
        cm->red32_cap = n;
    }
    const dim3 g((n + 255) / 256), b(256);
    hipLaunchKernelGGL(to_f32_kernel, g, b, 0, c.stream, n, d_buf, cm->d_red32);
    if (cm->kind == 2) {
        ++cm->n_allreduces;
        cm->allreduce_bytes += (long long)n * (long long)sizeof(float);
        nccl_ok(cm, g_nccl.AllReduce(cm->d_red32, cm->d_red32, (size_t)n, kNcclFloat, kNcclSum, cm->nccl, c.stream),
                "ncclAllReduce (fp32 payload)");
        hipLaunchKernelGGL(from_f32_kernel, g, b, 0, c.stream, n, cm->d_red32, d_buf);
        return;
    }
    hipLaunchKernelGGL(from_f32_kernel, g, b, 0, c.stream, n, cm->d_red32, d_buf);   // rounded contributions
    cm->f32_payload_pending = true;      // (counted as the fp32 payload the RCCL path moves)
    comm_allreduce(c, d_buf, n);
}

__global__ void halo_pack_kernel(int n_send, int neq, const int *__restrict__ idx,
                                 const double *__restrict__ vec, double *__restrict__ buf) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_send * neq) return;
    const int i = t / neq, s = t - i * neq;
    buf[t] = vec[(size_t)idx[i] * neq + s];
}

// the packed send buffer goes out, the neighbours' values arrive in recv_dst (n_ghost * w doubles,
// grouped by owner in neighbour order)
static void exchange_packed(Ctx &c, Comm *cm, double *recv_dst, hipStream_t st, int w) {
    if (cm->failed) return;
    ++cm->n_exchanges;
    cm->halo_bytes += (long long)cm->n_send * w * (long long)sizeof(double);
    if (cm->kind == 2) {
        // a group that was opened is always closed, whatever its members returned
        if (!nccl_ok(cm, g_nccl.GroupStart(), "ncclGroupStart")) return;
        for (int k = 0; k < cm->n_nb; ++k) {
            const int ns = cm->send_ptr[k + 1] - cm->send_ptr[k];
            const int nr = cm->recv_ptr[k + 1] - cm->recv_ptr[k];
            if (ns)
                nccl_ok(cm, g_nccl.Send(cm->d_sendbuf + (size_t)cm->send_ptr[k] * w, (size_t)ns * w, kNcclDouble,
                                        cm->nb_rank[k], cm->nccl, st), "ncclSend");
            if (nr)
                nccl_ok(cm, g_nccl.Recv(recv_dst + (size_t)cm->recv_ptr[k] * w, (size_t)nr * w, kNcclDouble,
                                        cm->nb_rank[k], cm->nccl, st), "ncclRecv");
        }
        nccl_ok(cm, g_nccl.GroupEnd(), "ncclGroupEnd");
        return;
    }
    if (cm->n_send)
        hipMemcpyAsync(cm->h_send, cm->d_sendbuf, sizeof(double) * cm->n_send * w, hipMemcpyDeviceToHost, st);
    hipStreamSynchronize(st);
    cm->exchange_cb(cm->h_send, cm->h_recv, w, cm->user);
    if (cm->n_ghost)
        hipMemcpyAsync(recv_dst, cm->h_recv, sizeof(double) * cm->n_ghost * w, hipMemcpyHostToDevice, st);
}

static void halo_on_stream(Ctx &c, Comm *cm, double *d_vec, hipStream_t st, int w) {
    if (cm->n_send)
        hipLaunchKernelGGL(halo_pack_kernel, dim3((cm->n_send * w + 255) / 256), dim3(256), 0, st,
                           cm->n_send, w, cm->d_send_idx, d_vec, cm->d_sendbuf);
    exchange_packed(c, cm, d_vec + (size_t)c.n_owned * w, st, w);
}

// the same for a compact single-precision vector ([vertex][w] floats: the iterate of the species
// sweeps): values travel as doubles through the same buffers
__global__ void halo_pack_f32_kernel(int n_send, int w, const int *__restrict__ idx,
                                     const float *__restrict__ vec, double *__restrict__ buf) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_send * w) return;
    const int i = t / w, s = t - i * w;
    buf[t] = (double)vec[(size_t)idx[i] * w + s];
}
__global__ void halo_unpack_f32_kernel(int n, const double *__restrict__ buf, float *__restrict__ ghost) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) ghost[t] = (float)buf[t];
}

static void halo_f32_on_stream(Ctx &c, Comm *cm, float *d_vec, int w, hipStream_t st) {
    if (cm->n_send)
        hipLaunchKernelGGL(halo_pack_f32_kernel, dim3((cm->n_send * w + 255) / 256), dim3(256), 0, st,
                           cm->n_send, w, cm->d_send_idx, d_vec, cm->d_sendbuf);
    exchange_packed(c, cm, cm->d_recvtmp, st, w);
    if (cm->n_ghost)
        hipLaunchKernelGGL(halo_unpack_f32_kernel, dim3((cm->n_ghost * w + 255) / 256), dim3(256), 0, st,
                           cm->n_ghost * w, cm->d_recvtmp, d_vec + (size_t)c.n_owned * w);
}

void comm_halo_f32(Ctx &c, float *d_vec, int w) {
    Comm *cm = c.comm;
    if (!cm || cm->kind == 0 || (cm->n_send == 0 && cm->n_ghost == 0)) return;
    halo_f32_on_stream(c, cm, d_vec, w, c.stream);
}

void comm_halo_exchange_scalar(Ctx &c, double *d_vec) {
    Comm *cm = c.comm;
    if (!cm || cm->kind == 0) return;
    hipStreamWaitEvent(cm->stream, cm->ev_ready, 0);
    if (cm->n_send || cm->n_ghost) halo_on_stream(c, cm, d_vec, cm->stream, 1);
    hipEventRecord(cm->ev_halo, cm->stream);
    hipStreamWaitEvent(c.stream, cm->ev_halo, 0);
}

// ... on the communication stream, after comm_halo_begin (see comm_halo_exchange)
void comm_halo_exchange_f32(Ctx &c, float *d_vec, int w) {
    Comm *cm = c.comm;
    if (!cm || cm->kind == 0) return;
    hipStreamWaitEvent(cm->stream, cm->ev_ready, 0);
    if (cm->n_send || cm->n_ghost) halo_f32_on_stream(c, cm, d_vec, w, cm->stream);
    hipEventRecord(cm->ev_halo, cm->stream);
    hipStreamWaitEvent(c.stream, cm->ev_halo, 0);
}

void comm_halo(Ctx &c, double *d_vec) {
    Comm *cm = c.comm;
    if (!cm || cm->kind == 0 || (cm->n_send == 0 && cm->n_ghost == 0)) return;
    halo_on_stream(c, cm, d_vec, c.stream, c.neq);
}

void comm_halo_scalar(Ctx &c, double *d_vec) {
    Comm *cm = c.comm;
    if (!cm || cm->kind == 0 || (cm->n_send == 0 && cm->n_ghost == 0)) return;
    halo_on_stream(c, cm, d_vec, c.stream, 1);
}

void comm_halo_begin(Ctx &c) {
    Comm *cm = c.comm;
    if (!cm || cm->kind == 0) return;
    hipEventRecord(cm->ev_ready, c.stream);  // the vector (and the send buffer's last use) complete
}

void comm_halo_exchange(Ctx &c, double *d_vec) {
    Comm *cm = c.comm;
    if (!cm || cm->kind == 0) return;
    hipStreamWaitEvent(cm->stream, cm->ev_ready, 0);
    if (cm->n_send || cm->n_ghost) halo_on_stream(c, cm, d_vec, cm->stream, c.neq);
    hipEventRecord(cm->ev_halo, cm->stream);
    hipStreamWaitEvent(c.stream, cm->ev_halo, 0);  // what follows on the compute stream sees the ghosts
}

}  // namespace fedm
