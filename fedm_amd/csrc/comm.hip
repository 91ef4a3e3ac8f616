#include "comm.hpp"

#include <dlfcn.h>

#include <cstring>

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
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
NcclApi g_nccl;
constexpr int kNcclDouble = 8;  // ncclFloat64
constexpr int kNcclSum = 0;

int load_nccl() {
    if (g_nccl.lib) return 0;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        g_nccl.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
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
    return 0;
}
}  // namespace

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

void Comm::release() {
    if (nccl && g_nccl.CommDestroy) g_nccl.CommDestroy(nccl);
    nccl = nullptr;
    if (d_send_idx) hipFree(d_send_idx);
    if (d_sendbuf) hipFree(d_sendbuf);
    if (h_send) hipHostFree(h_send);
    if (h_recv) hipHostFree(h_recv);
    if (h_red) hipHostFree(h_red);
    d_send_idx = nullptr;
    d_sendbuf = nullptr;
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
    FEDM_HIP_CHECK(hipHostMalloc((void **)&cm.h_send, sizeof(double) * ns * c.neq));
    FEDM_HIP_CHECK(hipHostMalloc((void **)&cm.h_recv, sizeof(double) * ng * c.neq));
    FEDM_HIP_CHECK(hipHostMalloc((void **)&cm.h_red, sizeof(double) * RED_K));
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

void comm_allreduce(Ctx &c, double *d_buf, int n) {
    Comm *cm = c.comm;
    if (!cm || cm->kind == 0) return;
    if (cm->kind == 2) {
        g_nccl.AllReduce(d_buf, d_buf, (size_t)n, kNcclDouble, kNcclSum, cm->nccl, c.stream);
        return;
    }
    hipMemcpyAsync(cm->h_red, d_buf, sizeof(double) * n, hipMemcpyDeviceToHost, c.stream);
    hipStreamSynchronize(c.stream);
    cm->allreduce_cb(cm->h_red, n, cm->user);
    hipMemcpyAsync(d_buf, cm->h_red, sizeof(double) * n, hipMemcpyHostToDevice, c.stream);
}

__global__ void halo_pack_kernel(int n_send, int neq, const int *__restrict__ idx,
                                 const double *__restrict__ vec, double *__restrict__ buf) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_send * neq) return;
    const int i = t / neq, s = t - i * neq;
    buf[t] = vec[(size_t)idx[i] * neq + s];
}

void comm_halo(Ctx &c, double *d_vec) {
    Comm *cm = c.comm;
    if (!cm || cm->kind == 0 || (cm->n_send == 0 && cm->n_ghost == 0)) return;
    const int w = c.neq;
    if (cm->n_send)
        hipLaunchKernelGGL(halo_pack_kernel, dim3((cm->n_send * w + 255) / 256), dim3(256), 0, c.stream,
                           cm->n_send, w, cm->d_send_idx, d_vec, cm->d_sendbuf);
    double *ghost = d_vec + (size_t)c.n_owned * w;
    if (cm->kind == 2) {
        g_nccl.GroupStart();
        for (int k = 0; k < cm->n_nb; ++k) {
            const int ns = cm->send_ptr[k + 1] - cm->send_ptr[k];
            const int nr = cm->recv_ptr[k + 1] - cm->recv_ptr[k];
            if (ns)
                g_nccl.Send(cm->d_sendbuf + (size_t)cm->send_ptr[k] * w, (size_t)ns * w, kNcclDouble,
                            cm->nb_rank[k], cm->nccl, c.stream);
            if (nr)
                g_nccl.Recv(ghost + (size_t)cm->recv_ptr[k] * w, (size_t)nr * w, kNcclDouble,
                            cm->nb_rank[k], cm->nccl, c.stream);
        }
        g_nccl.GroupEnd();
        return;
    }
    if (cm->n_send)
        hipMemcpyAsync(cm->h_send, cm->d_sendbuf, sizeof(double) * cm->n_send * w, hipMemcpyDeviceToHost, c.stream);
    hipStreamSynchronize(c.stream);
    cm->exchange_cb(cm->h_send, cm->h_recv, w, cm->user);
    if (cm->n_ghost)
        hipMemcpyAsync(ghost, cm->h_recv, sizeof(double) * cm->n_ghost * w, hipMemcpyHostToDevice, c.stream);
}

}  // namespace fedm
