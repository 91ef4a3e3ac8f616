// Multi-GPU plumbing: one process per GPU, each owning a vertex partition.  Rows of owned
// vertices are assembled locally from every cell that touches them (ghost cells are evaluated
// on both sides), so the only traffic is (i) ghost *input* values of a vector before an SpMV or
// an assembly and (ii) small all-reduces for dot products and norms.
// Transports: RCCL over xGMI (ncclSend/ncclRecv groups + ncclAllReduce on the library's
// stream), or host-staged callbacks (any transport the caller owns, e.g. gloo; used by tests).
#pragma once
#include <vector>

#include "fedm_internal.hpp"

namespace fedm {

struct Comm {
    int kind = 0;  // 0 none, 1 host callbacks, 2 RCCL
    int rank = 0, nranks = 1;
    // halo plan (vertex units).  Ghost vertices follow the owned ones in local numbering,
    // grouped by owner in neighbour order, so every receive lands contiguously in the vector.
    int n_nb = 0;
    std::vector<int> nb_rank, send_ptr, recv_ptr;
    int n_send = 0, n_ghost = 0;
    int *d_send_idx = nullptr;
    double *d_sendbuf = nullptr, *d_recvtmp = nullptr;  // packed values out; staging for non-fp64 vectors
    double *h_send = nullptr, *h_recv = nullptr, *h_red = nullptr;
    int h_red_cap = 0;  // doubles (host-staged all-reduce)
    float *d_red32 = nullptr;   // scratch of comm_allreduce_f32_payload
    int red32_cap = 0;
    fedm_allreduce_fn allreduce_cb = nullptr;
    fedm_exchange_fn exchange_cb = nullptr;
    void *user = nullptr;
    void *nccl = nullptr;  // ncclComm_t
    // overlap of the halo exchange with the interior rows of the Krylov SpMV: the exchange runs
    // on its own stream between two events; matrix slices are split into those that reference
    // no ghost column (interior) and the rest (boundary)
    hipStream_t stream = nullptr;
    hipEvent_t ev_ready = nullptr, ev_halo = nullptr;
    int n_interior = 0, n_boundary = 0;
    int *d_interior = nullptr, *d_boundary = nullptr;
    // the same split for the assembly patches (a patch = a matrix slice): interior patches stage
    // no ghost vertex, so the state halo can travel while they are assembled
    int n_patch_interior = 0, n_patch_boundary = 0;
    int *d_patch_interior = nullptr, *d_patch_boundary = nullptr;
    // A transport error latches here (first message kept, also in fedm_last_error): from then on
    // every exchange / reduction is skipped and the solver entry points return -1 -- a lost peer
    // must read as an error, not as NaNs or a hang.
    bool failed = false;
    std::string error;
    long n_exchanges = 0, n_allreduces = 0;   // issued so far (bench.py reports them per step)
    bool f32_payload_pending = false;
    long long halo_bytes = 0, allreduce_bytes = 0;   // payload this rank sent in halo exchanges / contributed to all-reduces
    void release();
    bool release_communicator();   // ncclCommAbort after a latched failure, ncclCommDestroy otherwise; true: aborted
};

int comm_setup_plan(Ctx &c, Comm &cm, int n_nb, const int32_t *nb_rank, const int32_t *send_ptr,
                    const int32_t *send_idx, const int32_t *recv_ptr);
int comm_init_rccl(Ctx &c, Comm &cm, const void *unique_id, int rank, int nranks);
int comm_unique_id(void *out128);
void comm_allreduce(Ctx &c, double *d_buf, int n);  // sum over ranks, in place, stream-ordered
void comm_allreduce_f32_payload(Ctx &c, double *d_buf, int n);  // the same with fp32 on the wire
int comm_reserve_reduction(Ctx &c, int n);          // host-staged transport: room for n doubles
void comm_halo(Ctx &c, double *d_vec);              // refresh ghost vertices of a block vector
void comm_halo_scalar(Ctx &c, double *d_vec);       // the same for one value per vertex
void comm_halo_f32(Ctx &c, float *d_vec, int w);    // the same for [vertex][w] floats
// The same exchange on the communication stream, overlapped with compute work: comm_halo_begin
// marks the point of the compute stream at which the vector is complete; work queued on the
// compute stream after it runs concurrently with comm_halo_exchange, which performs the exchange
// on the communication stream and makes the compute stream wait for its end.
void comm_halo_begin(Ctx &c);
void comm_halo_exchange(Ctx &c, double *d_vec);
void comm_halo_exchange_f32(Ctx &c, float *d_vec, int w);
void comm_halo_exchange_scalar(Ctx &c, double *d_vec);
int comm_fault_selftest(int fail_at, int64_t out[6]);  // error-path test hook (no GPU needed)
bool comm_failed(const Ctx &c);                     // a transport error has been latched
bool comm_poll_async_error(Ctx &c);                 // RCCL's asynchronous error state; latches, true on error

}  // namespace fedm
