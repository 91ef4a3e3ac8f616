// How long does a barrier over G resident workgroups take on this GPU?  (sense-free counter barrier in device memory:
// one agent-scope atomic add per workgroup, then spinning loads; a fence on either side so that plain stores of one
// phase are visible to every XCD in the next.)  Decides whether the coarse half of the multigrid cycle -- five
// dependent 5 us kernels -- could be one kernel with barriers between its stages.
//   hipcc --offload-arch=gfx950 -O3 -o bench bench.hip && ./bench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

__device__ __forceinline__ void grid_barrier(unsigned *counter, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < (1u << 24))
            __builtin_amdgcn_s_sleep(1);
        __threadfence();
    }
    __syncthreads();
}

// every phase: each thread reads a value another workgroup wrote in the phase before (checks visibility), writes one
__global__ void phases_kernel(unsigned *counter, double *buf, int n_phases, int *errors) {
    const int G = gridDim.x, T = blockDim.x;
    const int me = blockIdx.x * T + threadIdx.x, n = G * T;
    buf[me] = 1.0;
    for (int p = 1; p <= n_phases; ++p) {
        grid_barrier(counter, (unsigned)(p * G));
        const int other = (me + T * (1 + p % (G > 1 ? G - 1 : 1))) % n;   // a slot of another workgroup
        const double v = __builtin_nontemporal_load(&buf[(size_t)((p - 1) & 1) * n + other]);
        if (v != (double)p) atomicAdd(errors, 1);
        buf[(size_t)(p & 1) * n + me] = (double)(p + 1);
    }
}

int main() {
    unsigned *counter;
    double *buf;
    int *errors;
    hipMalloc(&counter, sizeof(unsigned));
    hipMalloc(&errors, sizeof(int));
    hipMalloc(&buf, sizeof(double) * 2 * 1024 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int G : {8, 32, 64, 128, 256, 512}) {
        for (int phases : {1, 101}) {
            hipMemset(counter, 0, sizeof(unsigned));
            hipMemset(errors, 0, sizeof(int));
            hipMemset(buf, 0, sizeof(double) * 2 * 1024 * 1024);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(phases_kernel, dim3(G), dim3(256), 0, 0, counter, buf, phases, errors);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            int err = 0;
            hipMemcpy(&err, errors, sizeof(int), hipMemcpyDeviceToHost);
            std::printf("G = %3d workgroups, %3d barriers: %8.2f us total%s, stale reads %d\n", G, phases, 1e3 * ms,
                        phases > 1 ? "" : " (launch + one barrier)", err);
        }
    }
    return 0;
}
