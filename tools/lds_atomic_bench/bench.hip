// Throughput of LDS fp64 accumulation on gfx950: ds_add_f64 (unsafeAtomicAdd) against plain
// ds_read_b64 + add + ds_write_b64 and bare ds_write_b64, with the access shapes of the patch
// assembly: every lane adds into column (vertex mod 64) of plane p -- conflict-free (all lanes
// distinct columns), 2-way (lanes l and l+16 share a bank class), same-address pairs.
// build: hipcc --offload-arch=gfx950 -O3 -o bench bench.hip ; run: ./bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int THREADS = 192, PLANES = 64, ITER = 64;   // 32 KB of accumulators per workgroup

template <int MODE>
__global__ __launch_bounds__(THREADS) void k(const int *__restrict__ col, double *__restrict__ out, int reps) {
    __shared__ double acc[PLANES * 64];
    for (int i = threadIdx.x; i < PLANES * 64; i += THREADS) acc[i] = 0.0;
    __syncthreads();
    const int c = col[blockIdx.x % 8 * THREADS + threadIdx.x];
    double v = 1.0 + threadIdx.x;
    for (int r = 0; r < reps; ++r) {
#pragma unroll 8
        for (int it = 0; it < ITER; ++it) {
            const int p = (it * 7 + (threadIdx.x >> 6)) & (PLANES - 1);
            double *a = &acc[p * 64 + c];
            if (MODE == 0) unsafeAtomicAdd(a, v);
            else if (MODE == 1) *a = *a + v;          // non-atomic read-modify-write (owner-exclusive use)
            else *a = v;                               // store only
            v += 1.0;
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) out[blockIdx.x * 64 + threadIdx.x] = acc[threadIdx.x];
}

template <int MODE>
float run(const int *dcol, double *dout, int blocks, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(THREADS), 0, 0, dcol, dout, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(THREADS), 0, 0, dcol, dout, reps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    const int blocks = 256 * 4, reps = 50;
    int *dcol;
    double *dout;
    hipMalloc(&dcol, sizeof(int) * 8 * THREADS);
    hipMalloc(&dout, sizeof(double) * blocks * 64);
    const char *names[] = {"distinct columns (conflict-free)", "lanes l, l+16 same bank class (2-way)",
                           "pairs of lanes same address", "4 lanes same address"};
    for (int shape = 0; shape < 4; ++shape) {
        std::vector<int> col(8 * THREADS);
        for (int b = 0; b < 8; ++b)
            for (int t = 0; t < THREADS; ++t) {
                const int l = t & 63;
                int c = l;
                // 2-way: within every 16-lane group, lanes k and k+8 hit columns that differ by 16 (same
                // bank class mod 16), all 64 columns distinct
                if (shape == 1) c = (l & 7) + 16 * ((l >> 3) & 1) + 8 * ((l >> 4) & 1) + 32 * ((l >> 5) & 1);
                if (shape == 2) c = l >> 1;
                if (shape == 3) c = l >> 2;
                col[b * THREADS + t] = c % 64;
            }
        hipMemcpy(dcol, col.data(), sizeof(int) * col.size(), hipMemcpyHostToDevice);
        const float a = run<0>(dcol, dout, blocks, reps), b = run<1>(dcol, dout, blocks, reps), c = run<2>(dcol, dout, blocks, reps);
        const double ops = (double)blocks * 3 /*waves*/ * ITER * reps;  // wave-instructions
        // 4 workgroups per CU -> 12 waves share one LDS; cycles per wave-instruction per CU at 2.4 GHz
        auto cyc = [&](float ms) { return ms * 1e-3 * 2.4e9 / (ops / 256.0); };
        printf("%-42s atomic %.3f ms (%.1f clk/instr/CU)  rmw %.3f ms (%.1f)  store %.3f ms (%.1f)\n", names[shape], a, cyc(a), b,
               cyc(b), c, cyc(c));
    }
    return 0;
}
