// Micro-test of global_load_lds_dwordx4 semantics on gfx950: where does each lane's data land?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void *lds_ptr;

__global__ void k(const double *__restrict__ a, const int *__restrict__ idx, double *__restrict__ out, int n) {
    __shared__ __align__(16) double buf[2 * 256];
    const int i = threadIdx.x;
    buf[2 * i] = -1.0;
    buf[2 * i + 1] = -1.0;
    __syncthreads();
    const int g = idx[i];
    if (g >= 0 && g < n)   // 16 bytes at a + 3 g + 1 (8-byte aligned only), into buf[2 * i .. 2 * i + 1]
        __builtin_amdgcn_global_load_lds(a + 3 * g + 1, (lds_ptr)(buf + 2 * (i & ~63)), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[2 * i] = buf[2 * i];
    out[2 * i + 1] = buf[2 * i + 1];
}

int main() {
    const int n = 1000, T = 256;
    std::vector<double> a(3 * n);
    for (int i = 0; i < 3 * n; ++i) a[i] = i;
    std::vector<int> idx(T);
    for (int i = 0; i < T; ++i) idx[i] = (i % 7 == 3) ? -1 : (i * 37) % n;   // some lanes inactive
    double *da, *dout;
    int *didx;
    hipMalloc(&da, a.size() * 8);
    hipMalloc(&dout, 2 * T * 8);
    hipMalloc(&didx, T * 4);
    hipMemcpy(da, a.data(), a.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(didx, idx.data(), T * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(T), 0, 0, da, didx, dout, n);
    std::vector<double> out(2 * T);
    if (hipMemcpy(out.data(), dout, 2 * T * 8, hipMemcpyDeviceToHost) != hipSuccess) { printf("HIP error\n"); return 2; }
    int bad = 0;
    for (int i = 0; i < T; ++i) {
        const double e0 = idx[i] < 0 ? -1.0 : 3 * idx[i] + 1, e1 = idx[i] < 0 ? -1.0 : 3 * idx[i] + 2;
        if (out[2 * i] != e0 || out[2 * i + 1] != e1) {
            if (bad < 5) printf("lane %d: got %g %g expected %g %g\n", i, out[2 * i], out[2 * i + 1], e0, e1);
            ++bad;
        }
    }
    printf("mismatches: %d of %d\n", bad, T);
    return bad ? 1 : 0;
}
