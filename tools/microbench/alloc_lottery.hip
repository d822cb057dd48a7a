// Does the rate of a plain streaming pass over a buffer depend on WHICH allocation it is?  K buffers of the same size allocated one after
// another (all alive), each timed several times: a non-temporal read, an in-place read-modify-write, and a copy into a second buffer.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/alloc_lottery.hip -o tools/microbench/alloc_lottery ; ./alloc_lottery [MB] [K]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2v __attribute__((ext_vector_type(2)));
constexpr int ITER = 16;
__global__ void __launch_bounds__(256) k_read(const d2v *a, double *out) {
    const d2v *p = a + (size_t)blockIdx.x * 256 * ITER + threadIdx.x;
    double acc = 0;
#pragma unroll
    for (int i = 0; i < ITER; ++i) { const d2v v = __builtin_nontemporal_load(p + 256 * i); acc += v.x + v.y; }
    if (acc == 0.12345) out[0] = acc;
}
__global__ void __launch_bounds__(256) k_rmw(d2v *a) {
    d2v *p = a + (size_t)blockIdx.x * 256 * ITER + threadIdx.x;
    d2v v[ITER];
#pragma unroll
    for (int i = 0; i < ITER; ++i) v[i] = p[256 * i];
#pragma unroll
    for (int i = 0; i < ITER; ++i) { v[i].x += 1.0; p[256 * i] = v[i]; }
}
__global__ void __launch_bounds__(256) k_copy(const d2v *b, d2v *a) {
    const size_t c = (size_t)blockIdx.x * 256 * ITER + threadIdx.x;
    d2v v[ITER];
#pragma unroll
    for (int i = 0; i < ITER; ++i) v[i] = __builtin_nontemporal_load(b + c + 256 * i);
#pragma unroll
    for (int i = 0; i < ITER; ++i) __builtin_nontemporal_store(v[i], a + c + 256 * i);
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
int main(int argc, char **argv) {
    const size_t mb = argc > 1 ? atoi(argv[1]) : 600;
    const int K = argc > 2 ? atoi(argv[2]) : 8;
    const int nb = (int)(mb * 1000000 / (256 * ITER * 16));
    const size_t n = (size_t)nb * 256 * ITER;
    d2v *A[16]; double *out;
    CK(hipMalloc(&out, 8));
    for (int k = 0; k < K; ++k) { CK(hipMalloc(&A[k], n * 16)); CK(hipMemset(A[k], 0, n * 16)); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("%d buffers of %.1f MB\n", K, n * 16 / 1e6);
    for (int round = 0; round < 2; ++round)
        for (int k = 0; k < K; ++k) {
            float best[3] = {1e9f, 1e9f, 1e9f};
            for (int rep = 0; rep < 6; ++rep)
                for (int op = 0; op < 3; ++op) {
                    CK(hipEventRecord(e0));
                    if (op == 0) k_read<<<nb, 256>>>(A[k], out);
                    else if (op == 1) k_rmw<<<nb, 256>>>(A[k]);
                    else k_copy<<<nb, 256>>>(A[k], A[(k + 1) % K]);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep && ms < best[op]) best[op] = ms;
                }
            printf("round %d buffer %2d at %p: read %6.1f us = %.2f TB/s | rmw %6.1f us = %.2f TB/s | copy to the next %6.1f us = %.2f TB/s\n", round, k, (void *)A[k],
                   best[0] * 1e3, n * 16.0 / best[0] / 1e9, best[1] * 1e3, 2 * n * 16.0 / best[1] / 1e9, best[2] * 1e3, 2 * n * 16.0 / best[2] / 1e9);
        }
    return 0;
}
