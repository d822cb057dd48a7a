// A buffer larger than the Infinity Cache read TWICE in a row (the pair list of consecutive Lanczos mat-vecs): does the second pass find
// anything on-die, and does it matter whether it walks the blocks in the same or in the opposite order?  Plain and non-temporal loads.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/reread_order.hip -o tools/microbench/reread_order ; ./reread_order [MB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2v __attribute__((ext_vector_type(2)));
constexpr int ITER = 16;
template <bool NT> __global__ void __launch_bounds__(256) k_read(const d2v *a, int rev, double *out) {
    const d2v *p = a + (size_t)(rev ? gridDim.x - 1 - blockIdx.x : blockIdx.x) * 256 * ITER + threadIdx.x;
    double acc = 0;
#pragma unroll
    for (int i = 0; i < ITER; ++i) { const d2v v = NT ? __builtin_nontemporal_load(p + 256 * i) : p[256 * i]; acc += v.x + v.y; }
    if (acc == 0.12345) out[0] = acc;
}
__global__ void __launch_bounds__(256) k_touch(d2v *b, int n) {   // a small streaming kernel between the passes (the vector update: 128 MB)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)n; i += (size_t)gridDim.x * 256) { d2v v = b[i]; v.x += 1.0; b[i] = v; }
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
int main(int argc, char **argv) {
    const size_t mb = argc > 1 ? atoi(argv[1]) : 474;
    const int nb = (int)(mb * 1000000 / (256 * ITER * 16));
    const size_t n = (size_t)nb * 256 * ITER;
    d2v *A, *B; double *out;
    const int nB = 4 * 1000 * 1000;   // 64 MB read + written
    CK(hipMalloc(&A, n * 16)); CK(hipMalloc(&B, (size_t)nB * 16)); CK(hipMalloc(&out, 8));
    CK(hipMemset(A, 0, n * 16)); CK(hipMemset(B, 0, (size_t)nB * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("buffer %.1f MB\n", n * 16 / 1e6);
    for (int nt = 0; nt < 2; ++nt)
        for (int touch = 0; touch < 2; ++touch)
            for (int alt = 0; alt < 2; ++alt) {
                float sum = 0; int cnt = 0;
                for (int rep = 0; rep < 12; ++rep) {
                    const int rev = alt ? (rep & 1) : 0;
                    if (touch) k_touch<<<2048, 256>>>(B, nB);
                    CK(hipEventRecord(e0));
                    if (nt) k_read<true><<<nb, 256>>>(A, rev, out); else k_read<false><<<nb, 256>>>(A, rev, out);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep >= 2) { sum += ms; ++cnt; }
                }
                printf("%s loads, %s between passes, %s: %.1f us per pass = %.2f TB/s\n", nt ? "non-temporal" : "plain", touch ? "a 128 MB update" : "nothing",
                       alt ? "ALTERNATING block order" : "same block order", sum / cnt * 1e3, n * 16.0 / (sum / cnt) / 1e9);
            }
    return 0;
}
