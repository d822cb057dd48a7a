// What a consumer kernel finds in the 256 MiB Infinity Cache of what its producer left there, for buffers of the far-field grids' size
// (406 MB: larger than the cache): producer = plain / non-temporal stores, an in-place read-modify-write, or a copy from a second
// buffer; consumer = a streamed read (or in-place read-modify-write) in the producer's block order or in the reverse one.
// Prints the consumer's GB/s.   hipcc --offload-arch=gfx950 -O3 tools/microbench/mall_order.hip -o tools/microbench/mall_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2v __attribute__((ext_vector_type(2)));
constexpr int ITER = 16;                       // 256 threads x 16 B x 16 = 64 KB per block, contiguous
__device__ __forceinline__ size_t chunk(int rev) { return (size_t)(rev ? gridDim.x - 1 - blockIdx.x : blockIdx.x) * 256 * ITER; }
template <bool NT> __global__ void __launch_bounds__(256) k_write(d2v *a, int rev) {
    d2v *p = a + chunk(rev) + threadIdx.x;
    d2v v; v.x = 1.0; v.y = 2.0;
#pragma unroll
    for (int i = 0; i < ITER; ++i) { if (NT) __builtin_nontemporal_store(v, p + 256 * i); else p[256 * i] = v; }
}
template <bool NT, bool NTS = NT> __global__ void __launch_bounds__(256) k_rmw(d2v *a, int rev) {
    d2v *p = a + chunk(rev) + threadIdx.x;
    d2v v[ITER];
#pragma unroll
    for (int i = 0; i < ITER; ++i) v[i] = NT ? __builtin_nontemporal_load(p + 256 * i) : p[256 * i];
#pragma unroll
    for (int i = 0; i < ITER; ++i) { v[i].x += 1.0; if (NTS) __builtin_nontemporal_store(v[i], p + 256 * i); else p[256 * i] = v[i]; }
}
template <bool NT, bool NTS = NT> __global__ void __launch_bounds__(256) k_copy(const d2v *b, d2v *a, int rev) {
    const size_t c = chunk(rev) + threadIdx.x;
    d2v v[ITER];
#pragma unroll
    for (int i = 0; i < ITER; ++i) v[i] = NT ? __builtin_nontemporal_load(b + c + 256 * i) : b[c + 256 * i];
#pragma unroll
    for (int i = 0; i < ITER; ++i) { if (NTS) __builtin_nontemporal_store(v[i], a + c + 256 * i); else a[c + 256 * i] = v[i]; }
}
template <bool NT> __global__ void __launch_bounds__(256) k_read(const d2v *a, int rev, double *out) {
    const d2v *p = a + chunk(rev) + threadIdx.x;
    double acc = 0;
#pragma unroll
    for (int i = 0; i < ITER; ++i) { const d2v v = NT ? __builtin_nontemporal_load(p + 256 * i) : p[256 * i]; acc += v.x + v.y; }
    if (acc == 0.12345) out[0] = acc;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
int main(int argc, char **argv) {
    const size_t mb = argc > 1 ? atoi(argv[1]) : 406;
    const int nb = (int)(mb * 1000000 / (256 * ITER * 16));
    const size_t n = (size_t)nb * 256 * ITER;
    d2v *A, *B; double *out;
    CK(hipMalloc(&A, n * 16)); CK(hipMalloc(&B, n * 16)); CK(hipMalloc(&out, 8));
    CK(hipMemset(A, 0, n * 16)); CK(hipMemset(B, 0, n * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("buffer %.1f MB, %d blocks of 64 KB\n", n * 16 / 1e6, nb);
    const char *pn[] = {"write plain", "write nt", "rmw plain", "rmw nt", "copy plain", "copy nt", "rmw ntL/plainS", "copy ntL/plainS"};
    const char *cn[] = {"read plain", "read nt", "rmw plain", "rmw ntL/plainS", "copy B->A ntL/plainS", "copy B->A nt", "copy A->B ntL/plainS"};
    for (int prod = 0; prod < 8; ++prod)
        for (int cons = 0; cons < 7; ++cons)
            for (int rev = 0; rev < 2; ++rev) {
                float best = 1e9f, sum = 0; float pbest = 1e9f;
                for (int rep = 0; rep < 6; ++rep) {
                    CK(hipEventRecord(e0));
                    switch (prod) {
                    case 0: k_write<false><<<nb, 256>>>(A, 0); break;
                    case 1: k_write<true><<<nb, 256>>>(A, 0); break;
                    case 2: k_rmw<false><<<nb, 256>>>(A, 0); break;
                    case 3: k_rmw<true><<<nb, 256>>>(A, 0); break;
                    case 4: k_copy<false><<<nb, 256>>>(B, A, 0); break;
                    case 5: k_copy<true><<<nb, 256>>>(B, A, 0); break;
                    case 6: k_rmw<true, false><<<nb, 256>>>(A, 0); break;
                    default: k_copy<true, false><<<nb, 256>>>(B, A, 0); break;
                    }
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float pms; CK(hipEventElapsedTime(&pms, e0, e1)); if (pms < pbest) pbest = pms;
                    CK(hipEventRecord(e0));
                    if (cons == 0) k_read<false><<<nb, 256>>>(A, rev, out);
                    else if (cons == 1) k_read<true><<<nb, 256>>>(A, rev, out);
                    else if (cons == 2) k_rmw<false><<<nb, 256>>>(A, rev);
                    else if (cons == 3) k_rmw<true, false><<<nb, 256>>>(A, rev);
                    else if (cons == 4) k_copy<true, false><<<nb, 256>>>(B, A, rev);
                    else if (cons == 5) k_copy<true, true><<<nb, 256>>>(B, A, rev);
                    else k_copy<true, false><<<nb, 256>>>(A, B, rev);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep) { sum += ms; if (ms < best) best = ms; }
                }
                const double bytes = n * 16.0 * (cons >= 2 ? 2 : 1);
                printf("%-15s (%.0f us) -> %-20s %s: %7.1f us mean, %7.1f best = %.2f TB/s\n", pn[prod], pbest * 1e3, cn[cons], rev ? "reversed" : "in order", sum / 5 * 1e3, best * 1e3,
                       bytes / (sum / 5 * 1e-3) / 1e12);
            }
    return 0;
}
