// Cost of a COALESCED wave load on gfx950 by width (lane l reads element base + l): dword, dwordx2, dwordx4 from an
// L2-resident (768 KB) and an L1-resident (12 KB) buffer -- what the pair list's (index, f, h) stream pays per instruction.
// hipcc --offload-arch=gfx950 -O3 tools/microbench/stream_widths.hip -o tools/microbench/stream_widths
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int ITER = 512;
template <typename T> __device__ double val(T v);
template <> __device__ double val(unsigned v) { return (double)v; }
template <> __device__ double val(double v) { return v; }
template <> __device__ double val(double2 v) { return v.x + v.y; }
template <typename T>
__global__ void __launch_bounds__(256) k(const T *__restrict__ tab, double *out, unsigned mask) {
    unsigned x = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2654435761u + 12345u;
    const int lane = threadIdx.x & 63;
    double acc = 0;
#pragma unroll 8
    for (int it = 0; it < ITER; ++it) {
        x = x * 1664525u + 1013904223u;
        const unsigned row = (x >> 10) & mask;          // wave-uniform random row of 64 elements
        acc += val(tab[(size_t)row * 64 + lane]);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <typename T> void run(const char *name, size_t bytes) {
    T *tab; double *out;
    (void)hipMalloc(&tab, 64 << 20); (void)hipMemset(tab, 0, 64 << 20);
    const int blocks = 256 * 8;
    (void)hipMalloc(&out, (size_t)blocks * 256 * sizeof(double));
    const unsigned mask = (unsigned)(bytes / (64 * sizeof(T))) - 1;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<T><<<blocks, 256>>>(tab, out, mask); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); k<T><<<blocks, 256>>>(tab, out, mask); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-10s %8zu KB  %.3f ms: %.1f clk per wave-instruction per CU (2.1 GHz)\n", name, bytes >> 10, ms,
           ms * 1e-3 * 2.1e9 / ((double)blocks * 4 * ITER / 256));
    (void)hipFree(tab); (void)hipFree(out);
}
int main() {
    for (size_t b : {(size_t)16 << 10, (size_t)1 << 20, (size_t)64 << 20}) {
        run<unsigned>("dword", b); run<double>("dwordx2", b); run<double2>("dwordx4", b);
    }
    return 0;
}
