// LDS atomic throughput on gfx950: lanes add to pseudo-random slots of a 24 KB LDS array.
// hipcc --offload-arch=gfx950 -O3 tools/microbench/lds_atomics.hip -o gpurun_out/lds_atomics && gpurun_out/lds_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr int NSLOT = 2048, ITER = 4096;
template <int MODE>
__global__ void __launch_bounds__(256) k(double *out, int stride) {
    __shared__ unsigned long long a[NSLOT];
    for (int i = threadIdx.x; i < NSLOT; i += 256) a[i] = 0;
    __syncthreads();
    // stride 0: a fixed pseudo-random slot per lane, all lanes advance together (random bank pattern, no same-address hits
    // beyond chance); stride s: lane-consecutive slots s apart (1: conflict-free)
    const unsigned x = (threadIdx.x * 2654435761u + blockIdx.x * 40503u + 1u) >> 8;
    const unsigned s0 = stride ? (x & ~63u) + (threadIdx.x & 63) * stride : x;
#pragma unroll 8
    for (int it = 0; it < ITER; ++it) {
        const unsigned slot = (s0 + it * 67u) & (NSLOT - 1);
        if (MODE == 0) atomicAdd((double *)&a[slot], 1.0);
        else if (MODE == 1) atomicAdd(&a[slot], 1ull);
        else if (MODE == 2) atomicAdd((unsigned *)&a[slot], 1u);
        else if (MODE == 3) atomicAdd((float *)&a[slot], 1.0f);
        else { double *p = (double *)&a[slot]; *p = *p + 1.0; }           // plain read-add-write (racy: rate only)
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (double)a[0];
}
template <int MODE> void run(const char *name, int stride) {
    double *out; hipMalloc(&out, 4096 * sizeof(double));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 4;
    k<MODE><<<blocks, 256>>>(out, stride); hipDeviceSynchronize();
    hipEventRecord(e0); k<MODE><<<blocks, 256>>>(out, stride); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)blocks * 256 * ITER;
    printf("%-28s stride %d: %.3f ms, %.1f G lane-ops/s, %.2f lanes/clk/CU (256 CUs @2.4 GHz)\n", name, stride, ms, ops / ms / 1e6,
           ops / (ms * 1e-3) / 256 / 2.4e9);
    hipFree(out);
}
int main() {
    for (int stride : {0, 1, 2, 8}) {
        run<0>("ds_add_f64", stride); run<1>("ds_add_u64", stride); run<2>("ds_add_u32", stride); run<3>("ds_add_f32", stride);
        run<4>("ds_read+ds_write f64", stride);
    }
    return 0;
}
