// Scattered-gather throughput of the texture addresser on gfx950: 48-byte records of a 48 MB table (L2/MALL-resident), fetched
//   A  one 16-byte load per lane, 64 random records per wave-instruction
//   B  three 16-byte loads per lane (the whole record), 64 random records per instruction  [the pair-list mat-vec]
//   C  one 16-byte load per lane, 16 random records per instruction: lane = (record, 16-byte part), parts 0..2 (+1 idle lane)
//   D  as C with 21 records per instruction (3 lanes per record, no idle lane)
// hipcc --offload-arch=gfx950 -O3 tools/microbench/gather_patterns.hip -o gpurun_out/gather_patterns && gpurun_out/gather_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr int ITER = 512;
__constant__ int NREC;   // records in the table (a power of two): 1 << 20 = 48 MB (L2/MALL), 1 << 8 = 12 KB (L1-resident)
template <int MODE>
__global__ void __launch_bounds__(256) k(const double2 *__restrict__ tab, double *out) {
    const int lane = threadIdx.x & 63;
    unsigned x = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    double2 acc = make_double2(0, 0);
#pragma unroll 4
    for (int it = 0; it < ITER; ++it) {
        x = x * 1664525u + 1013904223u;
        unsigned r = x >> 12;
        if (MODE == 2) r = __shfl(r, lane & ~3, 64);
        if (MODE == 3) r = __shfl(r, (lane / 3) * 3, 64);
        r &= NREC - 1;
        const double2 *p = tab + 3 * (size_t)r;
        if (MODE == 0) { const double2 a = p[0]; acc.x += a.x; acc.y += a.y; }
        if (MODE == 1) { const double2 a = p[0], b = p[1], c = p[2]; acc.x += a.x + b.x + c.x; acc.y += a.y + b.y + c.y; }
        if (MODE == 2) { const double2 a = p[min(lane & 3, 2)]; acc.x += a.x; acc.y += a.y; }
        if (MODE == 3) { const double2 a = p[lane % 3]; acc.x += a.x; acc.y += a.y; }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y;
}
template <int MODE> void run(const char *name, double recs_per_instr, int instr_per_iter) {
    double2 *tab; double *out;
    (void)hipMalloc(&tab, (size_t)(1 << 20) * 48); (void)hipMemset(tab, 0, (size_t)(1 << 20) * 48);
    const int blocks = 256 * 8;
    (void)hipMalloc(&out, (size_t)blocks * 256 * sizeof(double));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(tab, out); hipDeviceSynchronize();
    hipEventRecord(e0); k<MODE><<<blocks, 256>>>(tab, out); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double winstr = (double)blocks * 4 * ITER * instr_per_iter;
    const double recs = (double)blocks * 4 * ITER * recs_per_instr * (MODE == 1 ? 1 : 1);
    printf("%-44s %.3f ms: %.1f clk per wave-instruction per CU, %.2f clk per whole 48-byte record per CU\n", name, ms,
           ms * 1e-3 * 2.1e9 / (winstr / 256), ms * 1e-3 * 2.1e9 / (recs / 256) * (MODE == 0 ? 3 : 1));
    hipFree(tab); hipFree(out);
}
int main() {
    for (int lg : {20, 14, 8}) {
        const int n = 1 << lg;
        hipMemcpyToSymbol(HIP_SYMBOL(NREC), &n, sizeof n);
        printf("table of %d records (%.1f KB)\n", n, n * 48 / 1024.0);
        run<0>("A 64 records x 16 B (x3 for a record)", 64, 1);
        run<1>("B 64 records x 3 x 16 B", 64, 3);
        run<2>("C 16 records x (3+1 lanes) x 16 B", 16, 1);
        run<3>("D 21 records x 3 lanes x 16 B", 21, 1);
    }
    return 0;
}
