// fp64 matrix pipe of gfx950: issue rate of v_mfma_f64_16x16x4_f64 and v_mfma_f64_4x4x4_4b_f64 (the guide gives the layout,
// not the rate), of v_fma_f64 for comparison, and what the two cost each other on one SIMD (separate waves / one wave).
// Also checks the operand and result lane maps of the 16x16x4 form on exact integer data (A = asymmetric).
// hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_f64.hip -o tools/microbench/mfma_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int ITER = 4096;

// MODE 0: NACC independent 16x16x4 chains; 1: 4x4x4_4b chains; 2: v_fma_f64 chains (NACC x 4 independent);
// 3: waves 0-3 MFMA, waves 4-7 VALU fma (the SIMD partners; launch 512 threads); 4: one wave interleaves one MFMA with NV fmas
template <int MODE, int NACC, int NV>
__global__ void __launch_bounds__(512) k(double *out, double a0, double b0, long long *cyc) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double a = a0 + lane * 1e-9, b = b0 - lane * 1e-9;
    d4 acc[NACC];
    double s[NACC];
    double v[16];
#pragma unroll
    for (int q = 0; q < NACC; ++q) { acc[q] = d4{0, 0, 0, 0}; s[q] = 0.0; }
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = q * 1e-3;
    const bool mf = MODE == 0 || MODE == 1 || MODE == 4 || (MODE == 3 && wv < 4);
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
        if (MODE == 0 || MODE == 4 || (MODE == 3 && mf)) {
#pragma unroll
            for (int q = 0; q < NACC; ++q) {
                acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
                if (MODE == 4) {
#pragma unroll
                    for (int e = 0; e < NV; ++e) v[e & 15] = fma(v[e & 15], a, b);
                }
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int q = 0; q < NACC; ++q) s[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s[q], 0, 0, 0);
        } else {
#pragma unroll
            for (int q = 0; q < NACC; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[(q * 4 + e) & 15] = fma(v[(q * 4 + e) & 15], a, b);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    double r = 0;
#pragma unroll
    for (int q = 0; q < NACC; ++q) r += acc[q].x + acc[q].y + acc[q].z + acc[q].w + s[q];
#pragma unroll
    for (int q = 0; q < 16; ++q) r += v[q];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (lane == 0) cyc[(size_t)blockIdx.x * (blockDim.x >> 6) + wv] = t1 - t0;
}

template <int MODE, int NACC, int NV>
static void run(const char *name, int threads, double ops_per_iter_per_wave, const char *unit) {
    const int blocks = 256 * (threads == 256 ? 1 : 1);
    double *out; long long *cyc;
    (void)hipMalloc(&out, (size_t)blocks * threads * sizeof(double));
    (void)hipMalloc(&cyc, (size_t)blocks * (threads / 64) * sizeof(long long));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE, NACC, NV><<<blocks, threads>>>(out, 1.0, 0.5, cyc); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) k<MODE, NACC, NV><<<blocks, threads>>>(out, 1.0, 0.5, cyc);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    std::vector<long long> c((size_t)blocks * (threads / 64));
    (void)hipMemcpy(c.data(), cyc, c.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double lo_w = 0, hi_w = 0; int nlo = 0, nhi = 0;
    for (size_t i = 0; i < c.size(); ++i) { if ((int)(i % (threads / 64)) < 4) { lo_w += c[i]; ++nlo; } else { hi_w += c[i]; ++nhi; } }
    // s_memtime ticks at a fixed 100 MHz-derived rate? (guide: tick = shader cycle) -- print both per-op ticks and wall
    printf("%-44s %4d thr  %.3f ms  ticks/op: waves0-3 %.1f", name, threads, ms, lo_w / nlo / (ITER * ops_per_iter_per_wave));
    if (nhi) printf("  waves4-7 %.1f", hi_w / nhi / (ITER * ops_per_iter_per_wave));
    printf("   wall ns/op/wave %.2f (%s)\n", ms * 1e6 / (ITER * ops_per_iter_per_wave), unit);
    (void)hipFree(out); (void)hipFree(cyc);
}

// lane maps: D = A B with A[i][k] = 1 + i + 16 k (asymmetric), B[k][j] = 1 + 3 j + 100 k
__global__ void kmap(double *D) {
    const int l = threadIdx.x;
    const double a = 1 + (l & 15) + 16 * (l >> 4);            // A[i = l & 15][k = l >> 4]
    const double b = 1 + 3 * (l & 15) + 100 * (l >> 4);       // B[k = l >> 4][j = l & 15]
    d4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];   // row = (lane >> 4) + 4 reg, col = lane & 15
}

int main() {
    double *D; (void)hipMalloc(&D, 256 * sizeof(double));
    kmap<<<1, 64>>>(D);
    double h[256]; (void)hipMemcpy(h, D, sizeof h, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double ref = 0; for (int kk = 0; kk < 4; ++kk) ref += (1 + i + 16 * kk) * (1.0 + 3 * j + 100 * kk);
        if (h[i * 16 + j] != ref) ++bad;
    }
    printf("lane map check (A[l&15][l>>4], B[l>>4][l&15], D row=(l>>4)+4r col=l&15): %s (%d wrong)\n", bad ? "WRONG" : "ok", bad);

    run<0, 1, 0>("mfma_f64_16x16x4, 1 chain, 1 wave/SIMD", 256, 1, "MFMA");
    run<0, 2, 0>("mfma_f64_16x16x4, 2 chains, 1 wave/SIMD", 256, 2, "MFMA");
    run<0, 4, 0>("mfma_f64_16x16x4, 4 chains, 1 wave/SIMD", 256, 4, "MFMA");
    run<0, 4, 0>("mfma_f64_16x16x4, 4 chains, 2 waves/SIMD", 512, 4, "MFMA");
    run<1, 4, 0>("mfma_f64_4x4x4_4b, 4 chains, 1 wave/SIMD", 256, 4, "MFMA");
    run<1, 8, 0>("mfma_f64_4x4x4_4b, 8 chains, 1 wave/SIMD", 256, 8, "MFMA");
    run<2, 4, 0>("v_fma_f64, 16 chains, 1 wave/SIMD", 256, 16, "FMA");
    run<2, 4, 0>("v_fma_f64, 16 chains, 2 waves/SIMD", 512, 16, "FMA");
    run<3, 4, 0>("partners: waves0-3 MFMA(4) | waves4-7 16 fma", 512, 4, "MFMA|4 FMA");
    run<4, 4, 4>("one wave: per MFMA 4 v_fma_f64", 256, 4, "MFMA+4FMA");
    run<4, 4, 8>("one wave: per MFMA 8 v_fma_f64", 256, 4, "MFMA+8FMA");
    run<4, 4, 16>("one wave: per MFMA 16 v_fma_f64", 256, 4, "MFMA+16FMA");
    run<4, 4, 16>("two waves/SIMD: per MFMA 16 v_fma_f64", 512, 4, "MFMA+16FMA");
    return 0;
}
