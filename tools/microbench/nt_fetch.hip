// Calibration of FETCH_SIZE on gfx950 for the access shapes of the pair-list mat-vec (VERDICT r2 "weak" 3): is the x2 the guide
// prescribes for wide coalesced reads also right for NON-TEMPORAL 16-byte loads, and for a wave whose lanes stop at
// different slots (the ELL list padded to the wave's longest row)?  Each kernel reads a known number of bytes / 128-byte
// lines of a 640 MB buffer (larger than the Infinity Cache) exactly once; run under
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -- tools/microbench/nt_fetch
// and compare FETCH_SIZE (KB) per kernel with the printed byte counts.  Also prints each kernel's time -> GB/s.
// hipcc --offload-arch=gfx950 -O3 tools/microbench/nt_fetch.hip -o tools/microbench/nt_fetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <vector>
typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef double d2v __attribute__((ext_vector_type(2)));
constexpr size_t GROUP = 5120;   // [64 lanes][4 x u32] + [4 slots][64 lanes](f, h)

template <bool NT>
__global__ void __launch_bounds__(256) k_stream_x4(const u4v *__restrict__ p, size_t n, unsigned *out) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const u4v v = NT ? __builtin_nontemporal_load(p + i) : p[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
template <bool NT>
__global__ void __launch_bounds__(256) k_stream_x2(const double *__restrict__ p, size_t n, double *out) {
    double acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        acc += NT ? __builtin_nontemporal_load(p + i) : p[i];
    if (acc == 0.12345) out[0] = acc;
}
// the mat-vec's list stream: one wave per block of 64 rows, cap/4 groups; lane l reads groups while 4 g < cnt[row]
template <bool NT>
__global__ void __launch_bounds__(256) k_list(const char *__restrict__ data, const int *__restrict__ cnt, int rows, int cap, double *out) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    const int lane = row & 63, c = cnt[row];
    const char *rec = data + (size_t)(row >> 6) * (cap / 4) * GROUP;
    double acc = 0;
    for (int s0 = 0; s0 < c; s0 += 4) {
        const char *grp = rec + (size_t)(s0 >> 2) * GROUP;
        const u4v e = NT ? __builtin_nontemporal_load((const u4v *)grp + lane) : ((const u4v *)grp)[lane];
        acc += (double)(e.x ^ e.y ^ e.z ^ e.w);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const d2v fh = NT ? __builtin_nontemporal_load((const d2v *)(grp + 1024) + u * 64 + lane) : ((const d2v *)(grp + 1024))[u * 64 + lane];
            acc += fh.x + fh.y;
        }
    }
    if (acc == 0.12345) out[0] = acc;
}

static float timed(const char *name, double bytes, void (*launch)(void)) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s bytes %.1f MB  %.3f ms  %.2f TB/s\n", name, bytes / 1e6, ms, bytes / (ms * 1e-3) / 1e12);
    return ms;
}

static char *g_buf; static int *g_cnt; static double *g_out; static size_t g_bytes; static int g_rows, g_cap;
int main() {
    g_bytes = (size_t)1 << 30;
    (void)hipMalloc(&g_buf, g_bytes); (void)hipMemset(g_buf, 1, g_bytes);
    (void)hipMalloc(&g_out, 64);
    g_rows = 1000000; g_cap = 48;
    // neighbour counts like the metric point: Poisson(21.3), as a sum of uniforms is not it -- draw by inversion
    std::vector<int> cnt(g_rows);
    srand(7);
    double lines_fh = 0, lines_e = 0, bytes_lane = 0, bytes_wave_max = 0;
    for (int i = 0; i < g_rows; ++i) {
        const double L = 21.3; double p = 1.0; int kk = 0; const double lim = exp(-L);
        do { ++kk; p *= (rand() + 1.0) / (RAND_MAX + 2.0); } while (p > lim);
        cnt[i] = kk - 1 > g_cap ? g_cap : kk - 1;
    }
    for (int w = 0; w < g_rows / 64; ++w) {
        int wmax = 0;
        for (int l8 = 0; l8 < 8; ++l8) {          // a 128-byte line of (f, h) or of entries = 8 consecutive lanes
            int m = 0;
            for (int l = 0; l < 8; ++l) { const int c = cnt[w * 64 + l8 * 8 + l]; m = c > m ? c : m; bytes_lane += 20.0 * ((c + 3) / 4 * 4); }
            lines_fh += (m + 3) / 4 * 4;           // one line per slot the longest of the eight lanes reaches (whole groups)
            lines_e += (m + 3) / 4;                // one line of entries per group
            wmax = m > wmax ? m : wmax;
        }
        bytes_wave_max += (double)((wmax + 3) / 4) * GROUP;
    }
    (void)hipMalloc(&g_cnt, g_rows * sizeof(int));
    (void)hipMemcpy(g_cnt, cnt.data(), g_rows * sizeof(int), hipMemcpyHostToDevice);
    printf("list: %d rows, cap %d: lane-exact %.1f MB, by 128-B lines touched %.1f MB, padded to the wave maximum %.1f MB\n", g_rows, g_cap,
           bytes_lane / 1e6, (lines_fh + lines_e) * 128 / 1e6, bytes_wave_max / 1e6);
    timed("k_stream_x4<nt>", (double)g_bytes, [] { k_stream_x4<true><<<2048, 256>>>((const u4v *)g_buf, g_bytes / 16, (unsigned *)g_out); });
    timed("k_stream_x4<plain>", (double)g_bytes, [] { k_stream_x4<false><<<2048, 256>>>((const u4v *)g_buf, g_bytes / 16, (unsigned *)g_out); });
    timed("k_stream_x2<nt>", (double)g_bytes, [] { k_stream_x2<true><<<2048, 256>>>((const double *)g_buf, g_bytes / 8, g_out); });
    timed("k_stream_x2<plain>", (double)g_bytes, [] { k_stream_x2<false><<<2048, 256>>>((const double *)g_buf, g_bytes / 8, g_out); });
    timed("k_list<nt> (ragged rows)", (lines_fh + lines_e) * 128, [] { k_list<true><<<(g_rows + 255) / 256, 256>>>(g_buf, g_cnt, g_rows, g_cap, g_out); });
    timed("k_list<plain> (ragged rows)", (lines_fh + lines_e) * 128, [] { k_list<false><<<(g_rows + 255) / 256, 256>>>(g_buf, g_cnt, g_rows, g_cap, g_out); });
    return 0;
}
