// Measured and retired (round 3): the pair-list mat-vec with the list stream on the LDS-DMA path (global_load_lds_dwordx4).
// Every wave requests all of its list groups into LDS before anything else, so no registers are held while the bytes travel.
// Parity-green (tests/test_gpu_parity.py -k "lanczos or brownian or overflow"); 0.192 ms with two groups per wave and 0.231 ms
// with three against 0.155 ms for k_mreal_list<true, 4, 256, true, 4> in the same box: the kernel still needs 124 VGPRs for the
// gathers (four waves per SIMD) and the 41 / 61 KB of LDS leave three / two workgroups per CU instead of four -- fewer waves
// for the gather phase cost more than the earlier list bytes buy.  Not compiled into the library; kept for the record.
// (Fragment of pse_amd/csrc/pse_kernels.hip at the commit that removed it; needs that file's helpers.)

// The Lanczos mat-vec with the LIST STREAM ON THE LDS-DMA PATH.  k_mreal_list holds a group's 80 bytes per lane in registers from
// the load to the use, so what a CU has in flight is bounded by its registers (124 VGPRs: four waves per SIMD, 80 KB of list per
// CU -- by Little's law just what 5.4 TB/s at ~4 us need).  Here every wave requests ALL of its groups (up
// to NG; the 5120-byte group is contiguous: five 1024-byte wave loads) straight into LDS with global_load_lds_dwordx4 before it
// does anything else: no registers are held while the bytes travel, and the groups beyond the first are already there when the
// gathers of the first retire.  Four waves per block of 64 rows, each taking every fourth group, as in k_mreal_list<.., 4>.
template <int NG>
__global__ void __launch_bounds__(256)
k_mreal_list_dma(const double4 *__restrict__ pos_s, const double4 *__restrict__ vec_s, double4 *__restrict__ out_s, int lo, int hi,
                 DBox box, int shift_only, double self, NbList nb, LzFuse lz, const double2 *__restrict__ pv,
                 const int *__restrict__ cell_off, DCells nc, double rcut2, const double *__restrict__ coef, VerletList vl) {
    constexpr int WSP = 4, UNROLL = 4;
    constexpr size_t GRP = 4 * NB_REC;
    __shared__ double shift[27 * 3];
    __shared__ double red[(WSP - 1) * 3 * 64];
    extern __shared__ __attribute__((aligned(16))) char lbuf[];   // [WSP][NG][GRP]
    if (threadIdx.x < 27) {
        double sx, sy, sz;
        image_shift(threadIdx.x, box, sx, sy, sz);
        shift[threadIdx.x * 3] = sx; shift[threadIdx.x * 3 + 1] = sy; shift[threadIdx.x * 3 + 2] = sz;
    }
    const int wv = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
    const int blk = xcd_block(blockIdx.x, gridDim.x);
    const int i = lo + blk * 64 + lane;
    const bool active = i < hi;
    const int cnt = active ? nb.cnt[i] : 0;
    const int c = max(cnt, 0);
    int wmax = c;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wmax = max(wmax, __shfl_xor(wmax, o, 64));
    const char *rec = nb.data + (size_t)blk * nb.cap * NB_REC;
    char *mybuf = lbuf + (size_t)(wv * NG) * GRP;
#pragma unroll
    for (int k = 0; k < NG; ++k) {
        const int s0 = UNROLL * (wv + WSP * k);
        if (s0 < wmax) {                                           // wave-uniform
            const char *grp = rec + (size_t)(s0 >> 2) * GRP;
            char *dst = mybuf + (size_t)k * GRP;
            if (s0 < c) __builtin_amdgcn_global_load_lds((glb_void_t *)(grp + lane * 16), (lds_void_t *)dst, 16, 0, 0);
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
                if (s0 + u < c)
                    __builtin_amdgcn_global_load_lds((glb_void_t *)(grp + 1024 + u * 1024 + lane * 16), (lds_void_t *)(dst + 1024 + u * 1024), 16, 0, 0);
        }
    }
    double4 pi = make_double4(0.0, 0.0, 0.0, 0.0);
    if (active) pi = pos_s[i];
    __syncthreads();                                               // the shift table; the barrier also waits for this wave's loads
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    double ux = 0.0, uy = 0.0, uz = 0.0;
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    typedef double d2v __attribute__((ext_vector_type(2)));
    for (int k = 0; UNROLL * (wv + WSP * k) < wmax; ++k) {         // wave-uniform trip count
        const int s0 = UNROLL * (wv + WSP * k);
        if (s0 < c) {
            unsigned e[UNROLL];
            double f[UNROLL], h[UNROLL];
            if (k < NG) {
                const char *src = mybuf + (size_t)k * GRP;
                const u4v e4 = *(const u4v *)(src + lane * 16);
                e[0] = e4.x; e[1] = e4.y; e[2] = e4.z; e[3] = e4.w;
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    const d2v fh = *(const d2v *)(src + 1024 + u * 1024 + lane * 16);
                    f[u] = fh.x; h[u] = fh.y;
                }
            } else {                                               // rows longer than NG groups per wave: from memory, as k_mreal_list does
                const char *grp = rec + (size_t)(s0 >> 2) * GRP;
                const u4v e4 = __builtin_nontemporal_load((const u4v *)grp + lane);
                e[0] = e4.x; e[1] = e4.y; e[2] = e4.z; e[3] = e4.w;
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    const d2v fh = __builtin_nontemporal_load((const d2v *)(grp + 1024) + u * 64 + lane);
                    f[u] = fh.x; h[u] = fh.y;
                }
            }
#pragma unroll
            for (int u = 1; u < UNROLL; ++u) if (s0 + u >= c) e[u] = e[0];    // slots past the row's count were never written
            double2 ra[UNROLL], rb[UNROLL], rc[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const double2 *r = pv + 3 * (size_t)(e[u] & JMASK);
                ra[u] = r[0]; rb[u] = r[1]; rc[u] = r[2];
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const unsigned code = e[u] >> 27;
                const bool ok = s0 + u < c;
                double dx = pi.x - ra[u].x - shift[code * 3], dy = pi.y - ra[u].y - shift[code * 3 + 1], dz = pi.z - rb[u].x - shift[code * 3 + 2];
                if (!shift_only) min_image(box, dx, dy, dz);
                const double fu = ok ? f[u] : 0.0, hu = ok ? h[u] : 0.0;
                const double Fx = rb[u].y, Fy = rc[u].x, Fz = rc[u].y;
                const double rdF = (dx * Fx + dy * Fy + dz * Fz) * hu;
                ux += fu * Fx + rdF * dx;
                uy += fu * Fy + rdF * dy;
                uz += fu * Fz + rdF * dz;
            }
        }
    }
    if (active && cnt < 0 && wv == 0) mreal_row_fallback(i, pos_s, vec_s, box, cell_off, nc, rcut2, coef, vl, ux, uy, uz);
    if (wv > 0) { red[((wv - 1) * 3 + 0) * 64 + lane] = ux; red[((wv - 1) * 3 + 1) * 64 + lane] = uy; red[((wv - 1) * 3 + 2) * 64 + lane] = uz; }
    __syncthreads();
    if (wv > 0) return;
    for (int w = 0; w < WSP - 1; ++w) { ux += red[(w * 3 + 0) * 64 + lane]; uy += red[(w * 3 + 1) * 64 + lane]; uz += red[(w * 3 + 2) * 64 + lane]; }
    double a = 0.0, b = 0.0, cc = 0.0;
    if (active) {
        const double4 vi = vec_s[i];
        ux = fma(self, vi.x, ux); uy = fma(self, vi.y, uy); uz = fma(self, vi.z, uz);
        a = vi.x * vi.x + vi.y * vi.y + vi.z * vi.z;
        b = vi.x * ux + vi.y * uy + vi.z * uz;
        if (lz.vprev) {
            const double4 m = lz.vprev[i];
            cc = vi.x * m.x + vi.y * m.y + vi.z * m.z;
        }
    }
    a = wave_sum(a); b = wave_sum(b); cc = wave_sum(cc);
    if (threadIdx.x == 0) {
        lz.partials[blockIdx.x] = a; lz.partials[lz.npart_cap + blockIdx.x] = b; lz.partials[2 * lz.npart_cap + blockIdx.x] = cc;
    }
    if (active) out_s[i] = make_double4(ux, uy, uz, 0.0);
}

