// EXPERIMENT (round 3, measured and NOT kept; not part of the build): the spread on the fp64 matrix pipe.
// Paste between the register-accumulating spread and k_spread_atomic of pse_amd/csrc/pse_farfield.hip to rebuild it.
//
// Result on MI355X at the metric point (N = 1e6, 256^3, P = 6; rocprofv3, gpurun_out/spread2 of round 3):
//   k_spread_mfma<6>     0.778 ms   3.04 M MFMA, 139.7 M VALU, 20.1 M LDS instructions per launch, matrix pipe busy 11 %
//   k_spread_tiles<6,16> 0.44  ms   121.8 M VALU (profiled in the same run)
// Why: v_mfma_f64_16x16x4_f64 issues every 64 cycles = 16 multiply-adds per clock and SIMD, exactly the v_fma_f64 rate, and it
// shares the fp64 units (tools/microbench/mfma_f64.hip: a partner wave's v_fma_f64 drops to a seventh beside it) -- so the pipe
// offers no extra arithmetic, only issue slots.  The slots it frees (18 FMAs per particle-block pair) are spent again on
// operands: 17 vector instructions per MFMA in the loop (list entry, origin decode, three clamped table look-ups, two products)
// and ~450 per chunk and wave on clipping, six exponentials per candidate and the per-quarter survivor lists.  Every lane = node
// formulation tried in rounds 1-3 lands at 30-40 vector instructions per (particle, block) visit and 3.5 visits per particle.

// ---- spread on the fp64 matrix pipe ---------------------------------------------------------------------------------------
// The spread of a block of nodes is a matrix product: G[(x, y), (z, c)] = sum_p Wxy[(x, y), p] Wzf[p, (z, c)] with
// Wxy = ax_p[x - ox_p] ay_p[y - oy_p] and Wzf = az_p[z - oz_p] F_p[c] (zero outside the support).  v_mfma_f64_16x16x4_f64 takes
// sixteen (x, y) nodes (a 4 x 4 patch) times sixteen (z, c) pairs (4 z times 3 components + an idle column) times four
// particles per instruction: 1024 multiply-adds in 64 cycles per SIMD -- the rate of v_fma_f64, whose units it shares
// (tools/microbench/mfma_f64.hip: a partner wave's v_fma_f64 drops to a seventh beside it) -- but they cost ONE issue slot, the z
// extent needs no compile-time offset classes, and the vector pipe is free for the operands of the next one.
//
// A workgroup owns 8 x 8 x 16 nodes; wave w owns the 4 x 4 patch (w & 1, w >> 1) of every z and keeps its sixteen (z, c)
// columns of each of the four z quarters in four accumulators.  Per chunk of 256 candidate records (lane = candidate): clip
// against the block, compact the survivors (in candidate order: the result does not depend on timing), rebuild their separable
// weights (six exponentials) into zero-padded LDS rows.  Then every wave lists, per z quarter, the survivors whose support
// reaches its sub-block, pads the lists to fours with an all-zero row, and feeds them to the matrix pipe four at a time:
// per instruction two weight look-ups and a product for A (lane = node x particle), two and a product for B.
typedef double d4v __attribute__((ext_vector_type(4)));

template <int P, bool SHEAR>
__global__ void __launch_bounds__(256, 2)   // two waves per SIMD: <= 256 registers, so the accumulators stay in VGPRs (with one wave
k_spread_mfma(                              // per SIMD allowed hipcc puts every MFMA result in AGPRs and copies it back and forth)
const FarRec *__restrict__ rec, FarBins fb, double *__restrict__ gx, double *__restrict__ gy,
              double *__restrict__ gz, DGrid G, GaussConsts gc, FastDiv dz, FastDiv dy) {
    constexpr int TX = 8, TY = 8, TZ = 16, NT = 256, CAP = 256, P1 = P + 1, TS = 3 * P1 + 4, RMAX = 20, NQ = TZ / 4;
    __shared__ __attribute__((aligned(16))) double s_tab[(CAP + 1) * TS];   // row: ax[P], 0, ay[P], 0, az[P], 0, F[3], 0; row CAP: zeros
    __shared__ int s_meta[CAP + 1];                                            // support origin relative to the block, packed
    __shared__ unsigned s_list[4][NQ][CAP + 12];                               // survivor | origin << 9 (one look-up per operand pair)
    __shared__ int s_rb[RMAX], s_ro[RMAX + 1], s_wcnt[4];
    __shared__ double s_k[SHEAR ? 64 : 1];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int tz_, ty_;
    const int tx_ = fdiv(fdiv(xcd_block(blockIdx.x, gridDim.x), dz, tz_), dy, ty_);
    const int t0[3] = {G.x0 + tx_ * TX, ty_ * TY, tz_ * TZ};
    const int ext[3] = {min(TX, G.x0 + G.nxl - t0[0]), min(TY, G.Ny - t0[1]), min(TZ, G.Nz - t0[2])};
    const int Nn[3] = {G.Nx, G.Ny, G.Nz};
    const int nb[3] = {fb.nbx, fb.nby, fb.nbz};
    if (SHEAR && tid < 64) s_k[tid] = exp_lean(gc.lnk * (double)((tid >> 3) * (tid & 7)));   // K[t][v], t, v < 8
    for (int e = tid; e < TS; e += NT) s_tab[CAP * TS + e] = 0.0;
    if (tid == 0) s_meta[CAP] = 0x080808;
    // bins whose origins [t0 - P + 1, t0 + ext - 1] (cyclic) can reach the block; consecutive z bins are one record range
    int blo[3], bcnt[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        blo[a] = wrapi(t0[a] - (P - 1), Nn[a]) / BIN;
        bcnt[a] = min(nb[a], wrapi((t0[a] + ext[a] - 1) / BIN - blo[a], nb[a]) + 1);
    }
    const int zparts = blo[2] + bcnt[2] > nb[2] ? 2 : 1;
    const int nr = bcnt[0] * bcnt[1] * zparts;
    if (tid < nr) {
        const int zp = tid % zparts, r = tid / zparts, iy = r % bcnt[1], ix = r / bcnt[1];
        const int row = (((blo[0] + ix) % nb[0]) * nb[1] + (blo[1] + iy) % nb[1]) * nb[2];
        int z0 = blo[2], z1 = blo[2] + bcnt[2];
        if (zparts == 2) { if (zp == 0) z1 = nb[2]; else { z0 = 0; z1 = blo[2] + bcnt[2] - nb[2]; } }
        const int o = fb.off[row + z0];
        s_rb[tid] = o;
        s_ro[tid + 1] = fb.off[row + z1] - o;
    }
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        s_ro[0] = 0;
        for (int r = 0; r < nr; ++r) { run += s_ro[r + 1]; s_ro[r + 1] = run; }
    }
    __syncthreads();
    const int total = s_ro[nr];

    d4v acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = d4v{0.0, 0.0, 0.0, 0.0};
    // this lane's node of the wave's patch (operand A: row = lane & 15, particle = lane >> 4) and its (z, c) column (operand B)
    const int xn = 4 * (wv & 1) + (lane & 3), yn = 4 * (wv >> 1) + ((lane >> 2) & 3);
    const int zz = (lane & 15) >> 2, cc = lane & 3, kslot = lane >> 4;
    const unsigned long long below = (1ull << lane) - 1ull;

    for (int c0 = 0; c0 < total; c0 += NT) {
        const int k = c0 + tid;
        const bool valid = k < total;
        int r = 0;
        for (int q = 1; q < nr; ++q) r = s_ro[q] <= k ? q : r;
        const size_t slot = valid ? (size_t)(s_rb[r] + (k - s_ro[r])) : (size_t)s_rb[0];
        const double2 *rp = reinterpret_cast<const double2 *>(rec + slot);
        const double2 q0 = rp[0], q1 = rp[1], q2 = rp[2], q3 = rp[3];
        const int4 hd = *reinterpret_cast<const int4 *>(&q0);
        const int o[3] = {hd.x, hd.y, hd.z};
        int rel[3];
        bool hit = valid;
#pragma unroll
        for (int a = 0; a < 3; ++a) {   // support origin relative to the block (nearest image)
            int d = o[a] - t0[a];
            if (d < -Nn[a] / 2) d += Nn[a]; else if (d >= Nn[a] - Nn[a] / 2) d -= Nn[a];
            rel[a] = d;
            hit = hit && d + P > 0 && d < ext[a];
        }
        const unsigned long long mask = __ballot(hit);
        if (lane == 0) s_wcnt[wv] = __popcll(mask);
        __syncthreads();                       // the counts are in; every wave is through the previous chunk's tables and lists
        const int w0 = s_wcnt[0], w1 = s_wcnt[1], w2 = s_wcnt[2], w3 = s_wcnt[3];
        const int nsurv = w0 + w1 + w2 + w3;
        if (nsurv == 0) continue;              // uniform over the workgroup
        const int sidx = (wv > 0 ? w0 : 0) + (wv > 1 ? w1 : 0) + (wv > 2 ? w2 : 0) + __popcll(mask & below);
        if (mask) {   // the survivors' separable weights (six exponentials per particle)
            const double c = G.expfac;
            const double Y0 = G.hy * q1.y, Z0 = G.hz * q2.x, u = G.hx * q1.x + gc.s * Y0;
            double *row = s_tab + sidx * TS;
            double a[P];
            gauss_axis<P>(-c * u * u, -2.0 * c * G.hx * u, gc.rx, a);
            if (hit) {
#pragma unroll
                for (int t = 0; t < P; ++t) row[t] = a[t];
                row[P] = 0.0;
            }
            gauss_axis<P>(-c * Y0 * Y0, -2.0 * c * G.hy * (Y0 + gc.s * u), gc.ry, a);
            if (hit) {
#pragma unroll
                for (int t = 0; t < P; ++t) row[P1 + t] = a[t];
                row[P1 + P] = 0.0;
            }
            gauss_axis<P>(-c * Z0 * Z0, -2.0 * c * G.hz * Z0, gc.rz, a);
            if (hit) {
#pragma unroll
                for (int t = 0; t < P; ++t) row[2 * P1 + t] = a[t];
                row[2 * P1 + P] = 0.0;
                row[3 * P1] = q2.y; row[3 * P1 + 1] = q3.x; row[3 * P1 + 2] = q3.y; row[3 * P1 + 3] = 0.0;
                s_meta[sidx] = (rel[0] + 8) | ((rel[1] + 8) << 8) | ((rel[2] + 8) << 16);
            }
        }
        __syncthreads();                       // tables complete
        // the wave's lists: survivors that reach its 4 x 4 patch, per z quarter
        int cnt[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) cnt[q] = 0;
        const int x0w = 4 * (wv & 1), y0w = 4 * (wv >> 1);
        for (int s0 = 0; s0 < nsurv; s0 += 64) {
            const int sv = s0 + lane;
            const int m = s_meta[sv < nsurv ? sv : CAP];
            const int ox = (m & 255) - 8, oy = ((m >> 8) & 255) - 8, oz = ((m >> 16) & 255) - 8;
            const bool txy = sv < nsurv && ox <= x0w + 3 && ox + P - 1 >= x0w && oy <= y0w + 3 && oy + P - 1 >= y0w;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const bool t = txy && oz <= 4 * q + 3 && oz + P - 1 >= 4 * q;
                const unsigned long long mq = __ballot(t);
                if (t) s_list[wv][q][cnt[q] + __popcll(mq & below)] = (unsigned)sv | ((unsigned)m << 9);
                cnt[q] += __popcll(mq);
            }
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            if (lane < 11) s_list[wv][q][cnt[q] + lane] = (unsigned)CAP | (0x080808u << 9);   // zero rows: whole fours + two fours of look-ahead
        __builtin_amdgcn_wave_barrier();
        // The matrix pipe: four survivors per instruction.  Software-pipelined by hand -- the list entry of group g + 2 and the four
        // table values of group g + 1 are requested before the products of group g are formed, so no iteration waits for an LDS
        // round trip (left to the compiler every instruction waited for list -> table -> product: 46 % of the matrix rate).
        const int xn8 = xn + 8, yn8 = yn + 8;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int n4 = (cnt[q] + 3) >> 2;
            const unsigned *lst = &s_list[wv][q][kslot];
            const int zq8 = 4 * q + zz + 8;
            auto look = [&](unsigned e, double &ax, double &ay, double &az, double &fc, double &kf) __attribute__((always_inline)) {
                const unsigned sv = e & 511u;
                const unsigned tx = min((unsigned)(xn8 - (int)((e >> 9) & 255u)), (unsigned)P);
                const unsigned ty = min((unsigned)(yn8 - (int)((e >> 17) & 255u)), (unsigned)P);
                const unsigned tz = min((unsigned)(zq8 - (int)((e >> 25) & 127u)), (unsigned)P);
                const double *row = s_tab + sv * TS;
                ax = row[tx]; ay = row[P1 + ty]; az = row[2 * P1 + tz]; fc = row[3 * P1 + cc];
                kf = SHEAR ? s_k[(tx & 7) * 8 + (ty & 7)] : 1.0;   // K[tx][ty]; outside the support ax ay = 0 and any finite value will do
            };
            double ax, ay, az, fc, kf;
            look(lst[0], ax, ay, az, fc, kf);
            unsigned en = lst[4];
            for (int g = 0; g < n4; ++g) {
                double A = ax * ay;
                if (SHEAR) A *= kf;
                const double B = az * fc;
                look(en, ax, ay, az, fc, kf);
                en = lst[4 * g + 8];
                acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(A, B, acc[q], 0, 0, 0);
            }
        }
    }
    // lane holds, of quarter q and register r: node x = patch x0 + (lane >> 4), y = patch y0 + r, z = 4 q + zz, component cc
    const int xs = 4 * (wv & 1) + (lane >> 4);
    if (cc < 3 && xs < ext[0]) {
        double *g = cc == 0 ? gx : (cc == 1 ? gy : gz);
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int r2 = 0; r2 < 4; ++r2) {
                const int ys = 4 * (wv >> 1) + r2, zs = 4 * q + zz;
                if (ys < ext[1] && zs < ext[2])
                    g[((size_t)(t0[0] - G.x0 + G.hl + xs) * G.Ny + (t0[1] + ys)) * G.Nz + t0[2] + zs] = acc[q][r2];
            }
    }
}


template <int P>
static void launch_spread_mfma(const FarRec *rec, FarBins fb, double *gx, double *gy, double *gz, const DGrid &G, const GaussConsts &gc,
                               hipStream_t s) {
    const int ntx = (G.nxl + 7) / 8, nty = (G.Ny + 7) / 8, ntz = (G.Nz + 15) / 16;
    const dim3 g(ntx * nty * ntz), b(256);
    const FastDiv dz = fast_div(ntz), dy = fast_div(nty);
    if (gc.s != 0.0) hipLaunchKernelGGL((k_spread_mfma<P, true>), g, b, 0, s, rec, fb, gx, gy, gz, G, gc, dz, dy);
    else hipLaunchKernelGGL((k_spread_mfma<P, false>), g, b, 0, s, rec, fb, gx, gy, gz, G, gc, dz, dy);
}

