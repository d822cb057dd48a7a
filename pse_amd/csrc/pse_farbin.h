// Far-field binning shared by the gather-into-cell-order pass (k_permute, pse_kernels.hip) and the far-field kernels
// (pse_farfield.hip): the support of a particle (PSEv1/Mobility.cu:173-219) and its rank inside the 8^3 block of nodes ("bin")
// its support origin lies in.  Round 4: the ranking is done by the pass that already holds the sorted position -- the separate
// k_support launch and the 48 bytes per particle it parked for k_far_records are gone (the records recompute the support).
#pragma once
#include "pse_kernels.h"

namespace pse {

constexpr int BIN = 8;
__host__ __device__ inline int bins_of(int n) { return (n + BIN - 1) / BIN; }
__device__ __forceinline__ int bin_index(int ox, int oy, int oz, const FarBins &fb) {
    return ((ox / BIN) * fb.nby + (oy / BIN)) * fb.nbz + (oz / BIN);
}

// first node index per axis (unwrapped) and the offset of that node from the particle in grid units
__device__ __forceinline__ void support_start(double f, int n, int P, int &start, double &delta0) {
    const double s = f * n;
    const int i0 = (int)s;
    start = i0 - P / 2 + 1 - ((P & 1) && (s - i0 < 0.5) ? 1 : 0);
    delta0 = start - s;
}

// support origin (wrapped into the grid; .w = the node plane the particle sits in: decides which slab owns it) and the offset of
// the origin from the particle, from fractional coordinates
__device__ __forceinline__ void far_support(double fx, double fy, double fz, const DGrid &G, int4 &o, double4 &d) {
    support_start(fx, G.Nx, G.P, o.x, d.x);   // PSEv1/Mobility.cu:212-214
    support_start(fy, G.Ny, G.P, o.y, d.y);
    support_start(fz, G.Nz, G.P, o.z, d.z);
    o.x = wrapi(o.x, G.Nx); o.y = wrapi(o.y, G.Ny); o.z = wrapi(o.z, G.Nz);
    o.w = min((int)(fx * G.Nx), G.Nx - 1);
    d.w = 0.0;
}

// Rank of this lane's particle inside its bin (-1: `need` false -- a slab rank never touches the particle).  Called by whole
// wavefronts (lanes without a particle pass need = false).  Neighbouring lanes are neighbouring particles of the cell order and
// mostly share a bin: one atomic per distinct bin of the wave (the leader adds the group's size, members take consecutive ranks)
// instead of 64 same-address atomics.
__device__ __forceinline__ int far_bin_rank(bool need, int bin, int *__restrict__ cnt) {
    const int lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
    unsigned long long todo = __ballot(need);
    if (!need) bin = -1;
    int prefix = 0, count = 0, leader = lane;
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        const int b0 = __shfl(bin, src, 64);
        const unsigned long long m = __ballot(bin == b0) & todo;
        if (bin == b0) { prefix = __popcll(m & below); count = __popcll(m); leader = src; }
        todo &= ~m;
    }
    int base = 0;
    if (need && leader == lane) base = atomicAdd(&cnt[bin], count);
    base = __shfl(base, leader, 64);
    return need ? base + prefix : -1;
}

}  // namespace pse
