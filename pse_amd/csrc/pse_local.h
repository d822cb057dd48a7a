// Owned-particle decomposition of a team step (pse_team_step_local in include/pse_amd.h): the kernels that turn what a rank OWNS
// -- the particles of its cell slab, in the caller's arrays -- into the cell-sorted row space the near field, the far field and the
// Lanczos iteration of the rank work on, and back.  New design: the reference is single-GPU (PSEv1/Stokes.cc:104); what these
// replace is HOOMD's domain decomposition (Communicator::migrateParticles / exchangeGhosts), which the plugin never saw.
//
// Row space of a rank (capacities fixed at creation, the live counts known ON THE DEVICE only -- LocalRows in pse_kernels.h):
//     [0, c_own)                       the particles of the rank's own cell layers, in cell order, ordered by TAG inside a cell
//     [c_own, c_own + c_g)             ghosts: the `depth` cell layers below the slab (the left neighbour's last layers)
//     [c_own + c_g, c_own + 2 c_g)     ghosts: the `depth` layers above it
// Both neighbours hold the same particles in those layers and order them the same way (cells in index order, tags inside a cell), so
// row k of a ghost region IS row k of the neighbour's boundary rows: vectors are exchanged by position, in messages of the fixed
// size c_g rows -- no count ever has to reach a host.
#pragma once
#include "pse_kernels.h"

namespace pse {

struct LocalGeom {
    int rank, G, per, depth, nx;     // this rank, ranks, cell layers per rank, ghost layers per side, cell layers in all
    int c_own, c_g, c_x;             // row capacities (own, ghosts per side) and records per neighbour message of the step's first exchange
};
// cells of the three regions of the row space (own, left ghosts, right ghosts) in the storage order, the rows they start at, and the
// cells whose offsets bound the sub-ranges the step needs (DCells.xpad = 1: the last storage cell of every layer is empty, so its
// offset is the end of the layer's rows)
struct LocalRegions {
    int c0[3], c1[3], base[3], cap[3];
    int c_first_end;      // offset there = end of the first `depth` own layers
    int c_last_begin;     // offset there = begin of the last `depth` own layers
    int c_gl_adj;         // offset there = begin of the left ghosts' layer next to the slab
    int c_gr_adj;         // offset there = end of the right ghosts' layer next to the slab
};
constexpr int LOCAL_REC = 10;     // doubles per particle record of the first exchange: pos.xyzw | force.xyz, mass | (image.xyz, tag) as four 32-bit words
constexpr int LOCAL_HDR = 4;      // doubles in front of the records; on the receiving side the first holds the sender's two record counters (two ints)
enum { LOCAL_ERR_OWN = 1, LOCAL_ERR_GHOST = 2, LOCAL_ERR_MSG = 4, LOCAL_ERR_FAR = 8, LOCAL_ERR_COUNT = 16 };   // bits of the error word
// counters of a step, zeroed with the cell counts: [0] records in the left message, [1] in the right one (8-byte aligned: the pair
// travels as one double in the step's first exchange)
constexpr int LOCAL_NCOUNTER = 8;
constexpr int LOCAL_MAX_LAYERS = 512;    // cell layers along x of an owned-particle rank's grid (the per-layer counts of its sort) ...
constexpr int LOCAL_LAYER_SLOTS = 8;     // ... kept in this many copies, LOCAL_MAX_LAYERS ints apart: workgroup b adds to copy b % 8

struct LocalCaller {     // the caller's arrays of one rank (device): rows [0, *n_local) hold the particles it owns, in any order
    double4 *pos, *vel;
    double3 *accel;
    int3 *image;
    const double4 *force;
    unsigned *tag;
    unsigned *n_local;
};
struct LocalPool {       // per particle of the pool (own arrays + both incoming messages): cell, arrival rank inside it, tag
    unsigned *keys, *rank, *ptag;
    int *cnt;            // particles per storage cell
};
// (1) every particle of the caller's arrays: its cell; the particles in (or beyond) the boundary layers go into the messages for
// the neighbours (full records: the neighbour becomes the owner of those that have left the slab); what the rank keeps is counted
// (layer_cnt: kept particles per x layer in LOCAL_LAYER_SLOTS copies of LOCAL_MAX_LAYERS ints, zeroed with the cell counts -- what k_local_offsets starts the layers from)
void launch_local_classify(const LocalCaller &c, const LocalGeom &g, DBox box, DCells nc, LocalPool pool, double *send_l, double *send_r,
                           int *counters, int *err, int *layer_cnt, hipStream_t s);
// (2) the records that arrived: their cells, counted with the rest
void launch_local_bin_incoming(const double *recv_l, const double *recv_r, const LocalGeom &g, DBox box, DCells nc, LocalPool pool, int *err,
                               int *layer_cnt, hipStream_t s);
// (3) offsets of the kept cells with every region at its fixed base and the row ranges of the step (LocalRows): a workgroup per
// kept layer, no grid-wide scan; then the slot of every kept particle
void launch_local_offsets(const int *cell_cnt, const int *layer_cnt, int *cell_off, const LocalGeom &g, const LocalRegions &rg, int layer_cells,
                          LocalRows *rows, int *err, hipStream_t s);
void launch_local_scatter(const int *cell_off, const LocalGeom &g, LocalPool pool, unsigned *slots, const LocalRows *rows, hipStream_t s);
// (4) the rows themselves: ordered by tag inside their cell, gathered from the caller's arrays or from a message
struct LocalSorted {
    double4 *pos_s; float4 *posf_s; double2 *pv; double4 *f_s; unsigned *tag_s; double4 *porig_s; double *mass_s; int3 *image_s;
    double4 *psi_s;      // nullable: the particle noise of the step (K14, keyed by tag)
};
void launch_local_permute(const LocalCaller &c, const double *recv_l, const double *recv_r, const LocalGeom &g, DBox box, const int *cell_off,
                          LocalPool pool, const unsigned *slots, const LocalRows *rows, LocalSorted out, const FarBinArgs *far, uint32_t seed,
                          uint32_t timestep, const uint32_t *ts_off, hipStream_t s);
// (5) the end of the step on the own rows: Brownian part from the Lanczos basis (st: the device-side decision; null: none), sum of
// the three contributions, Euler update + wrap (K15, PSEv1/Stokes.cu:137-192), everything written to the caller's arrays in row
// order -- the caller's particles are then exactly the rank's own rows, *n_local their number
struct LocalFinish {
    const LocalRows *rows; const LzState *st;
    int *zero; int n_zero;                   // counters the NEXT step starts from zero (bin, cell and layer counts, message counters): cleared here, not by a memset in front of it
    const double4 *psi_s, *V; size_t stride; const double *scal; double scale;
    const double4 *uw_s, *ur_s;              // far field, near field (either may be null)
    const double4 *porig_s, *f_s; const double *mass_s; const int3 *image_s; const unsigned *tag_s;
    int integrate; double dt, shear_rate;
};
void launch_local_finish(const LocalFinish &a, const LocalCaller &c, DBox box, int rows_cap, hipStream_t s);
// redistribution of the particles among ALL ranks (after a Lees-Edwards flip of the tilt: pse_team_redistribute_local)
// (1) dest[i] = the rank that owns particle i under the box as it is now, counts[rank] += 1 (counts zeroed by the caller)
void launch_redist_count(const LocalCaller &c, const LocalGeom &g, DBox box, DCells nc, unsigned *dest, int *counts, int *err, hipStream_t s);
// (2) the particles as LOCAL_REC-double records, those for rank d at records[send_off[d] ..] (fill[G] zeroed by the caller)
void launch_redist_pack(const LocalCaller &c, const LocalGeom &g, const unsigned *dest, const int *send_off, int *fill, double *records, hipStream_t s);
// (3) `total` records -> rows [0, total) of the caller's arrays (velocity and acceleration zero: the next step writes them), *n_local
void launch_redist_unpack(const double *records, int total, const LocalCaller &c, hipStream_t s);

}  // namespace pse
