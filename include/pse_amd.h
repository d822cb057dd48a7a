/*
 * pse_amd.h -- C-ABI of the MI355X-native Positively-Split-Ewald (PSE) engine.
 *
 * This is the drop-in boundary for the hot path of stochasticHydroTools/PSE (HOOMD plugin PSEv1):
 * everything below `Stokes::integrateStepOne` (PSEv1/Stokes.cc:429-523), i.e. the driver
 * `gpu_stokes_step_one` (PSEv1/Stokes.cuh:75-111, PSEv1/Stokes.cu:234-365) and the set-up work of
 * `Stokes::setParams` (PSEv1/Stokes.cc:129-424).  Plain C types only; every array argument is a
 * caller-owned DEVICE pointer (as HOOMD's GPUArray handles are, PSEv1/Stokes.cc:436-470) unless a
 * comment says "host".  All entry points return 0 on success or a negative pse_status; the message
 * is available from pse_last_error().  Nothing here ever calls exit() (the reference does:
 * PSEv1/Stokes.cc:203-214, PSEv1/Brownian.cu:543-560).
 *
 * Units and conventions are the reference's: particle radius a = 1, mobility in units of
 * 1/(6 pi eta a) (PSEv1/Stokes.cc:314-319, PSEv1/Helper.cu:326); box centred on the origin with
 * HOOMD's triclinic tilt xy: a1=(Lx,0,0), a2=(xy*Ly,Ly,0), a3=(0,0,Lz) (PSEv1/Mobility.cu:223-230).
 * Arithmetic is fp64 (the reference is effectively fp32), with one stated exception: inside the Lanczos iteration
 * of M_real^{1/2} psi (tolerance `error`) the near-field operator reads its pair coefficients -- f(r) and d sqrt|(g - f) / r^2| -- from
 * the 16-byte records of the per-step pair list, rounded to single-precision accuracy (f to 2^-25 absolute, the vector to 2^-22 of its
 * largest component), and (single GPU) its neighbours' vector rows to 2^-39 of their largest component; sums accumulate in fp64, the
 * operator stays exactly symmetric, the deterministic U = M.F never sees those numbers (oracle/pse_oracle.c pair_term restates the
 * rounding of the coefficients operation by operation; DESIGN.md sections 2 and 4).
 */
#ifndef PSE_AMD_H
#define PSE_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

/* HOOMD Scalar4 / Scalar3 / int3 with Scalar = double */
typedef struct { double x, y, z, w; } pse_double4;
typedef struct { double x, y, z; } pse_double3;
typedef struct { int x, y, z; } pse_int3;

typedef struct pse_handle pse_handle;

enum pse_status {
    PSE_OK = 0,
    PSE_ERR_INVALID = -1,   /* bad argument / parameter combination */
    PSE_ERR_HIP = -2,       /* HIP runtime error */
    PSE_ERR_FFT = -3,       /* rocFFT error */
    PSE_ERR_COMM = -4,      /* RCCL error */
    PSE_ERR_NUMERIC = -5    /* Lanczos / eigen-solve breakdown */
};

/* Constructor arguments: what Stokes::Stokes + setShear + setParams receive
 * (PSEv1/Stokes.cc:85-111, PSEv1/Stokes.h:118-121, PSEv1/Stokes.cc:129-236), plus explicit
 * overrides the reference's rule cannot express. Zero means "use the reference rule". */
typedef struct pse_params {
    unsigned int n_max;      /* capacity in particles (N_total of the reference) */
    double Lx, Ly, Lz, xy;   /* box; |xy| <= 0.5 (HOOMD flips the box there; beyond it the minimum image is not exact): else PSE_ERR_INVALID */
    double xi;               /* Ewald splitting parameter (PSEv1/Stokes.cc:91) */
    double error;            /* tolerance for all approximations (m_error) */
    double max_strain;       /* largest |xy| the box will take; sizes P (PSEv1/Stokes.cc:217-229) */
    unsigned int seed;       /* RNG seed as used on the device (the host class hashes the user seed, Stokes.cc:102) */
    int Nx, Ny, Nz;          /* FFT grid override (0 = PSEv1/Stokes.cc:138-199).  The reference's rule makes the three spacings equal up
                                to the rounding of the sizes; an override whose coarsest spacing takes the spreading Gaussian (width from
                                the finest) out of the double range over its support is refused: PSE_ERR_INVALID (pse_set_box likewise) */
    int P;                   /* support override (0 = PSEv1/Stokes.cc:225-233) */
    double rcut;             /* real-space cutoff override (0 = PSEv1/Stokes.cc:135) */
    int device;              /* HIP device ordinal, -1 = current device */
    int n_slabs;             /* far-field slab decomposition: number of ranks (<=1: single GPU) */
    int slab_rank;           /* this rank's slab index */
    int local_rows;          /* != 0 (with n_slabs >= 2): an OWNED-PARTICLE rank -- it is driven through pse_team_step_local, n_max is the
                                capacity of its row space (own particles + the ghost layers of both neighbours), not the particle count
                                of the suspension; the replicated-state team calls refuse such a handle */
} pse_params;

typedef struct pse_info {
    int Nx, Ny, Nz, P;
    double rcut, xi, eta, gaussm, lambda, self_mobility, hx, hy, hz;
    int ncell_x, ncell_y, ncell_z;
    int lanczos_m;            /* vectors used by the last Brownian call */
    int lanczos_matvecs;      /* near-field mat-vecs of the last Brownian call */
    double lanczos_stepnorm;  /* last relative step norm */
    /* device time of the phases of the most recent call, ms (hipEvent, only if timing enabled) */
    double t_sort, t_spread, t_fft_fwd, t_scale, t_fft_inv, t_gather, t_real, t_lanczos, t_integrate, t_comm, t_total;
    unsigned long long device_bytes;  /* workspace owned by the handle */
    double t_matvec;                  /* one near-field mat-vec from the per-step pair list (inside t_lanczos) */
    double t_records;                 /* binning + the 64-byte far-field particle records (t_spread is the spread kernel alone) */
    int lanczos_exchanges;            /* team calls: exchanges the last Brownian call's Lanczos iteration issued (two iterations
                                         per exchange where the decomposition allows: ceil(m / 2) instead of m) */
    int lanczos_status;               /* queue-only Brownian calls (pse_set_async): 0 the step norm passed, 1 the queued iterations ran
                                         out first (the result uses the last size checked: raise the starting count), 2 a
                                         non-finite coefficient or a failed eigen-solve */
    unsigned long long lanczos_open_calls;   /* queue-only calls of this handle so far that ended with lanczos_status != 0 (sticky: a loop
                                                that reads pse_info every hundred steps still learns that one of them ran out) */
} pse_info;

/* -- life cycle: replaces Stokes::Stokes/setParams/~Stokes (PSEv1/Stokes.cc:85-118,129-424) ------------- */
int pse_create(const pse_params *params, pse_handle **out);
int pse_destroy(pse_handle *h);
/* box change under Lees-Edwards shear: what gpu_stokes_SetGridk_kernel re-derives every step
 * (PSEv1/Helper.cu:285-332, called at PSEv1/Stokes.cu:298). Only xy may differ from the creation box by more
 * than round-off unless the cell grid still fits. */
int pse_set_box(pse_handle *h, double Lx, double Ly, double Lz, double xy);
/* run on this hipStream_t (default: the null stream, as the reference does).  The work of every entry point is ordered on
 * this stream.  By default the entry points are not free of host synchronisation: (1) with a neighbour skin in use (the
 * default, below) every evaluation whose kept list may still be valid waits for the stream and reads one flag back before it
 * queues anything (the distance check HOOMD's NeighborList does on the host side too); (2) Brownian calls read the Lanczos
 * scalars back once per convergence check, as the reference does per iteration (PSEv1/Brownian.cu:446-499).
 * pse_set_async(h, 1) (or pse_set_neighbor_skin(h, 0)) removes (1): deterministic evaluations then only queue work. */
int pse_set_stream(pse_handle *h, void *hip_stream);
/* Asynchronous submission (off by default).  With it on, the deterministic entry points (pse_mobility, pse_pair_repulsion) only
 * QUEUE work on the handle's stream and return: nothing is read back, the host never waits, so a call can be captured into a
 * hipGraph by the caller (hipStreamBeginCapture on the handle's stream ... pse_mobility ... hipStreamEndCapture) and replayed
 * with new positions and forces in the same arrays -- tests/test_gpu_async.py does exactly that.  Whether the kept neighbour list
 * is still valid is then decided ON THE DEVICE: a call gathers the particles into the order of the last build, checks every
 * displacement against r_buff / 2, and queues BOTH chains -- sort + cell walk + new list, and the kept-list pass -- whose kernels
 * read the outcome and leave at once if it is not theirs (never a stale list, no round trip; about eight empty launches per
 * call are the price).  Brownian calls (pse_brownian_velocity, pse_step, pse_sqrt_mreal) rebuild the list every time in this mode
 * and take the Lanczos decision ON THE DEVICE too: the iterations of the starting count *lanczos_m are queued, one workgroup
 * computes the tridiagonal square roots and the step norm the reference computes on the host (LAPACKE_spteqr and the loops at
 * PSEv1/Brownian.cu:540-582, 673-724), PSE_LANCZOS_EXTRA (default 2) further iterations follow, each with its own decision
 * and gated on the outcome so far, and the final combination reads m and its coefficients from device memory.  Such a call can be
 * captured and replayed like the deterministic ones; on return *lanczos_m is the m of the most recent call whose outcome has
 * already reached the host (feed it to the next call), pse_get_info after a stream synchronisation gives the m, the step norm
 * and pse_info.lanczos_status of the last completed call (1: the queue ran out before the step norm passed).  Starting counts
 * above 96 - PSE_LANCZOS_EXTRA take the host-checked path.  Per-phase timing (pse_set_timing) synchronises by definition and
 * takes the host-checked path as well.  One warm-up call outside the capture first (kernels set their shared-memory attributes
 * on first use). */
int pse_set_async(pse_handle *h, int enabled);
/* The random numbers of a Brownian call are keyed by (particle or grid node, timestep): a captured call would replay the timestep
 * it was captured with.  With a device word registered here every Brownian call draws its noise at timestep + *device_word,
 * read on the device when the kernels run -- the caller advances the word between replays.  NULL (default): timestep alone. */
int pse_set_timestep_offset(pse_handle *h, const unsigned int *device_word);
/* test hook: the device-side decision of the most recent deterministic evaluation in asynchronous mode (synchronises):
 * 0 the kept list was reused, != 0 it was rebuilt, -1 the call did not take the two-chain path */
int pse_debug_last_gate(pse_handle *h, int *gate);
/* Neighbour list kept across calls: replaces the NeighborListGPUBinned(rcut, r_buff = 0.4) with setEvery(1, dist_check)
 * that PSEv1/integrate.py:60,79 builds and Stokes::integrateStepOne refreshes with m_nlist->compute (PSEv1/Stokes.cc:433).
 * Pairs closer than rcut + r_buff are remembered at a build; later calls with the same N, group and box first check that no
 * particle has moved more than r_buff / 2 since (one flag read back per call) and then skip the cell sort and the cell
 * walk.  Default 0.4 (PSE_SKIN in the environment overrides; 0 = rebuild every call).  r_buff may not exceed the
 * creation value: cells and list capacity are sized for it.  Not kept on slab ranks (multi-GPU) or when rcut + r_buff
 * exceeds half the box.  Results do not depend on it beyond the order of the fp64 pair sums. */
int pse_set_neighbor_skin(pse_handle *h, double r_buff);
/* the r_buff in use and how many calls built / reused the list so far (any pointer may be null) */
int pse_neighbor_stats(pse_handle *h, double *r_buff, unsigned long long *builds, unsigned long long *reuses);
/* per-phase hipEvent timing into pse_info (adds host synchronisation; off by default) */
int pse_set_timing(pse_handle *h, int enabled);
int pse_get_info(pse_handle *h, pse_info *info);
const char *pse_last_error(void);

/* -- the hot path ------------------------------------------------------------------------------------- */
/* U = M.F, deterministic (kT = 0 branch of gpu_stokes_CombinedMobilityBrownian_wrap, PSEv1/Brownian.cu:772-923;
 * the reference's dedicated but uncalled entry is gpu_stokes_Mobility_wrap, PSEv1/Mobility.cu:729-782).
 * group_members (device, may be NULL = identity) lists the N particle indices acted on, exactly like
 * d_group_members / group_size; vel[idx].xyz is written, vel[idx].w is preserved (PSEv1/Mobility.cu:473-475).
 * parts: 1 = real-space (+self) only (gpu_stokes_Mreal_kernel), 2 = wave-space only
 * (gpu_stokes_Mwave_wrap, PSEv1/Mobility.cu:515-575), 3 = both. */
int pse_mobility(pse_handle *h, const pse_double4 *pos, const pse_double4 *force, pse_double4 *vel,
                 const unsigned int *group_members, unsigned int N, int parts);

/* U = M.F + sqrt(2kT/dt) M^{1/2} psi  (PSEv1/Brownian.cu:772-923) without moving the particles.
 * lanczos_m: in = number of Lanczos vectors to start from, out = number used (the reference's int& m_Lanczos). */
int pse_brownian_velocity(pse_handle *h, const pse_double4 *pos, const pse_double4 *force, pse_double4 *vel,
                          const unsigned int *group_members, unsigned int N,
                          double kT, double dt, unsigned int timestep, int *lanczos_m);

/* The two halves of that evaluation on their own -- a FUNCTIONAL split for two GPUs (one GPU the real-space half, the other the
 * wave-space half, each on ALL particles: no slab all-to-all over the one link between them).  parts = 1:
 * M_real.F + sqrt(2kT/dt) M_real^{1/2} psi (the Lanczos half: *lanczos_m as above); parts = 2: M_wave.F + the k-space noise
 * (gpu_stokes_BrownianGridGenerate, PSEv1/Brownian.cu:153-345; *lanczos_m untouched); 3 = pse_brownian_velocity.  The halves of one
 * (kT, dt, timestep, seed) add up to the whole: the caller sums them (an all-reduce between the two ranks) and integrates with
 * pse_integrate -- K15 alone (gpu_stokes_step_one_kernel, PSEv1/Stokes.cu:137-192: pos += (vel + shear_rate y xhat) dt, wrap, accel). */
int pse_brownian_velocity_part(pse_handle *h, const pse_double4 *pos, const pse_double4 *force, pse_double4 *vel,
                               const unsigned int *group_members, unsigned int N, double kT, double dt, unsigned int timestep,
                               int parts, int *lanczos_m);
int pse_integrate(pse_handle *h, pse_double4 *pos, const pse_double4 *vel, pse_double3 *accel, pse_int3 *image,
                  const pse_double4 *net_force, const unsigned int *group_members, unsigned int N, double dt, double shear_rate);

/* One full integration step: the replacement for gpu_stokes_step_one (PSEv1/Stokes.cuh:75-111).
 * Computes vel, then pos += (vel + shear_rate*y*xhat)*dt, wraps into the box updating image, accel = F/mass
 * with mass = vel.w (PSEv1/Stokes.cu:137-192). */
int pse_step(pse_handle *h, pse_double4 *pos, pse_double4 *vel, pse_double3 *accel, pse_int3 *image,
             const pse_double4 *net_force, const unsigned int *group_members, unsigned int N,
             double kT, double dt, unsigned int timestep, double shear_rate, int *lanczos_m);

/* M_real^{1/2} psi by Lanczos alone (gpu_stokes_BrealLanczos_wrap, PSEv1/Brownian.cu:357-765, without the
 * sqrt(2kT/dt) factor): out = M_real^{1/2} psi for a caller-supplied psi. */
int pse_sqrt_mreal(pse_handle *h, const pse_double4 *pos, const pse_double4 *psi, pse_double4 *out,
                   const unsigned int *group_members, unsigned int N, double tol, int *lanczos_m);

/* The random vectors the Brownian step draws (for parity tests against the oracle's Philox stream):
 * psi[idx].xyz = particle noise (gpu_stokes_BrownianGenerate_kernel, PSEv1/Brownian.cu:99-130). */
int pse_random_psi(pse_handle *h, pse_double4 *psi, const unsigned int *group_members, unsigned int N,
                   unsigned int timestep);

/* -- introspection used by the parity tests ------------------------------------------------------------ */
/* evaluate the device's real-space functions f(r), g(r) (the replacement of the m_ewaldC1 table,
 * PSEv1/Stokes.cc:334-422) at n host radii -> host arrays */
int pse_eval_realspace(pse_handle *h, const double *r_host, int n, double *f_host, double *g_host);
/* Force provider next to the path (the step consumes net_force from its host, PSEv1/Stokes.cc:447, and the
 * reference's example has none): soft repulsion F_i = sum_j k (sigma - r)(r_i - r_j)/r over minimum-image pairs with
 * r < sigma <= rcut, evaluated from the engine's cell list.  force[group[i]].xyz is overwritten (accumulate = 0) or
 * incremented (accumulate = 1); w is kept. */
int pse_pair_repulsion(pse_handle *h, const pse_double4 *pos, pse_double4 *force, const unsigned *group, unsigned N,
                       double k, double sigma, int accumulate);
/* copy the three real-space grids (x-major, z fastest: idx = (x*Ny + y)*Nz + z, PSEv1/Mobility.cu:233) of the
 * most recent spread (stage 0) or inverse FFT (stage 1) to a host buffer of 3*nx_local*Ny*Nz doubles */
int pse_debug_copy_grid(pse_handle *h, int stage, double *host_out);
/* sort + spread only (gpu_stokes_Spread_kernel, PSEv1/Mobility.cu:114-252): leaves the three force grids in place for
 * pse_debug_copy_grid, so the spread can be compared node by node and not only through the gather */
int pse_debug_spread(pse_handle *h, const pse_double4 *pos, const pse_double4 *force, const unsigned int *group_members,
                     unsigned int N);
/* the k-space operator of n grid nodes (i, j, k) (host array of 3 n ints, 0 <= k < Nz: any node of the reference's full C2C
 * grid) exactly as the scaling kernels evaluate it, for the current box: out_host[5 t ..] = kx, ky, kz, w sinc^2,
 * sqrt(w) sinc -- what gpu_stokes_SetGridk_kernel tabulates (PSEv1/Helper.cu:285-332: index folding, sheared ky, scale
 * factor w) and gpu_stokes_Green_kernel / gpu_stokes_BrownianGridGenerate_kernel multiply by (PSEv1/Mobility.cu:290,
 * PSEv1/Brownian.cu:274-276) */
int pse_debug_kvector(pse_handle *h, int n, const int *ijk_host, double *out_host);
/* what pse_create's grid-placement planner did (single-GPU engines with grids of >= 96 MB of spectra; PSE_PLACE_TRIALS=K, default 10,
 * 0 or 1: off): the transform passes that read the spectra run 5 - 20 % faster or slower depending on where the driver placed the
 * two far-field grids, for the life of the allocation, so pse_create allocates the pair up to K times (it stops at the first pair 5 % below the slowest seen), times the x pass and the
 * inverse y + z passes on each and keeps the fastest (+ 5 - 20 ms at create; results do not depend on the choice).  *tried = pairs timed (0: planner off or
 * not applicable), ms_first / ms_kept = the probe's time on the first pair and on the kept one.  No reference counterpart (cuFFT
 * plans own their work areas, PSEv1/Stokes.cc:262-269). */
int pse_debug_grid_placement(pse_handle *h, int *tried, float *ms_first, float *ms_kept);
/* the 16-byte form in which the single GPU's pair-list mat-vec reads its NEIGHBOURS' rows of the Lanczos vector (three 40-bit
 * mantissas under one exponent, written beside every new vector; the double rows stay the truth for the diagonal term, the sums and
 * the final combination): rows_host [n][3] -> packed on the device -> unpacked -> out_host [n][3].  |out - in| <= 2^-39 of the row's
 * largest component.  Why: a gather of 64 scattered rows costs the texture path the same whether it fetches 8 or 16 bytes per lane,
 * and 24 bytes of doubles are two of them per pair.  PSE_VQ=0 reads doubles (A/B).  No reference counterpart (the reference's
 * mat-vec recomputes every pair from single-precision positions, PSEv1/Mobility.cu:495-600). */
int pse_debug_vq_roundtrip(int n, const double *rows_host, double *out_host);
/* the dominant kernel of a Brownian step by itself: `reps` launches of the pair-list mat-vec of a Lanczos iteration (k_mreal_list; the
 * list, the vectors and the mirror of the Brownian call that has just ended; results go to scratch) back to back on the engine's
 * stream between ONE pair of events -> milliseconds per launch.  What bench.py's `roofline` divides by: an event pair around a single
 * launch inside a step carries 5 - 8 us of marker latency, this one a fraction of a microsecond.  Single-GPU engines, right after
 * pse_step / pse_brownian_velocity with kT > 0 (PSE_ERR_INVALID otherwise). */
int pse_debug_matvec_ms(pse_handle *h, int reps, float *ms_per_launch);

/* -- multi-GPU: slab-decomposed far field + row-sharded near field (new design; the reference is single-GPU,
 *    PSEv1/Stokes.cc:104) ---------------------------------------------------------------------------------------
 * Each rank is a handle created with pse_params.n_slabs = G and its own slab_rank.  The far-field grid is cut into
 * G slabs of x planes (the slowest index of the reference layout, PSEv1/Mobility.cu:233): spread and the 2-D (y,z)
 * transforms are slab-local, an all-to-all transposes to y-slabs for the 1-D x transforms and the k-space scaling,
 * and back; the gather takes (P-1)/2 halo planes from the previous slab and (P+1)/2 from the next.  The near-field cell
 * layers along x are split the same way, so a rank owns one contiguous block of the cell-sorted rows: its near field,
 * its rows of every Lanczos vector and its gathered velocities, exchanged once per call.  The Lanczos iteration of a team
 * runs TWO iterations per exchange where the decomposition allows (two ghost cell layers per neighbour, the second near-field
 * product of a block on the own rows, the recurrence coefficients from eight summed products: pse_info.lanczos_exchanges),
 * one per iteration otherwise; every exchange of a team -- ghost rows with the ranks' partial sums, all-to-alls, plane halos,
 * the final row all-gather -- is ONE group of point-to-point transfers, all posted on one communication stream in program
 * order, while the far-field chain and the near field / Lanczos chain run on two compute streams next to each other.
 * Particle arrays are replicated: every rank passes the same pos/force and ends the call with the same vel/pos.
 * A team binds the local ranks to a transport: one member per process + the RCCL unique id of rank 0 (production, one
 * process per GPU), or all G members in one process with id = NULL (in-process loopback on one device, for tests). */
typedef struct pse_team pse_team;
int pse_team_unique_id(void *id128_host);   /* host buffer of 128 bytes, call on rank 0 and distribute */
int pse_team_create(pse_handle **members, int n_members, const void *id128_host, pse_team **out);
/* A third transport, supplied by the host program: one member per process as with RCCL, but every exchange is staged through
 * pinned host memory and handed to a callback -- a list of point-to-point transfers between ranks (every rank of the team
 * calls with its own list at the same point of the step; transfers between one pair of ranks match in list order).  Buffers are
 * host memory, counts are in doubles, send_to / recv_from = -1 where an entry has no send / no
 * receive; return 0 on success.  The exchanges are exactly those of the RCCL transport (same buffers, counts, peers and
 * order: both run through one transfer list), which is what makes the process-per-rank driver testable where RCCL cannot run
 * (two ranks on one GPU), e.g. over torch.distributed's gloo backend; it is also a fallback for nodes without xGMI. */
typedef struct pse_host_xfer {
    const double *send; size_t send_count; int send_to;
    double *recv; size_t recv_count; int recv_from;
} pse_host_xfer;
typedef struct pse_transport {
    void *user;
    int (*exchange)(void *user, int n_xfers, const pse_host_xfer *xfers);
    int (*allreduce_sum)(void *user, double *host_buf, size_t count);   /* not called since round 4 (may be NULL): the Lanczos sums
                                                                           travel as small blocks inside `exchange` */
} pse_transport;
int pse_team_create_transport(pse_handle *member, const pse_transport *transport, pse_team **out);
/* Destroy the team BEFORE its members: the members must outlive it (an in-process team lends member 0's side stream to the
 * others and hands every member its own back here). */
int pse_team_destroy(pse_team *team);
/* the three hot-path entry points for a team; pointer arrays have one entry per local member, in members order */
int pse_team_mobility(pse_team *team, const pse_double4 *const *pos, const pse_double4 *const *force,
                      pse_double4 *const *vel, const unsigned int *group_members, unsigned int N, int parts);
int pse_team_brownian_velocity(pse_team *team, const pse_double4 *const *pos, const pse_double4 *const *force,
                               pse_double4 *const *vel, const unsigned int *group_members, unsigned int N,
                               double kT, double dt, unsigned int timestep, int *lanczos_m);
int pse_team_step(pse_team *team, pse_double4 *const *pos, pse_double4 *const *vel, pse_double3 *const *accel,
                  pse_int3 *const *image, const pse_double4 *const *net_force, const unsigned int *group_members,
                  unsigned int N, double kT, double dt, unsigned int timestep, double shear_rate, int *lanczos_m);

/* -- owned-particle teams (round 5) ----------------------------------------------------------------------------------------
 * The calls above replicate the particle state: every rank passes all N particles, bins all of them, and takes part in an
 * all-gather of the velocities every step.  Here a rank passes ONLY THE PARTICLES IT OWNS -- those of its x slab of the box, as
 * HOOMD's domain decomposition hands them to a plugin (the reference is single-GPU, PSEv1/Stokes.cc:104; per-step data flow
 * PSEv1/Stokes.cc:433-514) -- and the engine does what HOOMD's Communicator would: particles that have left the slab migrate to the
 * neighbour, the ghost layers the near field, the spreading and the two-step Lanczos blocks need arrive from both neighbours,
 * all in ONE exchange of fixed-size messages at the start of the step.  No rank touches the other N (G - 1) / G particles, there
 * is no all-gather, and NOTHING is read back inside a step: row counts and ranges live in device memory, the Lanczos decision
 * is taken on the device (as with pse_set_async), every exchange has host-known sizes (capacities).
 *
 * Handles: pse_params.local_rows = 1, n_slabs = G >= 2, n_max = row capacity (pse_local_layout tells how it is divided).  Needs
 * >= 3 cell layers per rank along x (>= 4 with two ranks), the pair list and the real-space table in LDS.
 * Arrays (per member, device): rows [0, *n_local) of pos / vel / accel / image / net_force / tag are the particles the rank owns
 * (capacity: rows_own of pse_local_layout); tag = the particle's global index (keys the particle noise, PSEv1/Brownian.cu:117;
 * identifies it across ranks); vel.w = mass.  On entry a particle may have left the slab by less than a slab width (the last
 * step moved it).  On return the arrays hold the particles the rank owns NOW, in the engine's cell order (pos, vel, accel,
 * image, tag rewritten; net_force is an input: the caller recomputes it for the new order), *n_local their number (a device
 * word: the host learns it when it asks).  integrate = 0: velocities only, positions unchanged (but reordered likewise).
 * kT = 0: deterministic.  *lanczos_m as for queue-only calls (pse_set_async).  Errors that show on the device only (a capacity
 * exceeded, a particle that moved further than a neighbour) are sticky: pse_team_local_status reads them (synchronises), and
 * the next call refuses to start.
 * Shear: the slabs are slabs of the FRACTIONAL x coordinate, so between two calls of pse_set_box that follow the strain
 * continuously an affinely advected particle keeps its slab.  A Lees-Edwards FLIP of the tilt (xy + 0.5 -> - 0.5,
 * PSEv1/VariantShearFunction.cc:34-43) re-maps the fractional x of every particle by its fractional y: the owner of the particle
 * data redistributes them before the next call (what HOOMD's domain decomposition does at a flip): pse_team_redistribute_local
 * below -- a step cannot follow it (flag 8). */
int pse_team_step_local(pse_team *team, pse_double4 *const *pos, pse_double4 *const *vel, pse_double3 *const *accel,
                        pse_int3 *const *image, const pse_double4 *const *net_force, unsigned int *const *tag,
                        unsigned int *const *n_local, double kT, double dt, unsigned int timestep, double shear_rate,
                        int integrate, int *lanczos_m);
/* The Lees-Edwards flip for an owned-particle team: after pse_set_box has taken the tilt of EVERY member through the flip, one call
 * re-owns every particle by its fractional x under the new box -- any particle may go to any rank -- through the team's own transfer
 * list (count rows first, then exactly the records that move, in the wire format of the step's first exchange).  Arrays as for
 * pse_team_step_local (net_force is rewritten too: the rows have a new order; vel.xyz and accel come back zero, vel.w = mass kept);
 * on return -- the work is queued on the members' streams -- rows [0, *n_local) hold what the rank owns now.  The call waits for the
 * streams twice (the host sizes the second exchange): it runs once per unit of strain.  If some rank's arrays could not hold what it
 * would own, NOTHING is moved and every rank returns PSE_ERR_INVALID (every rank sees every count row).
 * HOOMD call site: where the reference's box-tilt updater wraps the strain (PSEv1/VariantShearFunction.cc:34-43). */
int pse_team_redistribute_local(pse_team *team, pse_double4 *const *pos, pse_double4 *const *vel, pse_double3 *const *accel,
                                pse_int3 *const *image, pse_double4 *const *net_force, unsigned int *const *tag,
                                unsigned int *const *n_local);
/* Iterations an owned-particle step queues beyond its starting count *lanczos_m, in blocks of two, every kernel of them gated on
 * the device-side decision (default -1: the members' PSE_LANCZOS_EXTRA, 2).  A time-stepping loop whose last steps all ended
 * at their starting count with pse_info.lanczos_status 0 can set 0: the gated block -- one more exchange, seven launches that
 * leave at once -- is then not queued at all (~ 45 us of a 0.65 ms rank step at the metric point); a step whose count then does
 * not suffice says so (lanczos_status 1, the result uses the last size) and the caller goes back to the default and a larger
 * count.  The same value on every rank of the team (it decides how many exchanges a step has): derive it from numbers every rank
 * holds -- pse_info after a synchronisation is one: all ranks take the same decisions from the same sums.
 * Between processes *lanczos_m of pse_team_step_local comes back UNCHANGED (within one process: the most recent m that has
 * reached the host): read pse_info.lanczos_m after a synchronisation and pass the same count on every rank. */
int pse_team_set_lanczos_extra(pse_team *team, int extra);
/* the same for the queue-only calls of a single-GPU handle (pse_set_async): iterations queued beyond the starting count, each gated on
 * the device-side decision (default -1: PSE_LANCZOS_EXTRA, 2).  With 0 a step whose starting count suffices queues no gated iteration at
 * all (ten launches that would leave at once: ~ 25 us of a 3 ms step at the metric point); pse_info.lanczos_status = 1 /
 * lanczos_open_calls say when it did not.  The reference iterates until the step norm passes (PSEv1/Brownian.cu:606-724). */
int pse_set_lanczos_extra(pse_handle *h, int extra);
/* row capacities of an owned-particle handle: own rows (= capacity the caller's arrays need), ghost rows per side, records per
 * neighbour message of the first exchange; cell layers along x in all and per rank (any pointer may be null) */
int pse_local_layout(pse_handle *h, int *rows_own, int *rows_ghost, int *records, int *layers, int *layers_per_rank);
/* waits for the team's streams; flags[member] = 0 or a combination of 1 own rows exceeded, 2 ghost rows exceeded, 4 message
 * capacity exceeded, 8 a particle moved beyond the neighbour's slab, 16 *n_local above the capacity; returns PSE_ERR_INVALID if any */
int pse_team_local_status(pse_team *team, int *flags);

/* -- self-diagnosis of a team call (for the first run on a multi-GPU node: one number would say nothing about where the time went)
 * With it on, every exchange of a call is bracketed by two events on the stream of the lane that issues it, and the spans of
 * the two lanes are taken the same way; pse_team_get_diag waits for the call and reads them.  kind: 0 the first exchange of an
 * owned-particle step (migrants + ghosts), 1 Lanczos (ghost rows of the mat-vec results + the ranks' partial sums), 2 all-to-all
 * of the far-field transpose, 3 plane halo of the gather, 4 velocity all-gather (replicated-state calls), 5 other ghost rows. */
#define PSE_DIAG_MAX 48
typedef struct pse_team_diag {
    int n_exchanges;
    int kind[PSE_DIAG_MAX];
    int lane[PSE_DIAG_MAX];                   /* 0 main lane (sort, near field, Lanczos, update), 1 far-field lane */
    double device_us[PSE_DIAG_MAX];           /* between the two events: for RCCL the hand-over to the communication stream, the
                                                 transfers and the hand-over back; host-staged transport: incl. the host's part */
    double host_us[PSE_DIAG_MAX];             /* host time spent issuing it (host-staged transport: the whole staged exchange) */
    unsigned long long bytes[PSE_DIAG_MAX];   /* sent by this rank */
    double main_lane_ms, side_lane_ms;        /* first to last event of the lane's stream inside the call (side: fork to join; 0: one lane) */
    double critical_path_ms;                  /* = main_lane_ms: the main lane joins the far-field lane before it ends */
} pse_team_diag;
int pse_team_set_diag(pse_team *team, int enabled);
int pse_team_get_diag(pse_team *team, pse_team_diag *out);

/* Developer switch for an IN-PROCESS team (measurement, not a result): from now on the team queues the work of the one member
 * with this slab rank only -- its kernels on both lanes and the copies that stand for what it receives; the other members'
 * buffers keep what the last full call left there.  The wall time of a call is then that rank's critical path with a GPU to
 * itself, lanes overlapping as they would (tools/perf_team.py --solo); the numbers it returns are not meaningful.  Call with the
 * same positions as the last full call and do not integrate; slab_rank < 0 switches it off. */
int pse_team_debug_solo(pse_team *team, int slab_rank);

/* host-only: t = T^{1/2} e_1 of the Lanczos tridiagonal (alpha[0..m), beta[1..m)); replaces LAPACKE_spteqr +
 * the host loops at PSEv1/Brownian.cu:540-582. Exposed so the eigen-solver can be tested without a GPU. */
int pse_host_lanczos_sqrt_e1(int m, const double *alpha, const double *beta, double *t);
/* host-only: the parameter rule of Stokes::setParams (PSEv1/Stokes.cc:129-236,319) without creating a handle */
int pse_host_select_params(const pse_params *params, pse_info *info);

#ifdef __cplusplus
}
#endif
#endif /* PSE_AMD_H */
