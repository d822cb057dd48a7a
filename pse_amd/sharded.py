"""Multi-GPU drivers of the PSE step (slab-decomposed far field, row-sharded near field; see include/pse_amd.h).

ShardedSimulation  one rank per process / GPU; the C++ team moves data with RCCL on the engine's stream.  The only
                   thing torch.distributed does here is hand the 128-byte RCCL unique id of rank 0 to the other ranks.
LoopbackSimulation all slab ranks in one process on one device (copies instead of collectives): exercises exactly the
                   same phase code as the multi-process path, so the decomposition can be parity-tested on one GPU.
"""
import numpy as np

from .engine import Engine, Team
from .distributed import _to4


def slab_plan(grid, world):
    """Which x planes / y rows of the far-field grid each rank owns (mirrors create_impl in csrc/pse_capi.hip)."""
    nx, ny = grid[0], grid[1]
    if nx % world or ny % world:
        raise ValueError(f"slab decomposition needs Nx and Ny divisible by the number of ranks ({nx} x {ny} over {world})")
    nxl, nyl = nx // world, ny // world
    return [{"rank": r, "x0": r * nxl, "nxl": nxl, "y0": r * nyl, "nyl": nyl} for r in range(world)]


def halo_planes(P):
    """(below, above): planes a rank stores beside its own x slab so that every particle whose own plane lies in the slab
    has its whole support available for the gather (create_impl in csrc/pse_capi.hip)."""
    return (P - 1) // 2, (P + 1) // 2


def cell_slabs(ncx, world):
    """Cell layers along x owned by each rank: the near field is sharded by whole cell layers, so the rows of a rank are
    one contiguous block of the cell-sorted particle arrays (set_cells in csrc/pse_capi.hip rounds ncx down to a
    multiple of the number of ranks)."""
    ncx = ncx // world * world
    if ncx < max(3, world):
        raise ValueError("box too small to split the near field over this many ranks")
    per = ncx // world
    return [(r * per, (r + 1) * per) for r in range(world)]


def kept_layers(ncx, world, rank, depth):
    """Cell layers along x a slab rank orders and keeps particle data for: its own and `depth` ghost layers on either side (two with
    the two-step Lanczos, one otherwise) -- or all of them when that covers the box (slab_need in csrc/pse_capi.hip)."""
    per = ncx // world
    if world == 1 or per + 2 * depth >= ncx:
        return list(range(ncx))
    return sorted({(rank * per - depth + k) % ncx for k in range(per + 2 * depth)})


def slab_book(ncx, world, rank, depth):
    """Where a rank counts the particles of layers it does not keep (prepare() / k_cell_keys in csrc/): per slab, on the first layer
    of that slab the rank does not keep -- every kept layer of a slab lies before or after all of its unkept layers, so the prefix
    sums of all kept cells and of every slab boundary come out global.  Returns {slab: layer} (None: the slab is kept whole)."""
    per = ncx // world
    kept = set(kept_layers(ncx, world, rank, depth))
    out = {}
    for q in range(world):
        out[q] = next((l for l in range(q * per, (q + 1) * per) if l not in kept), None)
    return out


def row_map(ranges):
    """List-row layout of up to three row ranges (own rows first): range k starts at list row base[k], a multiple of 256 (RowMap in
    csrc/pse_kernels.h).  Returns (bases, total list rows)."""
    bases, base = [], 0
    for lo, hi in ranges:
        bases.append(base)
        base += (hi - lo + 255) // 256 * 256
    return bases, base


def two_step_block(G, alpha_prev, beta, uu, first):
    """The scalars of one two-iteration Lanczos block from its Gram sums (k_lz_block in csrc/pse_kernels.hip), for tests: G =
    dict with the eight sums qw1, w1w1, w1w2, w2w2, pw1, pw2, uw2, qq of q = v_j (unnormalised in the first block), p = v_{j-1},
    u = M p, w1 = M q, w2 = M w1.  Returns alpha_j, beta_{j+1}, alpha_{j+1}, beta_{j+2}, |M v_{j+1}|^2."""
    import math
    s2 = 1.0 / G["qq"]; sc = math.sqrt(s2)
    a, b, f = G["qw1"] * s2, G["w1w1"] * s2, G["pw1"] * sc
    if first:
        beta, alpha_prev, uu = 0.0, 0.0, 0.0
    alpha = a
    bp = math.sqrt(b - alpha * alpha - 2.0 * beta * f + beta * beta)
    c, d, e, g, h = G["w1w2"] * s2, G["w2w2"] * s2, b, G["pw2"] * sc, G["uw2"] * sc
    r1w2 = c - alpha * e - beta * g
    r1w1 = b - alpha * a - beta * f
    r1u = g - alpha * f - beta * alpha_prev
    alpha1 = (r1w2 - alpha * r1w1 - beta * r1u) / (bp * bp)
    zz = (d + alpha * alpha * b + beta * beta * uu - 2.0 * alpha * c - 2.0 * beta * h + 2.0 * alpha * beta * g) / (bp * bp)
    qz = (e - alpha * a - beta * f) / bp
    bpp = math.sqrt(max(zz - alpha1 * alpha1 - 2.0 * bp * qz + bp * bp, 0.0))
    return alpha, bp, alpha1, bpp, zz


def exchange_unique_id(rank, make_id, dist):
    """Rank 0 creates the RCCL unique id; everyone receives it (works on any torch.distributed backend)."""
    box = [make_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return box[0]


class TorchTransport:
    """The host-staged transport of a process-per-rank team over ANY torch.distributed backend (gloo on CPU tensors here): what
    the C++ team hands over is host memory, so the exchange is a batch of isend / irecv on tensors that alias it.  Transfers
    between one pair of ranks match in list order on both sides (the team builds its lists that way)."""

    def __init__(self, dist, group=None):
        self.dist, self.group = dist, group

    def exchange(self, ops):
        import torch
        reqs = []
        for send, to, recv, frm in ops:          # receives first: nothing blocks on an unposted receive
            if recv is not None:
                reqs.append(self.dist.irecv(torch.from_numpy(recv), src=frm, group=self.group))
        for send, to, recv, frm in ops:
            if send is not None:
                reqs.append(self.dist.isend(torch.from_numpy(send), dst=to, group=self.group))
        for r in reqs:
            r.wait()

    def allreduce_sum(self, buf):
        import torch
        self.dist.all_reduce(torch.from_numpy(buf), op=self.dist.ReduceOp.SUM, group=self.group)


class _State:
    def __init__(self, n, pos, force, mass):
        import torch
        self.pos = _to4(pos, 0.0)
        self.force = _to4(force, 0.0)
        self.vel = _to4(np.zeros((n, 3)), mass)
        self.accel = torch.zeros((n, 3), dtype=torch.float64, device="cuda")
        self.image = torch.zeros((n, 3), dtype=torch.int32, device="cuda")


class ShardedSimulation:
    def __init__(self, n, box, world, rank, transport="rccl", **kw):
        """transport: "rccl" (one GPU per rank, xGMI) or "host" (staged through host memory over torch.distributed's default group:
        any backend, ranks may even share a GPU -- the process-per-rank driver is tested that way on a one-GPU box)."""
        import torch.distributed as dist
        self.n, self.world, self.rank = n, world, rank
        self.engine = Engine(n, box, n_slabs=world, slab_rank=rank, **kw)
        if transport == "host":
            self.team = Team([self.engine], transport=TorchTransport(dist))
        else:
            uid = exchange_unique_id(rank, Team.unique_id, dist)
            self.team = Team([self.engine], unique_id=uid)

    def describe(self):
        import os
        mode = os.environ.get("PSE_WAVE_MODE") or ("replicated" if self.world == 2 else "slab")
        if getattr(self.team, "_transport", None) is not None:
            return (f"{self.world} ranks over the HOST-STAGED transport (torch.distributed gloo, not RCCL): a functional run of the "
                    f"process-per-rank driver, far field {mode}; not a scaling number")
        if mode == "replicated":
            return (f"{self.world} GPUs: far field kept whole on every rank (at two ranks the all-to-all would cross one xGMI link "
                    f"each way), near field / Lanczos vectors owned by the rank whose cell slab holds the particle (two Lanczos "
                    f"iterations per exchange: two ghost cell layers + the partial sums in one send/recv group), one velocity "
                    f"all-gather per step; particle arrays replicated")
        return (f"{self.world} GPUs: far-field grid in {self.world} x-slabs (RCCL all-to-all transpose, two-sided plane halo for "
                f"the gather) on a side lane next to the near field / Lanczos vectors owned by the rank whose cell slab holds "
                f"the particle (two Lanczos iterations per exchange: two ghost cell layers + the partial sums in one send/recv "
                f"group; all RCCL calls on one communication stream), one velocity all-gather per step; particle arrays replicated")

    def load(self, pos, force, mass=1.0):
        self.s = _State(self.n, pos, force, mass)

    def info(self):
        return self.engine.info()

    def set_timing(self, on):
        self.engine.set_timing(on)

    def phase_times(self):
        return {k: v for k, v in self.engine.info().items() if k.startswith("t_")}

    def mobility(self):
        self.team.mobility([self.s.pos], [self.s.force], [self.s.vel])
        return self.s.vel

    def brownian_velocity(self, kT, dt, timestep, lanczos_m=2):
        s = self.s
        return self.team.brownian_velocity([s.pos], [s.force], [s.vel], kT, dt, timestep, lanczos_m=lanczos_m)

    def step(self, kT, dt, timestep, shear_rate=0.0, lanczos_m=2):
        s = self.s
        return self.team.step([s.pos], [s.vel], [s.accel], [s.image], [s.force], kT, dt, timestep,
                              shear_rate=shear_rate, lanczos_m=lanczos_m)


class LoopbackSimulation:
    def __init__(self, n, box, world, **kw):
        self.n, self.world = n, world
        self.engines = [Engine(n, box, n_slabs=world, slab_rank=r, **kw) for r in range(world)]
        self.team = Team(self.engines)

    def load(self, pos, force, mass=1.0):
        self.s = [_State(self.n, pos, force, mass) for _ in range(self.world)]

    def mobility(self, parts=3):
        self.team.mobility([s.pos for s in self.s], [s.force for s in self.s], [s.vel for s in self.s], parts=parts)
        return [s.vel for s in self.s]

    def brownian_velocity(self, kT, dt, timestep, lanczos_m=2):
        return self.team.brownian_velocity([s.pos for s in self.s], [s.force for s in self.s], [s.vel for s in self.s],
                                           kT, dt, timestep, lanczos_m=lanczos_m)

    def step(self, kT, dt, timestep, shear_rate=0.0, lanczos_m=2):
        S = self.s
        return self.team.step([s.pos for s in S], [s.vel for s in S], [s.accel for s in S], [s.image for s in S],
                              [s.force for s in S], kT, dt, timestep, shear_rate=shear_rate, lanczos_m=lanczos_m)


# ---- owned-particle teams (pse_team_step_local) ------------------------------------------------------------------------------
def x_layer(pos, box, layers):
    """Cell layer along x of every particle, as the engine computes it (frac_coords / cell_coord in csrc/pse_device.h)."""
    xy = box[3] if len(box) > 3 else 0.0
    fx = (pos[:, 0] - xy * pos[:, 1]) / box[0] + 0.5
    fx -= np.floor(fx)
    fx[fx >= 1.0] = 0.0
    return np.minimum((fx * layers).astype(np.int64), layers - 1)


def owner_of(pos, box, layers, world):
    """Rank that owns each particle: whole cell layers along x, layers // world per rank (LocalGeom in csrc/pse_local.h)."""
    return x_layer(pos, box, layers) // (layers // world)


def tilt_flipped(old_box, new_box):
    """True when a box change is a Lees-Edwards flip (xy + 0.5 -> - 0.5: the lattice vector b becomes b -+ a).  A flip re-maps the
    fractional x of a particle by its fractional y: EVERY particle may change slab at once, which the migration inside
    pse_team_step_local (one neighbour per step) cannot follow -- the owner of the particle data redistributes, as HOOMD's domain
    decomposition does.  (Between flips the slabs are co-moving with the shear: an affinely advected particle keeps its slab.)"""
    return abs((new_box[3] if len(new_box) > 3 else 0.0) - (old_box[3] if len(old_box) > 3 else 0.0)) > 0.25


def local_capacity(n, world, layers_per_rank, depth=2, slack=1.3):
    """A row capacity (pse_params.n_max of an owned-particle handle) for n particles of roughly uniform density over `world` ranks:
    own rows + `depth` ghost cell layers on either side, times a slack."""
    per = layers_per_rank
    return int(slack * n / world * (per + 2 * depth) / per) + 4096


class LanczosCount:
    """Starting count of the Lanczos iteration for a time-stepping loop of owned-particle steps, and whether a step queues gated
    extra iterations at all (pse_team_set_lanczos_extra).  A queue-only step never waits for more iterations: it runs the count it
    is given (+ the gated extras) and says afterwards whether that sufficed.  `seen` is fed pse_info.lanczos_m / lanczos_status of a
    COMPLETED step (after a synchronisation): every rank takes the same decisions from the same sums, so every rank holds the same
    two numbers and this object stays identical on all ranks without a message.  Policy: status 1 -> count + 2 and the gated extras
    back on; `settle` consecutive steps that ended at their starting count with status 0 -> no extras (one exchange and seven
    launches fewer per step); anything else -> follow the count the step reported."""

    def __init__(self, team, m=2, settle=3, adaptive=True):
        self.team, self.m, self.settle, self.adaptive = team, max(int(m), 2), settle, adaptive
        self.steady, self.extras_off = 0, False

    def seen(self, lanczos_m, lanczos_status):
        if lanczos_status == 1:
            self.m = max(self.m, int(lanczos_m)) + 2
            self.steady = 0
            self._extras(False)
        elif int(lanczos_m) == self.m:
            self.steady += 1
            if self.adaptive and self.steady >= self.settle:
                self._extras(True)
        elif int(lanczos_m) >= 2:
            self.m, self.steady = int(lanczos_m), 0
        return self.m

    def _extras(self, off):
        if off != self.extras_off:
            self.team.set_lanczos_extra(0 if off else -1)
            self.extras_off = off


class _LocalState:
    """The caller's arrays of one owned-particle rank: capacity rows_own, the first n_local rows live."""

    def __init__(self, cap):
        import torch
        z = lambda *shape, dtype=torch.float64: torch.zeros(shape, dtype=dtype, device="cuda")   # noqa: E731
        self.cap = cap
        self.pos, self.vel, self.force = z(cap, 4), z(cap, 4), z(cap, 4)
        self.accel = z(cap, 3)
        self.image = z(cap, 3, dtype=torch.int32)
        self.tag = z(cap, dtype=torch.int32)
        self.n_local = z(1, dtype=torch.int32)

    def load(self, idx, pos, force, mass):
        import torch
        n = len(idx)
        if n > self.cap:
            raise ValueError(f"{n} particles for a rank whose arrays hold {self.cap}")
        for t in (self.pos, self.vel, self.force, self.accel, self.image):     # (a re-load starts from a clean state: images, velocities)
            t.zero_()
        self.pos[:n, :3] = torch.from_numpy(np.ascontiguousarray(pos[idx])).cuda()
        self.force[:n, :3] = torch.from_numpy(np.ascontiguousarray(force[idx])).cuda()
        self.vel[:n, 3] = mass
        self.tag[:n] = torch.from_numpy(idx.astype(np.int32)).cuda()
        self.n_local.fill_(n)

    def state(self):
        """(tag, pos, image, mass) of the live rows, as NumPy arrays -- synchronises."""
        n = int(self.n_local.item())
        return (self.tag[:n].cpu().numpy(), self.pos[:n].cpu().numpy(), self.image[:n].cpu().numpy(), self.vel[:n, 3].cpu().numpy())

    def load_state(self, tag, pos4, image, mass):
        """Replace the live rows (a redistribution: the particles this rank owns now)."""
        import torch
        n = len(tag)
        if n > self.cap:
            raise ValueError(f"{n} particles for a rank whose arrays hold {self.cap}")
        self.pos[:n] = torch.from_numpy(np.ascontiguousarray(pos4)).cuda()
        self.image[:n] = torch.from_numpy(np.ascontiguousarray(image.astype(np.int32))).cuda()
        self.vel[:n] = 0.0
        self.vel[:n, 3] = torch.from_numpy(np.ascontiguousarray(mass)).cuda()
        self.tag[:n] = torch.from_numpy(np.ascontiguousarray(tag.astype(np.int32))).cuda()
        self.n_local.fill_(n)

    def snapshot(self):
        return [t.clone() for t in (self.pos, self.vel, self.force, self.accel, self.image, self.tag, self.n_local)]

    def restore(self, snap):
        for t, c in zip((self.pos, self.vel, self.force, self.accel, self.image, self.tag, self.n_local), snap):
            t.copy_(c)

    def refresh_force(self, force_dev):
        """force rows in the order the step left the particles in (force_dev: (N, 4) CUDA tensor indexed by tag)."""
        self.force.copy_(force_dev[self.tag.long().clamp_(0, force_dev.shape[0] - 1)])

    def gather(self, out_pos=None, out_vel=None, out_image=None):
        n = int(self.n_local.item())
        tg = self.tag[:n].cpu().numpy()
        if out_pos is not None:
            out_pos[tg] = self.pos[:n, :3].cpu().numpy()
        if out_vel is not None:
            out_vel[tg] = self.vel[:n, :3].cpu().numpy()
        if out_image is not None:
            out_image[tg] = self.image[:n].cpu().numpy()
        return tg


class LocalLoopbackSimulation:
    """All ranks of an owned-particle team in one process on one device (copies instead of RCCL): the decomposition, the
    migration and the ghost exchange of pse_team_step_local on a single GPU."""

    def __init__(self, n, box, world, n_max=None, slack=1.3, **kw):
        probe = host_layers(box, world, **kw)
        self.n, self.world, self.box = n, world, tuple(box)
        cap = n_max or local_capacity(n, world, probe // world, slack=slack)
        self.engines = [Engine(cap, box, n_slabs=world, slab_rank=r, local_rows=1, **kw) for r in range(world)]
        self.layout = self.engines[0].local_layout()
        self.team = Team(self.engines)
        self.s = [_LocalState(self.layout["rows_own"]) for _ in range(world)]

    def load(self, pos, force, mass=1.0):
        import torch
        own = owner_of(pos, self.box, self.layout["layers"], self.world)
        for r in range(self.world):
            self.s[r].load(np.nonzero(own == r)[0], pos, force, mass)
        f4 = np.zeros((len(force), 4)); f4[:, :3] = force
        self.force_dev = torch.from_numpy(f4).cuda()

    def set_box(self, Lx, Ly, Lz, xy):
        flip = tilt_flipped(self.box, (Lx, Ly, Lz, xy))
        for e in self.engines:
            e.set_box(Lx, Ly, Lz, xy)
        self.box = (Lx, Ly, Lz, xy)
        if flip:
            self.redistribute()

    def redistribute(self, on_host=False):
        """Every particle to the rank that owns it under the CURRENT box (after a tilt flip; see tilt_flipped): the engine's own
        pse_team_redistribute_local.  on_host: the host-side re-ownership (gather, owner_of, reload) the tests hold it against."""
        S = self.s
        if not on_host:
            self.team.redistribute_local([s.pos for s in S], [s.vel for s in S], [s.accel for s in S], [s.image for s in S], [s.force for s in S],
                                         [s.tag for s in S], [s.n_local for s in S])
            return
        self.team.local_status()
        parts = [s.state() for s in self.s]
        tag, pos4, image, mass = (np.concatenate([p[k] for p in parts]) for k in range(4))
        own = owner_of(pos4[:, :3], self.box, self.layout["layers"], self.world)
        for r, s in enumerate(self.s):
            idx = np.nonzero(own == r)[0]
            s.load_state(tag[idx], pos4[idx], image[idx], mass[idx])
            s.refresh_force(self.force_dev)

    def step(self, kT, dt, timestep, shear_rate=0.0, lanczos_m=2, integrate=True):
        S = self.s
        m = self.team.step_local([s.pos for s in S], [s.vel for s in S], [s.accel for s in S], [s.image for s in S], [s.force for s in S],
                                 [s.tag for s in S], [s.n_local for s in S], kT, dt, timestep, shear_rate=shear_rate,
                                 integrate=integrate, lanczos_m=lanczos_m)
        for s in S:
            s.refresh_force(self.force_dev)
        return m

    def gather(self):
        """(pos, vel, image) of all particles by tag, and the owner of each -- synchronises."""
        self.team.local_status()
        pos, vel = np.full((self.n, 3), np.nan), np.full((self.n, 3), np.nan)
        image, owner = np.zeros((self.n, 3), dtype=np.int64), np.full(self.n, -1)
        for r, s in enumerate(self.s):
            owner[s.gather(pos, vel, image)] = r
        return pos, vel, image, owner


def host_layers(box, world, xi=0.5, error=1e-3, max_strain=0.5, grid=(0, 0, 0), P=0, rcut=0.0, **_):
    """Cell layers along x of a team of `world` ranks (cells_for in csrc/pse_capi.hip): floor(width / rcut) rounded down to a
    multiple of the rank count, the width taken at the largest tilt the box will see."""
    from .engine import host_select_params
    info = host_select_params(box, xi=xi, error=error, max_strain=max_strain, grid=grid, P=P, rcut=rcut)
    xy = abs(box[3]) if len(box) > 3 else 0.0
    gam = max(xy, max_strain)
    wx = box[0] / np.sqrt(1.0 + gam * gam)
    return int(np.floor(wx / info["rcut"])) // world * world


class LocalShardedSimulation:
    """One owned-particle rank per process: RCCL ("rccl") or the host-staged transport over torch.distributed ("host")."""

    def __init__(self, n, box, world, rank, transport="rccl", n_max=None, slack=1.3, **kw):
        import torch.distributed as dist
        self.n, self.world, self.rank, self.box = n, world, rank, tuple(box)
        cap = n_max or local_capacity(n, world, host_layers(box, world, **kw) // world, slack=slack)
        self.engine = Engine(cap, box, n_slabs=world, slab_rank=rank, local_rows=1, **kw)
        self.layout = self.engine.local_layout()
        if transport == "host":
            self.team = Team([self.engine], transport=TorchTransport(dist))
        else:
            self.team = Team([self.engine], unique_id=exchange_unique_id(rank, Team.unique_id, dist))
        self.s = _LocalState(self.layout["rows_own"])

    def load(self, pos, force, mass=1.0):
        import torch
        own = owner_of(pos, self.box, self.layout["layers"], self.world)
        self.s.load(np.nonzero(own == self.rank)[0], pos, force, mass)
        f4 = np.zeros((len(force), 4)); f4[:, :3] = force
        self.force_dev = torch.from_numpy(f4).cuda()

    def set_box(self, Lx, Ly, Lz, xy):
        flip = tilt_flipped(self.box, (Lx, Ly, Lz, xy))
        self.engine.set_box(Lx, Ly, Lz, xy)
        self.box = (Lx, Ly, Lz, xy)
        if flip:
            self.redistribute()

    def redistribute(self):
        """Every particle to the rank that owns it under the CURRENT box (after a tilt flip; see tilt_flipped): the engine's own
        pse_team_redistribute_local, over the team's transport -- nothing passes through Python."""
        s = self.s
        self.team.redistribute_local([s.pos], [s.vel], [s.accel], [s.image], [s.force], [s.tag], [s.n_local])

    def step(self, kT, dt, timestep, shear_rate=0.0, lanczos_m=2, integrate=True):
        s = self.s
        m = self.team.step_local([s.pos], [s.vel], [s.accel], [s.image], [s.force], [s.tag], [s.n_local], kT, dt, timestep,
                                 shear_rate=shear_rate, integrate=integrate, lanczos_m=lanczos_m)
        s.refresh_force(self.force_dev)
        return m

    def gather_local(self):
        """(tags, pos, vel, image) of the particles this rank owns -- synchronises."""
        self.team.local_status()
        n = int(self.s.n_local.item())
        return (self.s.tag[:n].cpu().numpy(), self.s.pos[:n, :3].cpu().numpy(), self.s.vel[:n, :3].cpu().numpy(),
                self.s.image[:n].cpu().numpy())


class SplitSimulation:
    """TWO ranks, a FUNCTIONAL split instead of a spatial one: both hold all N particles; rank 0 computes the
    real-space half of the Brownian velocity (near-field M.F + the Lanczos M_real^{1/2} psi), rank 1 the wave-space half (spread, FFTs,
    k-space scaling + noise, gather) -- each a single-GPU engine on the whole suspension, no slab all-to-all over the one xGMI link
    between two GPUs.  ONE exchange per step: an all-reduce of the two halves (32 bytes per particle each way); both ranks then
    integrate all particles, bit-identically (a + b = b + a).  `dist`: torch.distributed, world size 2."""

    def __init__(self, n, box, rank, dist, **kw):
        import torch
        if dist.get_world_size() != 2:
            raise ValueError("the functional split is a two-rank mode")
        self.n, self.rank, self.dist, self.box = n, rank, dist, tuple(box)
        self.engine = Engine(n, box, **kw)
        z = lambda *shape, dtype=torch.float64: torch.zeros(shape, dtype=dtype, device="cuda")   # noqa: E731
        self.pos, self.vel, self.force, self.part = z(n, 4), z(n, 4), z(n, 4), z(n, 4)
        self.accel, self.image = z(n, 3), z(n, 3, dtype=torch.int32)
        self._cpu = dist.get_backend() == "gloo"      # (the tests: two processes on one GPU)

    def load(self, pos, force, mass=1.0):
        import torch
        self.pos[:, :3] = torch.from_numpy(np.ascontiguousarray(pos)).cuda()
        self.force[:, :3] = torch.from_numpy(np.ascontiguousarray(force)).cuda()
        self.vel.zero_(); self.vel[:, 3] = mass
        self.accel.zero_(); self.image.zero_()

    def set_box(self, Lx, Ly, Lz, xy):
        self.engine.set_box(Lx, Ly, Lz, xy)
        self.box = (Lx, Ly, Lz, xy)

    def velocity(self, kT, dt, timestep, lanczos_m=2):
        """vel.xyz = the Brownian velocity of all particles, on both ranks; returns the Lanczos count (rank 0's, sent along)."""
        self.part.zero_()
        _, m = self.engine.brownian_velocity_part(self.pos, self.force, kT, dt, timestep, 1 if self.rank == 0 else 2, vel=self.part,
                                                  lanczos_m=lanczos_m)
        self.part[0, 3] = float(m) if self.rank == 0 else 0.0       # (the w column is free: the count rides in it)
        if self._cpu:
            t = self.part.cpu()
            self.dist.all_reduce(t)
            self.part.copy_(t)
        else:
            self.dist.all_reduce(self.part)
        m = int(round(float(self.part[0, 3].item()))) if kT > 0 else lanczos_m
        self.vel[:, :3] = self.part[:, :3]
        return m

    def step(self, kT, dt, timestep, shear_rate=0.0, lanczos_m=2):
        m = self.velocity(kT, dt, timestep, lanczos_m=lanczos_m)
        self.engine.integrate(self.pos, self.vel, self.accel, self.image, self.force, dt, shear_rate=shear_rate)
        return m
