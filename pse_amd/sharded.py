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
