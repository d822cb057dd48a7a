"""CPU tests (gloo, world_size 2, rendezvous on 127.0.0.1) of the multi-GPU decomposition's host logic and index math.
The kernels are stood in for by the NumPy oracle; what is under test is what the C++ team driver does between kernels:
slab ownership, the [3][G][nxl][nyl][Nzh] pack layout, the all-to-all transpose, the transposed-layout k-space scaling
and the way back, plus the unique-id hand-off helper used to bootstrap RCCL."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from conftest import make_suspension
    from oracle import pse_port as pp
    from pse_amd.sharded import cell_slabs, exchange_unique_id, halo_planes, slab_plan
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        uid = exchange_unique_id(rank, lambda: bytes(range(128)), dist)
        assert uid == bytes(range(128))

        n = 200
        pos, force, box = make_suspension(n, L=16.0, xy=0.25)
        p = pp.select_params(box, 0.5, 1e-3, 0.5, grid=(16, 12, 10))
        Nx, Ny, Nz = p["grid"]; Nzh = Nz // 2 + 1
        plan = slab_plan(p["grid"], world)[rank]
        nxl, nyl, x0, y0 = plan["nxl"], plan["nyl"], plan["x0"], plan["y0"]
        real = pp.spread(pos, force, box, p)                                    # every rank could spread everything ...
        mine = real[:, x0:x0 + nxl]                                             # ... but only owns these planes
        spec2d = np.fft.rfft2(mine, axes=(2, 3))                                # [3][nxl][Ny][Nzh]
        pack = spec2d.reshape(3, nxl, world, nyl, Nzh).transpose(0, 2, 1, 3, 4) # [3][G][nxl][nyl][Nzh]
        recv = np.empty_like(pack)
        for c in range(3):
            a = torch.from_numpy(np.ascontiguousarray(pack[c])); b = torch.empty_like(a)
            dist.all_to_all_single(b, a)
            recv[c] = b.numpy()
        tr = recv.reshape(3, Nx, nyl, Nzh)                                      # [3][Nx][nyl][Nzh], x-major
        tr = np.fft.fft(tr, axis=1)
        tr = pp.wave_scale(tr, box, p, y0=y0, nyl=nyl)                         # the k-space operator on this rank's y rows
        tr = np.fft.ifft(tr, axis=1) * Nx                                       # unnormalised inverse
        back = np.empty_like(pack)
        send = tr.reshape(3, world, nxl, nyl, Nzh)
        for c in range(3):
            a = torch.from_numpy(np.ascontiguousarray(send[c])); b = torch.empty_like(a)
            dist.all_to_all_single(b, a)
            back[c] = b.numpy()
        spec_back = back.transpose(0, 2, 1, 3, 4).reshape(3, nxl, Ny, Nzh)
        ug = np.fft.irfft2(spec_back, s=(Ny, Nz), axes=(2, 3)) * (Ny * Nz)
        full = np.fft.irfftn(pp.wave_scale(np.fft.rfftn(real, axes=(1, 2, 3)), box, p), s=p["grid"], axes=(1, 2, 3), norm="forward")
        err = np.abs(ug - full[:, x0:x0 + nxl]).max() / np.abs(full).max()
        assert err < 1e-12, err

        # gather ownership: the rank whose slab holds the particle's own node plane; its support then lies inside the
        # rank's stored planes [x0 - below, x0 + nxl + above)
        below, above = halo_planes(p["P"])
        f = pp.fractional(pos, box)
        plane = np.floor(f[:, 0] * Nx).astype(int)
        owned = ((plane - x0) % Nx) < nxl
        cnt = torch.tensor([int(owned.sum())]); dist.all_reduce(cnt)
        assert int(cnt) == n
        idx, dlt = pp._support(pos, box, p)
        first = (idx[0][owned, 0] - (x0 - below)) % Nx
        assert np.all(first + p["P"] <= nxl + below + above)
        # cell slabs tile the cell layers
        cs = cell_slabs(9, world)
        assert cs[0][0] == 0 and cs[-1][1] == 8 and all(cs[i][1] == cs[i + 1][0] for i in range(world - 1))
        out.put((rank, "ok"))
    except Exception as e:   # noqa: BLE001
        out.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_slab_transpose_pipeline_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_slab_plan_and_cell_slabs():
    from pse_amd.sharded import cell_slabs, halo_planes, slab_plan
    pl = slab_plan((256, 256, 256), 8)
    assert [p["x0"] for p in pl] == list(range(0, 256, 32)) and all(p["nxl"] == 32 and p["nyl"] == 32 for p in pl)
    with pytest.raises(ValueError):
        slab_plan((250, 256, 256), 8)
    assert halo_planes(6) == (2, 3) and halo_planes(5) == (2, 3) and halo_planes(4) == (1, 2) and halo_planes(8) == (3, 4)
    assert cell_slabs(58, 8) == [(7 * r, 7 * r + 7) for r in range(8)]
    with pytest.raises(ValueError):
        cell_slabs(5, 8)


def _transport_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from pse_amd.sharded import TorchTransport
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tr = TorchTransport(dist)
        peer = 1 - rank
        # two transfers to the same peer in one exchange (the ghost exchange of a two-rank team): they must arrive in list order
        a, b = np.full(5, 10.0 + rank), np.full(3, 20.0 + rank)
        ra, rb = np.zeros(5), np.zeros(3)
        tr.exchange([(a, peer, ra, peer), (b, peer, rb, peer), (None, -1, None, -1)])
        assert np.all(ra == 10.0 + peer) and np.all(rb == 20.0 + peer)
        # a send without a receive and a receive without a send
        if rank == 0:
            tr.exchange([(np.arange(4.0), 1, None, -1)])
        else:
            r = np.zeros(4); tr.exchange([(None, -1, r, 0)]); assert np.all(r == np.arange(4.0))
        s = np.array([1.0 + rank, 2.0, -3.0 * rank])
        tr.allreduce_sum(s)
        assert np.allclose(s, [3.0, 4.0, -3.0])
        out.put((rank, "ok"))
    except Exception as e:   # noqa: BLE001
        out.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_host_staged_transport_over_gloo():
    """pse_amd.sharded.TorchTransport: the Python half of the team's third transport (include/pse_amd.h pse_transport)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_transport_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


# ---- round 4: host logic of the two-step Lanczos and of the per-slab counting (mirrors in pse_amd/sharded.py) ---------------------
@pytest.mark.parametrize("ncx,world,depth", [(56, 8, 2), (8, 2, 2), (9, 3, 2), (8, 4, 2), (12, 4, 1), (6, 2, 1), (24, 8, 2), (5, 1, 2)])
def test_per_slab_counting_gives_global_offsets_of_kept_cells(ncx, world, depth):
    """A rank counts the particles of layers it does not keep per slab, on the first unkept layer of that slab: the prefix sums of
    every kept cell and of every slab boundary must equal those of the full per-cell count (csrc: prepare() + k_cell_keys)."""
    from pse_amd.sharded import kept_layers, slab_book
    rng = np.random.default_rng(ncx * 100 + world)
    cells_per_layer = 7
    ncell = ncx * cells_per_layer
    cell_of = rng.integers(0, ncell, 5000)
    full = np.bincount(cell_of, minlength=ncell)
    off_full = np.concatenate([[0], np.cumsum(full)])
    per = ncx // world
    for rank in range(world):
        kept = set(kept_layers(ncx, world, rank, depth))
        book = slab_book(ncx, world, rank, depth)
        cnt = np.zeros(ncell, dtype=np.int64)
        for c in cell_of:
            layer = c // cells_per_layer
            if layer in kept:
                cnt[c] += 1
            else:
                b = book[layer // per]
                assert b is not None and b not in kept
                cnt[b * cells_per_layer + int(rng.integers(0, cells_per_layer))] += 1     # spread over the cells of that layer
        off = np.concatenate([[0], np.cumsum(cnt)])
        for layer in kept:                                            # every kept cell: begin and end
            lo, hi = layer * cells_per_layer, (layer + 1) * cells_per_layer
            assert np.array_equal(off[lo:hi + 1], off_full[lo:hi + 1]), (rank, layer)
        for q in range(world + 1):                                    # every slab boundary
            assert off[q * per * cells_per_layer] == off_full[q * per * cells_per_layer]


def test_row_map_layout():
    from pse_amd.sharded import row_map
    bases, total = row_map([(1000, 1700), (5000, 5010), (0, 300)])
    assert bases == [0, 768, 1024] and total == 1024 + 512
    assert row_map([(0, 1_000_000)]) == ([0], 1_000_192)
    assert row_map([(10, 10), (20, 21)]) == ([0, 0], 256)             # an empty own range takes no list rows


def test_two_step_block_reproduces_lanczos():
    """The closed-form scalars of a two-iteration block (Gram sums of {p, q, u, w1, w2}) against the textbook three-term recurrence on
    a random symmetric positive definite matrix -- including the first block with an unnormalised start vector."""
    from pse_amd.sharded import two_step_block
    rng = np.random.default_rng(3)
    n = 60
    A = rng.normal(size=(n, n)); M = A @ A.T / n + 0.3 * np.eye(n)
    psi = rng.normal(size=n)
    # reference: plain Lanczos
    V = [psi / np.linalg.norm(psi)]; al, be = [], [0.0]
    for j in range(8):
        w = M @ V[j] - (be[j] * V[j - 1] if j else 0.0)
        al.append(V[j] @ w); w = w - al[j] * V[j]
        be.append(np.linalg.norm(w)); V.append(w / be[-1])
    # two-step blocks
    q, p, u = psi.copy(), np.zeros(n), np.zeros(n)
    alpha_prev, beta, uu = 0.0, 0.0, 0.0
    got_a, got_b = [], [0.0]
    for blk in range(4):
        w1 = M @ q; w2 = M @ w1
        G = dict(qw1=q @ w1, w1w1=w1 @ w1, w1w2=w1 @ w2, w2w2=w2 @ w2, pw1=p @ w1, pw2=p @ w2, uw2=u @ w2, qq=q @ q)
        a0, b1, a1, b2, zz = two_step_block(G, alpha_prev, beta, uu, first=blk == 0)
        sc = 1.0 / np.sqrt(G["qq"])
        v1 = (sc * w1 - a0 * sc * q - beta * p) / b1
        z1 = (sc * w2 - a0 * sc * w1 - beta * u) / b1
        v2 = (z1 - a1 * v1 - b1 * sc * q) / b2
        got_a += [a0, a1]; got_b += [b1, b2]
        p, q, u, alpha_prev, beta, uu = v1, v2, z1, a1, b2, zz
    assert np.allclose(got_a, al, rtol=1e-9) and np.allclose(got_b[1:], be[1:], rtol=1e-9)
    assert abs(abs(q @ V[8]) - 1.0) < 1e-9                            # v_8 up to sign


def test_owned_particle_layout_helpers():
    """Host-side mirror of the owned-particle decomposition (pse_amd/sharded.py <-> LocalGeom in csrc/pse_local.h): cell layers along x,
    the owner of a particle, the rows a rank needs.  Pure NumPy: runs without a GPU."""
    import numpy as np
    from pse_amd.sharded import host_layers, local_capacity, owner_of, x_layer
    L, world = 347.29, 8
    box = (L, L, L, 0.2)
    layers = host_layers(box, world, xi=0.441, error=1e-3, grid=(256, 256, 256))
    assert layers % world == 0 and layers // world >= 3            # what pse_create asks of an owned-particle rank
    rng = np.random.default_rng(0)
    pos = (rng.uniform(size=(20000, 3)) - 0.5) * L
    lay = x_layer(pos, box, layers)
    assert lay.min() >= 0 and lay.max() == layers - 1
    # the layer follows the FRACTIONAL x coordinate: a shift along the tilted lattice vector a2 = (xy Ly, Ly, 0) changes nothing
    shifted = pos + np.array([box[3] * L, L, 0.0])
    assert np.array_equal(x_layer(shifted, box, layers), lay)
    own = owner_of(pos, box, layers, world)
    assert np.array_equal(own, lay // (layers // world)) and set(own) == set(range(world))
    # uniform density: every rank within a few per cent of N / world, and the capacity rule leaves room for the ghosts
    counts = np.bincount(own, minlength=world)
    assert counts.max() < 1.1 * len(pos) / world
    per = layers // world
    cap = local_capacity(len(pos), world, per)
    assert cap > counts.max() * (per + 4) / per
    # two ranks need four layers each (both ghost zones come from the same neighbour)
    assert host_layers((60.0, 60.0, 60.0, 0.0), 2, xi=0.5, error=1e-3) % 2 == 0
    # the slabs are co-moving with the strain: an affinely advected particle keeps its layer when the tilt follows ...
    from pse_amd.sharded import tilt_flipped
    d_xy = 0.013
    adv = pos + np.array([1.0, 0.0, 0.0]) * (d_xy * pos[:, 1:2])
    box2 = (L, L, L, box[3] + d_xy)
    assert not tilt_flipped(box, box2)
    assert (x_layer(adv, box2, layers) != lay).mean() < 1e-3          # (rounding at layer faces only)
    # ... but a Lees-Edwards flip (xy -> xy - 1: b becomes b - a) re-maps fractional x by y / L: most particles change layer,
    # many by more than a slab -- the owner of the particle data redistributes (the step's one-neighbour migration cannot follow)
    box3 = (L, L, L, 0.5 - 1.0)
    assert tilt_flipped((L, L, L, 0.5), box3)
    lay_a, lay_b = x_layer(pos, (L, L, L, 0.5), layers), x_layer(pos, box3, layers)
    expect = (lay_a + np.floor(pos[:, 1] / L * layers).astype(np.int64)) % layers          # x - (xy - 1) y = (x - xy y) + y
    assert (np.abs(((lay_b - expect + layers // 2) % layers) - layers // 2) <= 1).all()
    hop = np.abs(((lay_b - lay_a + layers // 2) % layers) - layers // 2)
    assert (hop > layers // world).mean() > 0.5


def test_two_step_exchange_message_sizes():
    """Sizes of the fixed messages of an owned-particle step, from the capacities alone (no count reaches a host): the first
    exchange carries 10 doubles per record, a Lanczos block two vectors of c_g rows per neighbour."""
    from pse_amd.sharded import local_capacity
    n, world, per, depth = 1_000_000, 8, 6, 2
    cap = local_capacity(n, world, per)
    c_g = int(cap * depth / (per + 2 * depth)) // 256 * 256
    c_own = (cap - 2 * c_g) // 256 * 256
    assert c_g >= 256 and c_own >= c_g and c_own + 2 * c_g <= cap
    first = (4 + 10 * c_g) * 8
    block = 2 * c_g * 32
    assert 3.5e6 < first < 6e6 and 2.5e6 < block < 4.5e6              # a few MB per neighbour: link time, not latency, bounds them


def _lanczos_count_worker(rank, world, port, out):
    """Every rank feeds its own LanczosCount the numbers the ranks agree on after a step (an all-reduce here; on the device every rank
    derives them from the same sums): the policy objects of all ranks stay identical -- same starting count, same decision whether a
    step queues its gated extra block (it decides how many exchanges a step has: a rank that differed would hang the team)."""
    import os
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    try:
        import torch
        import torch.distributed as dist
        from pse_amd.sharded import LanczosCount
        dist.init_process_group("gloo", rank=rank, world_size=world)

        class FakeTeam:
            def __init__(self): self.calls = []
            def set_lanczos_extra(self, k): self.calls.append(k)

        team = FakeTeam()
        lc = LanczosCount(team, m=2, settle=3)
        true_m = [7, 7, 7, 7, 7, 7, 8, 8, 8, 8, 8, 7, 7, 7, 7]          # what the suspension needs, step by step
        trace = []
        for need in true_m:
            start, extras = lc.m, 0 if lc.extras_off else 2
            # what a queue-only step reports: converged at `need` if its queue reached that far, else status 1 at the last size it had
            reached = start + extras
            m_step, status = (max(need, start), 0) if reached >= need else (reached, 1)
            # a rank whose host mirror lags would report the previous step's numbers: the agreed values are the maxima over the ranks
            t = torch.tensor([float(m_step), float(status)])
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            lc.seen(int(t[0]), int(t[1]))
            trace.append((start, extras, m_step, status))
        got = [None] * world
        dist.all_gather_object(got, (trace, team.calls, lc.m, lc.extras_off))
        assert all(g == got[0] for g in got), got
        # the walk: counts 2, 4, 6 fall short (status 1: + 2 each), 8 settles ... at 7 it does not come down by itself (the decision only
        # ever walks upward from the starting count, as the reference's loop does), the rise to 8 is caught by the gated block or by a status 1
        assert trace[0] == (2, 2, 4, 1) and trace[1] == (6, 2, 7, 0) and lc.m >= 7 and any(e == 0 for _, e, _, _ in trace)
        assert all(st == 0 for (_, _, _, st) in trace[2:6])
        assert 0 in team.calls                                              # the extras were switched off in the steady state ...
        k8 = true_m.index(8)
        assert trace[k8][3] == 1 or trace[k8][1] == 2                       # ... and the rise of the needed count was either covered by them or reported
        out.put((rank, "ok"))
    except Exception as e:   # noqa: BLE001
        import traceback
        out.put((rank, "".join(traceback.format_exception(type(e), e, e.__traceback__))[-1500:]))
    finally:
        try:
            dist.destroy_process_group()
        except Exception:   # noqa: BLE001
            pass


def test_lanczos_count_policy_stays_identical_across_ranks():
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_lanczos_count_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=30)
    assert res == [(0, "ok"), (1, "ok")], res
