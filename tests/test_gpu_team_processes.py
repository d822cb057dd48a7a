"""The process-per-rank team driven by REAL processes ("weak" 9): G ranks = G processes on the one-GPU box, every
exchange of the C++ team carried by the host-staged transport over torch.distributed (gloo).  The transfer lists are the ones the
RCCL transport posts (one code path builds them: team_exchange in csrc/pse_capi.hip), so buffers, counts, peers and order of
every all-to-all, halo exchange, ghost-row exchange, Lanczos exchange and row all-gather run between separate address spaces
here -- what RCCL refuses to do on a single device ("Duplicate GPU detected")."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


# The cases of this module are independent of one another (each has its own processes, port and handles) and most of a case's time is
# interpreter and library start-up in its ranks: the first test that asks starts them ALL, four at a time, and every test reads its
# own outcome (tests/test_gpu_switches.py does the same with its child runs).
_JOBS, _OUTCOMES = {}, {}


def _job(key, fn):
    _JOBS[key] = fn


def _outcome(key):
    if not _OUTCOMES:
        from concurrent.futures import ThreadPoolExecutor
        keys = list(_JOBS)

        def run(k):
            try:
                return _JOBS[k]()
            except Exception as e:   # noqa: BLE001
                import traceback
                return ("exception", "".join(traceback.format_exception(type(e), e, e.__traceback__))[-3000:])
        with ThreadPoolExecutor(max_workers=4) as pool:
            for k, out in zip(keys, pool.map(run, keys)):
                _OUTCOMES[k] = out
    return _OUTCOMES[key]


def _spawn_ranks(target, world, args_of):
    """`world` processes running target(*args_of(rank, port, out)); the list of (rank, "ok" | traceback) they put on the queue."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=args_of(r, port, out)) for r in range(world)]
    return _run_ranks(procs, out)


def _worker(rank, world, port, mode, xy, out):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    if mode:
        os.environ["PSE_WAVE_MODE"] = mode
    try:
        import torch
        import torch.distributed as dist
        from conftest import make_suspension, to4
        import pse_amd
        from pse_amd.sharded import ShardedSimulation
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        n, grid = 24_000, 64
        pos, force, box = make_suspension(n, phi=0.1, xy=xy)
        import math
        xi = math.pi * grid / (2.0 * box[0] * math.sqrt(-math.log(1e-3)))       # xi from the fixed grid
        kw = dict(xi=xi, error=1e-3, seed=5, grid=(grid,) * 3)
        sim = ShardedSimulation(n, box, world, rank, transport="host", **kw)
        sim.load(pos, force)
        u_mf = sim.mobility().cpu().numpy()[:, :3].copy()
        _, m = sim.brownian_velocity(1.0, 1e-3, 7, lanczos_m=2)
        u_b = sim.s.vel.cpu().numpy()[:, :3].copy()
        m2 = sim.step(1.0, 1e-3, 8, shear_rate=0.3, lanczos_m=m)
        p_new = sim.s.pos.cpu().numpy()[:, :3].copy()
        # every rank ends with the same arrays
        for a in (u_mf, u_b, p_new):
            t = torch.from_numpy(a.copy()); ref = t.clone()
            dist.broadcast(ref, src=0)
            assert torch.equal(t, ref), "ranks disagree"
        if rank == 0:   # and they are the single-GPU engine's
            eng = pse_amd.Engine(n, box, **kw)
            dpos, dF = to4(pos), to4(force)
            r_mf = eng.mobility(dpos, dF).cpu().numpy()[:, :3]
            vel = to4(np.zeros((n, 3)), 1.0)
            _, mr = eng.brownian_velocity(dpos, dF, 1.0, 1e-3, 7, vel=vel, lanczos_m=2)
            r_b = vel.cpu().numpy()[:, :3].copy()
            accel = torch.zeros((n, 3), dtype=torch.float64, device="cuda"); image = torch.zeros((n, 3), dtype=torch.int32, device="cuda")
            mr2 = eng.step(dpos, vel, accel, image, dF, 1.0, 1e-3, 8, shear_rate=0.3, lanczos_m=mr)
            rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)   # noqa: E731
            assert rel(u_mf, r_mf) < 1e-11, rel(u_mf, r_mf)
            assert m == mr and rel(u_b, r_b) < 1e-9, (m, mr, rel(u_b, r_b))
            assert m2 == mr2 and np.abs(p_new - dpos.cpu().numpy()[:, :3]).max() < 1e-9
        dist.barrier()
        out.put((rank, "ok"))
    except Exception as e:   # noqa: BLE001
        import traceback
        out.put((rank, "".join(traceback.format_exception(type(e), e, e.__traceback__))[-1500:]))
    finally:
        try:
            dist.destroy_process_group()
        except Exception:   # noqa: BLE001
            pass


def _run_ranks(procs, out, deadline_s=600):
    """Start the rank processes and collect one result each.  A rank that dies without reporting fails the test at once (its exit
    code is polled), and whatever happens no rank is left behind blocked in a collective while it holds the GPU."""
    import queue
    import time
    for p in procs:
        p.start()
    res, t_end = [], time.time() + deadline_s
    try:
        while len(res) < len(procs):
            try:
                res.append(out.get(timeout=1.0))
                continue
            except queue.Empty:
                pass
            reported = {r for r, _ in res}
            dead = [i for i, p in enumerate(procs) if p.exitcode not in (None, 0) and i not in reported]
            if dead:
                res += [(i, f"rank process exited with code {procs[i].exitcode} without a result") for i in dead]
                break
            if time.time() > t_end:
                res.append((-1, "timed out waiting for the ranks"))
                break
        return res
    finally:
        for p in procs:
            p.join(timeout=30 if len(res) >= len(procs) else 1)
        for p in procs:
            if p.is_alive():
                p.terminate()
        for p in procs:
            if p.is_alive():
                p.join(timeout=10)
            if p.is_alive():
                p.kill()


_TEAM_CASES = [(2, "", 0.0), (2, "slab", 0.2), (4, "", 0.25), (8, "", 0.0)]
for _c in _TEAM_CASES:
    _job(("team",) + _c, lambda c=_c: _spawn_ranks(_worker, c[0], lambda r, port, out: (r, c[0], port, c[1], c[2], out)))


@pytest.mark.parametrize("world,mode,xy", _TEAM_CASES)
def test_team_of_processes_matches_single_gpu(world, mode, xy):
    res = _outcome(("team", world, mode, xy))
    assert sorted(res) == [(r, "ok") for r in range(world)], res


def _bench_own_ranks():
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PSE_TEAM_LANES")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--transport", "host", "--steps", "2", "--warmup", "1",
                           "--no-cpu", "--n", "100000", "--grid", "128", "--cfg4-n", "120000", "--cfg4-grid", "96", "--cfg4-phi", "0.2"],
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=1200)


_job("bench_own_ranks", _bench_own_ranks)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` as the driver calls it: the parent starts the rank processes itself -- a single-GPU segment, then the
    team in BOTH lane modes (one stream; two lanes + communication stream), each a fresh set of ranks that verifies itself against a
    single-GPU engine before it is timed -- and prints ONE line: the faster verified mode, both recorded, the metric point and a
    config-4 block (shrunk here), the speed-ups against the single GPU of the same run.  On the one-GPU box the ranks share the
    device through the host-staged transport (RCCL needs a GPU per rank)."""
    import json
    r = _outcome("bench_own_ranks")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["value"] > 0 and d["scaling"] == "strong"
    assert ("HOST-STAGED" in d["config"]["parallelism"] and "owned-particle" in d["config"]["parallelism"]) or d["mode"] == "split"
    # both lane modes, each verified and timed by its own set of rank processes; the value is the faster verified one
    assert set(d["modes"]) == {"one_stream", "lanes", "split"} and d["mode"] in d["modes"]     # (two ranks: + the functional split)
    for name, md in d["modes"].items():
        assert md["verify"]["ok"] and md["ms_per_step"] > 0 and md["lanczos_status"] == 0 and (name == "split" or md["device_flags"] == [0]), (name, md)
        assert md["config4"]["ms_per_step"] > 0 and md["config4"]["verify"]["ok"] and md["config4"]["speedup_vs_single"] > 0, (name, md["config4"])
    assert d["ms_per_step"] == min(md["ms_per_step"] for md in d["modes"].values()) == d["modes"][d["mode"]]["ms_per_step"]
    assert ("PSE_TEAM_LANES=1" in d["config"]["parallelism"]) == (d["mode"] == "lanes") and ("FUNCTIONAL split" in d["config"]["parallelism"]) == (d["mode"] == "split")
    assert d["modes"]["lanes"]["lanes_ms"]["side"] > 0 and d["modes"]["one_stream"]["lanes_ms"]["side"] == 0
    # ... against the single GPU of the same run
    assert d["single_gpu"]["ms_per_step"] > 0 and d["single_gpu"]["config4_ms_per_step"] > 0
    assert abs(d["speedup_vs_single"] - d["single_gpu"]["ms_per_step"] / d["ms_per_step"]) < 1e-9
    assert abs(d["config4"]["speedup_vs_single"] - d["single_gpu"]["config4_ms_per_step"] / d["config4"]["ms_per_step"]) < 1e-9
    assert d["north_star"]["speedup_at_this_gpu_count"][1] == d["speedup_vs_single"]
    # the self-diagnosis of a multi-rank line: exchanges per step, device time of every exchange by kind,
    # the spans of both lanes, the critical path (of the owned-particle team: whichever mode the line's value is)
    d = dict(d, **d["modes"]["lanes"]) if d["mode"] == "split" else d
    assert d["exchanges_per_step"] == sum(len(v) for v in d["exchange_us"].values()) >= 5
    assert set(d["exchange_us"]) >= {"migrate_ghosts", "lanczos", "all_to_all", "halo"}
    assert len(d["exchange_us"]["all_to_all"]) == 2 and len(d["exchange_us"]["halo"]) == 1 and len(d["exchange_us"]["migrate_ghosts"]) == 1
    assert all(t > 0 for v in d["exchange_us"].values() for t in v)
    assert d["critical_path_ms"] > 0 and d["lanes_ms"]["main"] > 0
    assert d["lanczos_status"] == 0 and d["particles_owned_sum"] == 100000 and d["device_flags"] == [0]
    assert d["lanczos_extras_off"] is True        # the warm-up reached the steady state: no gated block in the timed steps
    # ... and its verdict on correctness: three untimed steps next to a single-GPU engine on rank 0
    v = d["verify"]
    assert v["ok"] and v["steps"] == 3 and v["max_abs_position_diff_vs_single_gpu"] < 1e-3 and v["images_equal"] and v["lanczos_m_equal"], v
    assert v["particles_that_changed_rank"] > 0, v


def _bench_rccl_refused():
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PSE_TEAM_LANES")}
    env["PSE_BENCH_FALLBACK_ANYWAY"] = "1"      # (ranks sharing the one GPU of the box: never done outside this test)
    env["PSE_BENCH_SEGMENT_TIMEOUT"] = "300"
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu", "--n", "100000",
                           "--grid", "128", "--no-cfg4"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=1200)


_job("bench_rccl_refused", _bench_rccl_refused)


def test_bench_without_a_working_rccl_mode_reports_the_host_transport_and_says_so():
    """`bench.py --gpus 2` with the default transport on a box with ONE GPU: every RCCL segment is refused by its ranks (a communicator needs
    a GPU per rank) -- what a broken RCCL set-up on a real node looks like to the supervisor.  The line is then the one-stream step over the
    host transport, verified against the single GPU, and says that it is the fallback; without the test's variable a box with fewer GPUs than
    ranks gets no line at all (exit code 3)."""
    import json
    import subprocess
    r = _outcome("bench_rccl_refused")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["mode"] == "host_fallback" and "fallback transport" in d["mode_note"] and d["verify"]["ok"] and d["value"] > 0
    assert all("error" in d["modes"][m] for m in ("one_stream", "lanes", "split")) and d["modes"]["host_fallback"]["verify"]["ok"]
    assert "needs one GPU per rank" in r.stderr
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PSE_TEAM_LANES", "PSE_BENCH_FALLBACK_ANYWAY")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu", "--n", "100000", "--grid", "128",
                        "--no-cfg4", "--no-single"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=600)
    assert r.returncode == 3 and not [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]


def _bench_under_launcher():
    import subprocess
    port = _free_port()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PSE_TEAM_LANES")}
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                           "--transport", "host", "--no-cpu", "--particles", "100000", "--grid", "128", "--modes", "lanes", "--no-cfg4"],
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=900)


_job("bench_under_launcher", _bench_under_launcher)


def test_bench_as_ranks_of_torch_distributed_run():
    """The driver's own multi-GPU launch line (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N --steps K --warmup W`): every worker is then the supervisor of its own rank -- it never touches
    the GPU, starts its rank's process for every segment on a port the supervisors agree on through the launcher's store -- and rank 0
    prints the one merged line (one mode and no extra blocks here: the full sequence is test_bench_launches_its_own_ranks)."""
    import json
    r = _outcome("bench_under_launcher")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1                                      # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["value"] > 0 and d["mode"] == "lanes" and set(d["modes"]) == {"lanes"}
    assert d["verify"]["ok"] and d["single_gpu"]["ms_per_step"] > 0 and d["speedup_vs_single"] > 0


def _random_worker(rank, seed, port, out):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    try:
        import torch
        import torch.distributed as dist
        from conftest import to4
        import pse_amd
        from pse_amd.sharded import ShardedSimulation
        from test_gpu_random_teams import config
        c = config(seed)
        world = c["world"]
        if c["mode"]:
            os.environ["PSE_WAVE_MODE"] = c["mode"]
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        kw = dict(xi=c["xi"], error=c["err"], seed=c["seed"], grid=c["grid"])
        sim = ShardedSimulation(c["n"], c["box"], world, rank, transport="host", **kw)
        sim.load(c["pos"], c["force"])
        u_mf = sim.mobility().cpu().numpy()[:, :3].copy()
        _, m = sim.brownian_velocity(0.8, 1e-3, 3 + seed, lanczos_m=2)
        u_b = sim.s.vel.cpu().numpy()[:, :3].copy()
        for a in (u_mf, u_b):
            t = torch.from_numpy(a.copy()); ref = t.clone()
            dist.broadcast(ref, src=0)
            assert torch.equal(t, ref), "ranks disagree"
        if rank == 0:
            eng = pse_amd.Engine(c["n"], c["box"], **kw)
            dpos, dF = to4(c["pos"]), to4(c["force"])
            r_mf = eng.mobility(dpos, dF).cpu().numpy()[:, :3]
            r_b, mr = eng.brownian_velocity(dpos, dF, 0.8, 1e-3, 3 + seed, lanczos_m=2)
            rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)   # noqa: E731
            assert rel(u_mf, r_mf) < 1e-11, rel(u_mf, r_mf)
            assert m == mr and rel(u_b, r_b.cpu().numpy()[:, :3]) < 1e-9, (m, mr)
        dist.barrier()
        out.put((rank, "ok"))
    except Exception as e:   # noqa: BLE001
        import traceback
        out.put((rank, "".join(traceback.format_exception(type(e), e, e.__traceback__))[-1500:]))
    finally:
        try:
            dist.destroy_process_group()
        except Exception:   # noqa: BLE001
            pass


_RANDOM_SEEDS = [5, 7, 16, 18, 20, 22]


def _random_case(seed):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_random_teams import config
    world = config(seed)["world"]
    return world, _spawn_ranks(_random_worker, world, lambda r, port, out: (r, seed, port, out))


for _s in _RANDOM_SEEDS:
    _job(("random", _s), lambda s_=_s: _random_case(s_))


@pytest.mark.parametrize("seed", _RANDOM_SEEDS)
def test_random_team_of_processes(seed):
    """Shapes of tests/test_gpu_random_teams.py (non-cubic, sheared, supports up to 13, odd Nz, grids on the generic far-field
    kernels, 240-point mixed-radix x pass) between real processes: the transfer lists of the process-per-rank branch -- the ones
    RCCL posts -- with halo widths and block sizes the 64^3 cases above do not have."""
    world, res = _outcome(("random", seed))
    assert sorted(res) == [(r, "ok") for r in range(world)], res


def _local_worker(rank, world, port, xy0, n, grid, out):
    """Owned-particle team between real processes (pse_team_step_local over the host-staged transport): 20 sheared steps against the
    single-GPU engine on rank 0 -- ten deterministic ones (positions 1e-9), ten Brownian ones (TRAJ_TOL_BROWNIAN, tests/conftest.py:
    the pair coefficients of the Lanczos mat-vecs are rounded to single-precision accuracy), equal Lanczos counts, particles migrating across every slab face."""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    try:
        import math
        import torch
        import torch.distributed as dist
        from conftest import TRAJ_TOL_BROWNIAN, TRAJ_TOL_DETERMINISTIC, make_suspension, to4
        import pse_amd
        from pse_amd.sharded import LocalShardedSimulation, owner_of
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        pos, force, box = make_suspension(n, phi=0.12, xy=xy0)
        xi = math.pi * grid / (2.0 * box[0] * math.sqrt(-math.log(1e-3)))
        kw = dict(xi=xi, error=1e-3, seed=9, grid=(grid,) * 3)
        sim = LocalShardedSimulation(n, box, world, rank, transport="host", **kw)
        sim.load(pos, force)
        kT, dt, rate = 1.0, 0.25, 0.02
        if rank == 0:
            ref = pse_amd.Engine(n, box, **kw)
            dpos, dF, vel = to4(pos), to4(force), to4(np.zeros((n, 3)), 1.0)
            accel = torch.zeros((n, 3), dtype=torch.float64, device="cuda"); image = torch.zeros((n, 3), dtype=torch.int32, device="cuda")
            _, m0 = ref.brownian_velocity(dpos, dF, kT, dt, 99, vel=to4(np.zeros((n, 3)), 1.0), lanczos_m=2)
        box_m = [m0 if rank == 0 else None]
        dist.broadcast_object_list(box_m, src=0)
        m, xy = box_m[0], xy0
        own0 = owner_of(pos, box, sim.layout["layers"], world)
        crossed = set()
        for k in range(20):
            kTk = 0.0 if k < 10 else kT
            sim.step(kTk, dt, 100 + k, shear_rate=rate, lanczos_m=m)
            tg, p, u, im = sim.gather_local()
            info = sim.engine.info()
            got = [None] * world if rank == 0 else None
            dist.gather_object((tg, p, im, info["lanczos_m"], info["lanczos_status"]), got, dst=0)
            if rank == 0:
                mr = ref.step(dpos, vel, accel, image, dF, kTk, dt, 100 + k, shear_rate=rate, lanczos_m=m)
                P, IM, owner = np.full((n, 3), np.nan), np.zeros((n, 3), dtype=np.int64), np.full(n, -1)
                for r, (t_, p_, im_, m_, st_) in enumerate(got):
                    if kTk > 0:
                        assert st_ == 0 and m_ == mr, (k, r, m_, mr, st_)
                    P[t_] = p_; IM[t_] = im_; owner[t_] = r
                assert (owner >= 0).all()
                tol = TRAJ_TOL_DETERMINISTIC if k < 10 else TRAJ_TOL_BROWNIAN
                assert np.abs(P - dpos.cpu().numpy()[:, :3]).max() < tol, (k, np.abs(P - dpos.cpu().numpy()[:, :3]).max())
                assert np.array_equal(IM, image.cpu().numpy())
                moved = np.nonzero(owner != own0)[0]
                crossed |= {(int(a), int(b)) for a, b in zip(own0[moved], owner[moved])}
                own0 = owner
                if kTk > 0:
                    m = mr
            box_m = [m if rank == 0 else None]
            dist.broadcast_object_list(box_m, src=0)
            m = box_m[0]
            xy += rate * dt
            if xy > 0.5:                  # Lees-Edwards flip: every rank redistributes (LocalShardedSimulation.set_box)
                xy -= 1.0
            sim.set_box(box[0], box[1], box[2], xy)
            if rank == 0:
                ref.set_box(box[0], box[1], box[2], xy)
        if rank == 0:
            faces = {(r, (r + 1) % world) for r in range(world)} | {((r + 1) % world, r) for r in range(world)}
            assert faces <= crossed, sorted(faces - crossed)
        dist.barrier()
        out.put((rank, "ok"))
    except Exception as e:   # noqa: BLE001
        import traceback
        out.put((rank, "".join(traceback.format_exception(type(e), e, e.__traceback__))[-1500:]))
    finally:
        try:
            dist.destroy_process_group()
        except Exception:   # noqa: BLE001
            pass


_LOCAL_CASES = [(2, 0.1, 40_000, 96), (4, -0.15, 40_000, 96), (8, 0.0, 80_000, 128), (3, 0.47, 30_000, 96)]   # (the last: through a tilt flip)
for _c in _LOCAL_CASES:
    _job(("local",) + _c, lambda c=_c: _spawn_ranks(_local_worker, c[0], lambda r, port, out: (r, c[0], port, c[1], c[2], c[3], out)))


@pytest.mark.parametrize("world,xy0,n,grid", _LOCAL_CASES)
def test_owned_particle_team_of_processes_follows_single_gpu(world, xy0, n, grid):
    res = _outcome(("local", world, xy0, n, grid))
    assert sorted(res) == [(r, "ok") for r in range(world)], res


def _split_worker(rank, port, xy0, n, grid, out):
    """The two-rank FUNCTIONAL split (pse_amd.sharded.SplitSimulation: rank 0 the real-space half + Lanczos, rank 1 the wave-space half,
    one all-reduce per step) between two real processes: eight sheared Brownian steps against the single-GPU engine on rank 0."""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    try:
        import math
        import torch
        import torch.distributed as dist
        from conftest import TRAJ_TOL_BROWNIAN, make_suspension, to4
        import pse_amd
        from pse_amd.sharded import SplitSimulation
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=2)
        pos, force, box = make_suspension(n, phi=0.12, xy=xy0)
        xi = math.pi * grid / (2.0 * box[0] * math.sqrt(-math.log(1e-3)))
        kw = dict(xi=xi, error=1e-3, seed=9, grid=(grid,) * 3)
        sim = SplitSimulation(n, box, rank, dist, **kw)
        sim.load(pos, force)
        kT, dt, rate, m, xy = 1.0, 0.05, 0.1, 2, xy0
        if rank == 0:
            ref = pse_amd.Engine(n, box, **kw)
            dpos, dF, vel = to4(pos), to4(force), to4(np.zeros((n, 3)), 1.0)
            accel = torch.zeros((n, 3), dtype=torch.float64, device="cuda"); image = torch.zeros((n, 3), dtype=torch.int32, device="cuda")
            mr = 2
        for k in range(8):
            m = sim.step(kT, dt, 100 + k, shear_rate=rate, lanczos_m=m)
            if rank == 0:
                mr = ref.step(dpos, vel, accel, image, dF, kT, dt, 100 + k, shear_rate=rate, lanczos_m=mr)
                assert m == mr, (k, m, mr)
                assert (sim.pos - dpos).abs().max().item() < (1e-12 if k == 0 else TRAJ_TOL_BROWNIAN), (k, (sim.pos - dpos).abs().max().item())
                assert torch.equal(sim.image, image)
            xy += rate * dt
            sim.set_box(box[0], box[1], box[2], xy)
            if rank == 0:
                ref.set_box(box[0], box[1], box[2], xy)
        # both ranks hold the same particles, bit for bit
        t = sim.pos.cpu(); other = t.clone()
        dist.broadcast(other, src=0)
        assert torch.equal(t, other), "the two ranks' replicas differ"
        dist.barrier()
        out.put((rank, "ok"))
    except Exception as e:   # noqa: BLE001
        import traceback
        out.put((rank, "".join(traceback.format_exception(type(e), e, e.__traceback__))[-1500:]))
    finally:
        try:
            dist.destroy_process_group()
        except Exception:   # noqa: BLE001
            pass


_job("split", lambda: _spawn_ranks(_split_worker, 2, lambda r, port, out: (r, port, 0.1, 40_000, 96, out)))


def test_functional_split_of_two_processes_follows_single_gpu():
    res = _outcome("split")
    assert sorted(res) == [(0, "ok"), (1, "ok")], res
