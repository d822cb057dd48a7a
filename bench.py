#!/usr/bin/env python3
"""bench.py -- the PSE hot path on MI355X, one JSON line (contract in the task statement, section 4).

A "step" is one full PSE Brownian step through the C-ABI (pse_step): sort -> spread -> rocFFT -> k-space scaling with
in-k-space noise -> inverse FFT -> gather -> near-field M.F -> Lanczos M_real^{1/2} psi -> Euler update, on the
synthetic random-sphere suspension of BASELINE.json's metric point (N = 1e6, phi = 0.1, 256^3 grid, fp64), inputs
resident in HBM before the timed region.  value = particle-steps/s summed over all ranks.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--n PARTICLES] [--grid G] [--no-cpu] [--traffic FILE] [--dry-run]
With --gpus N > 1 and no torch.distributed environment the process launches the N ranks itself (one child
`python bench.py ...` per GPU with RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set; the parent never touches the GPU) and relays
rank 0's JSON line; started under torch.distributed.run (WORLD_SIZE set) it is one of the ranks.  See DESIGN.md "Multi-GPU".
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def suspension(n, phi, seed=12345, fseed=54321):
    L = (4.0 * math.pi * n / (3.0 * phi)) ** (1.0 / 3.0)
    pos = (np.random.default_rng(seed).uniform(0.0, 1.0, (n, 3)) - 0.5) * L
    force = np.random.default_rng(fseed).normal(size=(n, 3))
    force -= force.mean(axis=0)
    return pos, force, L


def cpu_baseline(budget_s=12.0):
    """Direct O(N^2) periodic-RPY M.F (the oracle, oracle/pse_oracle.c) on BASELINE config 1, all host cores."""
    from oracle import pse_port as pp
    n, phi, xi = 1000, 0.05, 0.5
    pos, force, L = suspension(n, phi)
    box = (L, L, L, 0.0)
    cores = pp.max_threads()
    pp.mobility_direct(pos, force, box, xi)            # warm-up
    times = []
    t_end = time.time() + budget_s
    while len(times) < 5 or (time.time() < t_end and len(times) < 50):
        t0 = time.perf_counter()
        pp.mobility_direct(pos, force, box, xi)
        times.append(time.perf_counter() - t0)
    t = float(np.median(times))
    return {
        "value": n / t, "unit": "particle-evals/s (M.F only)", "cores": int(cores), "kind": "port", "comparable": False,
        "sample": f"NOT the metric workload (an O(N^2) evaluator cannot run N=1e6; the reference has no CPU path): BASELINE "
                  f"config 1, direct Ewald-summed periodic RPY M.F, N={n}, phi={phi}, xi={xi}, fp64, tol 1e-14, "
                  f"median of {len(times)} evals ({t * 1e3:.1f} ms each); O(N^2): extrapolates to "
                  f"{t * (1e6 / n) ** 2:.3g} s per eval at N=1e6",
        "evals_per_s": 1.0 / t,
    }


def measure_traffic(kernel_prefix, args):
    """HBM bytes per launch of one kernel, measured NOW: two child runs of this same script (a few steps, one stream, no
    extras) under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` -- separate passes, as MI355X_MICROARCH.md prescribes; the
    counters cannot be read from inside a process.  2 x FETCH_SIZE + WRITE_SIZE (the x2 of the guide, calibrated for every access
    shape in use by tools/microbench/nt_fetch.hip).  None if the profiler is not there or a pass fails: the caller falls back to
    the --traffic file and says so."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    env = dict(os.environ, TMPDIR="/tmp", PSE_OVERLAP="0")
    py = os.path.realpath(sys.executable)
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="pse_pmc_", dir="/tmp")
        cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--", py, os.path.abspath(__file__),
               "--steps", "3", "--warmup", "1", "--no-cpu", "--no-ref-grid", "--no-traffic", "--n", str(args.n), "--phi", str(args.phi),
               "--grid", str(args.grid), "--error", str(args.error), "--kT", str(args.kT), "--dt", str(args.dt)]
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None
            v = [float(row["Counter_Value"]) for row in csv.DictReader(open(files[0]))
                 if row["Counter_Name"] == counter and row["Kernel_Name"].replace("void ", "").startswith(kernel_prefix)]
            if not v:
                return None
            vals[counter] = sum(v) / len(v) * 1024.0            # the counters are reported in KiB
        except Exception:   # noqa: BLE001  (a profiler hiccup must not cost the bench line)
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return 2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]


def launch_ranks(args, argv, script=None):
    """--gpus N without a torch.distributed environment: start the N ranks as children (one process per GPU, the environment
    torch.distributed.run would give them: RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR = 127.0.0.1, a free MASTER_PORT) and relay
    rank 0's line.  This process has not initialised HIP (torch is not even imported yet) and never replaces itself: the ranks
    are children, the first non-zero exit code among them is ours, and when one fails the others (exactly those PIDs) are ended."""
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    rest = [a for a in argv if a != "--dry-run"]
    cmd = [sys.executable, script or os.path.abspath(__file__)] + rest
    rank_env = {"WORLD_SIZE": str(args.gpus), "LOCAL_WORLD_SIZE": str(args.gpus), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)}
    if args.dry_run:
        print(json.dumps({"launch": cmd, "env": dict(rank_env, RANK="<r>", LOCAL_RANK="<r>"), "n_gpus": args.gpus}))
        return 0
    env = dict(os.environ, **rank_env)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    procs = []
    for r in range(args.gpus):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True, env=e))
    found = []

    def relay():
        for out in procs[0].stdout:
            out = out.rstrip("\n")
            try:
                obj = json.loads(out)
                if isinstance(obj, dict) and "metric" in obj:      # a bare JSON scalar ("0", "true") is just a line of output
                    found.append(out)
                    continue
            except Exception:   # noqa: BLE001  (whatever rank 0 prints, this thread must keep draining its pipe)
                pass
            print(out, file=sys.stderr)

    reader = threading.Thread(target=relay, daemon=True)      # rank 0 may be the one left waiting when another rank dies
    reader.start()
    rc = 0
    pending = list(procs)
    t_all = time.time() + float(os.environ.get("PSE_BENCH_LAUNCH_TIMEOUT", "3600"))   # the whole run
    t_kill = None                                       # once ranks were told to end: when the survivors are killed
    while pending:
        for p in list(pending):
            code = p.poll()
            if code is None:
                continue
            pending.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in pending:                       # a rank failed: the rest would wait in a collective for ever
                    q.terminate()
                t_kill = time.time() + 20.0
        now = time.time()
        if pending and t_kill is None and now > t_all:
            rc = rc or 124
            print(f"bench.py: the ranks did not finish within PSE_BENCH_LAUNCH_TIMEOUT: ending them", file=sys.stderr)
            for q in pending:
                q.terminate()
            t_kill = now + 20.0
        elif pending and t_kill is not None and now > t_kill:
            for q in pending:                           # SIGTERM was not enough (stuck in a driver or RCCL call): exactly these PIDs
                q.kill()
            t_kill = now + 20.0
        if pending:
            time.sleep(0.05)
    reader.join(timeout=30)
    line = found[-1] if found else None
    if rc != 0:
        print(f"bench.py: the {args.gpus}-rank run failed (exit code {rc})", file=sys.stderr)
        return rc
    if line is None:
        print("bench.py: rank 0 printed no result line", file=sys.stderr)
        return 1
    print(line)
    return 0


def run_owned_particle_team(args, world, rank, host_transport, dist, torch):
    """--gpus N > 1 (default): every rank owns the particles of its x slab (pse_team_step_local): migration + ghosts in one exchange
    of fixed-size messages, the whole step queue-only.  The JSON line carries what a first run on a multi-GPU node needs to be
    read: exchanges per step, device time of every exchange by kind, the spans of both lanes, the critical path."""
    from pse_amd.sharded import LocalShardedSimulation
    n, grid = args.n, args.grid
    pos, force, L = suspension(n, args.phi)
    box = (L, L, L, 0.0)
    xi = math.pi * grid / (2.0 * L * math.sqrt(-math.log(args.error)))      # xi from the fixed grid
    sim = LocalShardedSimulation(n, box, world, rank, transport="host" if host_transport else "rccl", xi=xi, error=args.error, seed=1,
                                 grid=(grid,) * 3)
    sim.load(pos, force, mass=1.0)
    # Before anything is timed: a few steps of the team next to the single-GPU engine on rank 0 (same suspension, same noise) -- the
    # first run on a multi-GPU node then says whether the exchanges moved the right bytes, not only how long they took.
    verify = None
    if not args.no_verify:
        verify = verify_team_against_single_gpu(sim, pos, force, box, dict(xi=xi, error=args.error, seed=1, grid=(grid,) * 3), args, world, rank,
                                                dist, torch)
        sim.load(pos, force, mass=1.0)

    def barrier():
        dist.barrier()
        torch.cuda.synchronize()

    def agree(v):     # the starting count of the next step: the largest any rank reports (they all take the same decisions)
        t = torch.tensor([float(v)], dtype=torch.float64, device="cpu" if host_transport else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return int(t[0])

    # warm-up: the starting count of the Lanczos iteration grows until a step converges inside its queue (a queue-only step never
    # waits for more iterations; pse_info.lanczos_status says when its queue was too short)
    m, status = 2, 1
    for it in range(max(args.warmup, 1)):
        sim.step(args.kT, args.dt, it, lanczos_m=m)
        torch.cuda.synchronize()
        i = sim.engine.info()
        status = agree(i["lanczos_status"])
        m = agree(max(i["lanczos_m"], 2) + (2 if i["lanczos_status"] == 1 else 0))
        if it >= args.warmup - 1 and status == 0:
            break
    # deterministic M.F (kT = 0, no update)
    barrier()
    n_mf = max(3, args.steps // 2)
    t0 = time.perf_counter()
    for it in range(n_mf):
        sim.step(0.0, args.dt, 0, integrate=False)
    barrier()
    t_mf = (time.perf_counter() - t0) / n_mf
    # headline: EXACTLY --steps steps between two barriers; nothing is read back inside (the step only queues work)
    barrier()
    t0 = time.perf_counter()
    for it in range(args.steps):
        sim.step(args.kT, args.dt, args.warmup + it, lanczos_m=m)
    barrier()
    elapsed = time.perf_counter() - t0
    i = sim.engine.info()
    status = agree(i["lanczos_status"])
    # a few more steps with every exchange bracketed by events
    sim.team.set_diag(True)
    diags = []
    for it in range(5):
        sim.step(args.kT, args.dt, args.warmup + args.steps + it, lanczos_m=m)
        diags.append(sim.team.diag())
    sim.team.set_diag(False)
    flags = sim.team.local_status()
    t = torch.tensor([elapsed, t_mf], dtype=torch.float64, device="cpu" if host_transport else "cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, t_mf = float(t[0]), float(t[1])
    n_loc = torch.tensor([float(sim.s.n_local.item())], dtype=torch.float64, device="cpu" if host_transport else "cuda")
    dist.all_reduce(n_loc, op=dist.ReduceOp.SUM)
    if rank != 0:
        return
    info = sim.engine.info()
    d = diags[-1]
    med = lambda key: {k: [round(float(np.median([x[key][k][j] for x in diags])), 2) for j in range(len(d[key][k]))] for k in d[key]}   # noqa: E731
    t_step = elapsed / args.steps
    transport = ("HOST-STAGED transport (torch.distributed gloo, ranks may share a GPU): a functional run of the process-per-rank driver, "
                 "not a scaling number" if host_transport else "RCCL over xGMI")
    lay = sim.layout
    out = {
        "metric": "BD particle-steps/s (full PSE Brownian step: M.F + k-space noise + Lanczos M^1/2.psi + Euler), N=1e6, phi=0.1",
        "value": n / t_step, "unit": "particle-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": t_step * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"random-sphere suspension N={n}, phi={args.phi}, cubic L={L:.2f}, grid {grid}^3, xi={xi:.4f}, "
                               f"rcut={info['rcut']:.3f}, P={info['P']}, error={args.error}, kT={args.kT}, dt={args.dt}",
                   "parallelism": f"{world} ranks, {transport}; owned-particle decomposition (pse_team_step_local): {lay['layers_per_rank']} of "
                                  f"{lay['layers']} cell layers along x per rank + 2 ghost layers per side, far-field grid in {world} x-slabs "
                                  f"(2 all-to-alls + 1 plane halo per step on the far-field lane), two Lanczos iterations per exchange, "
                                  f"row capacity {sim.engine.params.n_max} per rank; the force provider of the bench re-gathers its fixed forces "
                                  f"by tag after every step"},
        "steps_per_s": 1.0 / t_step, "mf_evals_per_s": 1.0 / t_mf, "lanczos_m": info["lanczos_m"], "lanczos_status": status,
        "lanczos_exchanges": info["lanczos_exchanges"], "particles_owned_sum": int(n_loc[0]), "device_flags": flags,
        # what the first run on a multi-GPU node needs: per-exchange device time by kind (median of 5 steps), the
        # host time spent issuing each, bytes this rank sent, the spans of the two lanes
        "exchanges_per_step": d["exchanges_per_step"], "exchange_us": med("exchange_us"), "exchange_host_us": med("exchange_host_us"),
        "exchange_bytes": d["exchange_bytes"],
        "lanes_ms": {k: round(float(np.median([x["lanes_ms"][k] for x in diags])), 4) for k in ("main", "side")},
        "critical_path_ms": round(float(np.median([x["critical_path_ms"] for x in diags])), 4),
        "verify": verify,
        "roofline": None, "cpu_baseline": None,
    }
    if verify is not None and not verify["ok"]:
        print(f"bench.py: the team's trajectory differs from the single GPU's: {verify}", file=sys.stderr)
    print(json.dumps(out))


def verify_team_against_single_gpu(sim, pos, force, box, kw, args, world, rank, dist, torch, steps=3):
    """`steps` Brownian steps of the owned-particle team and of a single-GPU engine on rank 0's device, from the same suspension with the
    same noise: largest position difference, images and Lanczos counts.  Untimed; the engine is released before the bench goes on."""
    import pse_amd
    n = len(pos)
    dt = 0.05                                    # large steps: particles cross slab faces within the check
    if rank == 0:
        ref = pse_amd.Engine(n, box, **kw)
        to4 = lambda a, w=0.0: torch.tensor(np.hstack([a, np.full((len(a), 1), w)]), dtype=torch.float64, device="cuda")   # noqa: E731
        dpos, dF, vel = to4(pos), to4(force), to4(np.zeros((n, 3)), 1.0)
        accel = torch.zeros((n, 3), dtype=torch.float64, device="cuda"); image = torch.zeros((n, 3), dtype=torch.int32, device="cuda")
        _, m0 = ref.brownian_velocity(dpos, dF, args.kT, dt, 999, vel=to4(np.zeros((n, 3)), 1.0), lanczos_m=2)
    box_m = [m0 if rank == 0 else None]
    dist.broadcast_object_list(box_m, src=0)
    m = box_m[0]
    worst, images_equal, m_equal, status_ok, migrated = 0.0, True, True, True, 0
    own0 = None
    for k in range(steps):
        sim.step(args.kT, dt, 1000 + k, lanczos_m=m)
        tg, p, _, im = sim.gather_local()
        info = sim.engine.info()
        got = [None] * world if rank == 0 else None
        dist.gather_object((tg, p, im, info["lanczos_m"], info["lanczos_status"]), got, dst=0)
        if rank == 0:
            mr = ref.step(dpos, vel, accel, image, dF, args.kT, dt, 1000 + k, lanczos_m=m)
            P, IM, owner = np.full((n, 3), np.nan), np.zeros((n, 3), dtype=np.int64), np.full(n, -1)
            for r, (t_, p_, im_, m_, st_) in enumerate(got):
                P[t_] = p_; IM[t_] = im_; owner[t_] = r
                m_equal = m_equal and m_ == mr
                status_ok = status_ok and st_ == 0
            diff = np.abs(P - dpos.cpu().numpy()[:, :3])
            worst = float("nan") if (owner < 0).any() else max(worst, float(diff.max()))
            images_equal = images_equal and bool(np.array_equal(IM, image.cpu().numpy()))
            if own0 is not None:
                migrated += int((owner != own0).sum())
            own0 = owner
            m = mr
        box_m = [m if rank == 0 else None]
        dist.broadcast_object_list(box_m, src=0)
        m = box_m[0]
    if rank != 0:
        return None
    del ref
    torch.cuda.empty_cache()
    return {"steps": steps, "dt": dt, "max_abs_position_diff_vs_single_gpu": worst, "images_equal": images_equal, "lanczos_m_equal": m_equal,
            "lanczos_status_zero": status_ok, "particles_that_changed_rank": migrated,
            # (1e-7, not 1e-9: the pair coefficients of the Lanczos mat-vecs are single precision, and a coefficient that rounds the other way
            # in one of the two runs moves a particle by ~1e-8 dt -- tests/conftest.py TRAJ_TOL_BROWNIAN has the mechanism)
            "ok": bool(worst < 1e-7 and images_equal and m_equal and status_ok)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50, help="timed steps (BASELINE.md section 4: >= 50)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--n", "--particles", dest="n", type=int, default=1_000_000,
                    help="(--particles under torch.distributed.run, whose parser reads a bare --n as an abbreviation of its own options)")
    ap.add_argument("--phi", type=float, default=0.1)
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--error", type=float, default=1e-3)
    ap.add_argument("--kT", type=float, default=1.0)
    ap.add_argument("--dt", type=float, default=1e-3)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--traffic", default=os.path.join(ROOT, "profiles", "traffic.json"),
                    help="HBM bytes per kernel from separate rocprofv3 --pmc passes of this command (tools/round_profile.sh)")
    ap.add_argument("--no-traffic", action="store_true", help="do not measure the dominant kernel's HBM bytes with rocprofv3 child passes")
    ap.add_argument("--no-ref-grid", action="store_true", help="skip the extra steps on the reference rule's 360^3 grid")
    ap.add_argument("--transport", choices=["rccl", "host"], default="rccl",
                    help="multi-rank runs: RCCL over xGMI (one GPU per rank), or the host-staged transport over gloo (ranks may "
                         "share a GPU: exercises the process-per-rank driver on a one-GPU box; never the headline)")
    ap.add_argument("--replicated", action="store_true",
                    help="multi-rank runs: the replicated-state team calls (every rank passes all N particles) instead of the "
                         "owned-particle step (pse_team_step_local), which is the default")
    ap.add_argument("--no-verify", action="store_true",
                    help="multi-rank runs: skip the three untimed steps next to a single-GPU engine on rank 0 (the `verify` object of the line)")
    ap.add_argument("--dry-run", action="store_true", help="with --gpus N > 1: print the launch command and stop")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args, sys.argv[1:]))
    if args.dry_run:
        print(json.dumps({"launch": None, "n_gpus": args.gpus}))
        return

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the torch.distributed environment has WORLD_SIZE={world}")
    host_transport = args.transport == "host"
    n_dev = torch.cuda.device_count()
    if n_dev == 0:
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    if not host_transport and local_rank >= n_dev:
        raise SystemExit(f"--gpus {args.gpus} over RCCL needs one GPU per rank and this node has {n_dev}; "
                         "--transport host lets ranks share a device (never the headline)")
    device = local_rank % n_dev if host_transport else local_rank
    torch.cuda.set_device(device)
    use_dist = world > 1 or "PSE_FORCE_SHARDED" in os.environ
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if host_transport:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))

    import pse_amd
    from pse_amd import distributed as pdist

    if world > 1 and not args.replicated:
        run_owned_particle_team(args, world, rank, host_transport, dist, torch)
        dist.destroy_process_group()
        return
    n, grid = args.n, args.grid
    pos, force, L = suspension(n, args.phi)
    xi = math.pi * grid / (2.0 * L * math.sqrt(-math.log(args.error)))      # xi from the fixed grid
    sim = pdist.make_simulation(n, (L, L, L, 0.0), xi=xi, error=args.error, seed=1, grid=(grid,) * 3,
                                world=world, rank=rank, **({"transport": "host"} if host_transport and use_dist else {}))
    sim.load(pos, force, mass=1.0)
    info = sim.info()

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    m = 2
    for it in range(args.warmup):
        m = sim.step(args.kT, args.dt, it, lanczos_m=m)
    # M.F evals/s (deterministic part only), a few evaluations
    barrier()
    t0 = time.perf_counter()
    n_mf = max(3, args.steps // 2)
    for it in range(n_mf):
        sim.mobility()
    barrier()
    t_mf = (time.perf_counter() - t0) / n_mf
    # the same evaluation with MOVING particles: every call sees positions displaced by more than r_buff / 2 from the last one, so
    # the kept neighbour list is never reused and every call sorts and walks the cells (single GPU; slab ranks keep no list)
    t_mf_moving = None
    if world == 1 and hasattr(sim, "engine") and hasattr(sim, "pos"):
        rng = np.random.default_rng(99)
        moved = []
        for k in range(4):
            p = sim.pos.clone()
            p[:, :3] += torch.tensor(rng.uniform(-0.5, 0.5, (n, 3)), dtype=torch.float64, device="cuda")
            moved.append(p)
        for k in range(2):
            sim.engine.mobility(moved[k], sim.force, vel=sim.vel)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(n_mf):
            sim.engine.mobility(moved[it % 4], sim.force, vel=sim.vel)
        torch.cuda.synchronize()
        t_mf_moving = (time.perf_counter() - t0) / n_mf
        del moved

    # headline: EXACTLY --steps steps, nothing else inside the timed region (no phase timing, no host synchronisation beyond
    # what the step itself needs: the Lanczos convergence check)
    ms = []
    barrier()
    t0 = time.perf_counter()
    for it in range(args.steps):
        m = sim.step(args.kT, args.dt, args.warmup + it, lanczos_m=m)
        ms.append(m)
    barrier()
    elapsed = time.perf_counter() - t0
    # distribution of single steps (median, p10, p90), each bracketed by device events
    n_pct = max(10, min(50, args.steps))
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_pct)]
    for it in range(n_pct):
        ev[it][0].record()
        m = sim.step(args.kT, args.dt, args.warmup + args.steps + it, lanczos_m=m)
        ev[it][1].record()
    torch.cuda.synchronize()
    per_step = sorted(a.elapsed_time(b) for a, b in ev)
    pct = {"p10": per_step[int(0.1 * (n_pct - 1))], "p50": per_step[n_pct // 2], "p90": per_step[int(round(0.9 * (n_pct - 1)))],
           "n": n_pct}
    # per-phase device times from a separate loop (the library records hipEvents on its own stream and synchronises after
    # every call to read them, so this loop is never the headline)
    sim.set_timing(True)
    phase_sum = {}
    n_ph = max(5, min(20, args.steps))
    for it in range(n_ph):
        m = sim.step(args.kT, args.dt, args.warmup + args.steps + n_pct + it, lanczos_m=m)
        for k, v in sim.phase_times().items():
            phase_sum[k] = phase_sum.get(k, 0.0) + v
    sim.set_timing(False)
    if world > 1:
        t = torch.tensor([elapsed, t_mf], dtype=torch.float64, device="cpu" if host_transport else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, t_mf = float(t[0]), float(t[1])
    info = sim.info()
    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return

    t_step = elapsed / args.steps
    ng = grid ** 3
    nloc, ngloc = n / world, ng / world
    phases = {k: v / n_ph for k, v in phase_sum.items()}                     # ms per step, rank 0
    m_avg = float(np.mean(ms))
    # algorithmic bytes per launch (BASELINE.md section 3 / SURVEY.md 8d), per rank
    alg = {
        "t_spread": 64 * nloc + 24 * ngloc, "t_fft_fwd": 48 * ngloc, "t_scale": 48 * ngloc, "t_fft_inv": 48 * ngloc,
        "t_gather": 24 * ngloc + 64 * nloc, "t_real": 96 * nloc, "t_matvec": 96 * nloc,
    }
    per_launch_ms = {k: phases.get(k, 0.0) for k in alg}
    # share of the step: the pair-list mat-vec runs once per Lanczos iteration except the first, whose M.psi is delivered by
    # the near-field pass that builds the list
    weight = dict(per_launch_ms)
    weight["t_matvec"] = per_launch_ms["t_matvec"] * max(1, info["lanczos_matvecs"] - 1)
    dom = max(weight, key=weight.get)
    names = {"t_spread": "k_spread_tiles (spread, incl. binning + records)", "t_fft_fwd": "rocFFT 2-D R2C x3",
             "t_scale": "k_xfft_scale (x FFT + k-space scale/noise + inverse x FFT)",
             "t_fft_inv": "rocFFT 2-D C2R x3", "t_gather": "k_gather_bins (gather)",
             "t_real": "k_mreal_cells (near-field M_real.F from the cell list, writes the pair list)",
             "t_matvec": "k_mreal_list (near-field mat-vec from the pair list, once per Lanczos iteration after the first)"}
    pmc_names = {"t_spread": "pse::k_spread_tiles", "t_scale": "pse::k_xfft_scale", "t_gather": "pse::k_gather_bins",
                 "t_real": "pse::k_mreal_cells<true", "t_matvec": "pse::k_mreal_list"}
    ach = alg[dom] / (per_launch_ms[dom] * 1e-3) / 1e9 if per_launch_ms[dom] > 0 else 0.0
    # HBM bytes of the dominant kernel: rocprofv3 counters cannot be collected from inside this process, so the figure comes
    # from counter passes of this same command kept in a file (--traffic, default profiles/traffic.json: separate --pmc
    # FETCH_SIZE and WRITE_SIZE runs, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16-byte-per-lane reads -- the
    # factor is calibrated for this kernel's non-temporal list loads by tools/microbench/nt_fetch) and carries the commit the
    # passes were taken at, so a stale file shows
    traffic, traffic_src = None, None
    under_profiler = any("rocprof" in os.environ.get(k, "") for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB"))
    if world == 1 and not args.no_traffic and not under_profiler and dom in pmc_names:      # no profiler inside a profiled run
        traffic = measure_traffic(pmc_names[dom], args)       # child processes with their own engine (this one idles meanwhile)
        if traffic is not None:
            traffic_src = ("measured in this run: 2 x FETCH_SIZE + WRITE_SIZE of that kernel, per launch, from two rocprofv3 --pmc child "
                           "passes of this command (3 steps, one stream)")
    tr_file = args.traffic
    if traffic is None and tr_file and os.path.exists(tr_file) and world == 1 and n == 1_000_000 and grid == 256:
        try:
            tj = json.load(open(tr_file))
            for k, v in tj.items():
                if k.startswith(pmc_names.get(dom, "?")):
                    traffic = 2 * v["fetch_raw"] + v["write"]
                    traffic_src = (f"{os.path.relpath(tr_file, ROOT)} ({tj.get('_source', 'separate rocprofv3 --pmc passes')}; "
                                   f"kernels as of commit {tj.get('_commit', 'unrecorded')}); not measured in this run")
        except Exception:
            traffic = None
    # spread + gather: with the binning and the 64-byte records both kernels read (ms), and the two kernels alone (kernels_ms)
    rec_ms = phases.get("t_records", 0.0)
    sg_ms = per_launch_ms["t_spread"] + per_launch_ms["t_gather"] + rec_ms
    sg_kernels_ms = per_launch_ms["t_spread"] + per_launch_ms["t_gather"]
    sg_bytes = alg["t_spread"] + alg["t_gather"]
    a_step = (192.0 * ng + 224.0 * n) + 32.0 * n + (m_avg + 1.0) * 256.0 * n + 32.0 * n * m_avg + 160.0 * n
    # what the kept neighbour list did in this run (the M.F evaluations repeat at fixed positions and reuse it; the Brownian steps
    # at this kT dt outrun r_buff / 2 every step, so every step sorts and walks the cells, as with PSE_SKIN=0)
    nl_note = None
    eng = getattr(sim, "engine", None)
    if eng is not None and hasattr(eng, "neighbor_stats"):
        rb, nb_, nr_ = eng.neighbor_stats()
        nl_note = {"r_buff": rb, "calls_that_built": nb_, "calls_that_reused": nr_,
                   "note": "mf_evals_per_s is measured at fixed positions and runs on the kept list; the steps of the headline do not"}
    out = {
        "metric": "BD particle-steps/s (full PSE Brownian step: M.F + k-space noise + Lanczos M^1/2.psi + Euler), "
                  "N=1e6, phi=0.1",
        "value": n / t_step, "unit": "particle-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": t_step * 1e3, "ms_per_step_percentiles": pct, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"random-sphere suspension N={n}, phi={args.phi}, cubic L={L:.2f}, grid {grid}^3, "
                               f"xi={xi:.4f}, rcut={info['rcut']:.3f}, P={info['P']}, error={args.error}, kT={args.kT}, "
                               f"dt={args.dt}", "parallelism": sim.describe()},
        "steps_per_s": 1.0 / t_step, "mf_evals_per_s": 1.0 / t_mf, "mf_particle_evals_per_s": n / t_mf,
        "mf_evals_per_s_moving": (1.0 / t_mf_moving) if t_mf_moving else None,
        # the whole step against the roofline: SURVEY.md 8(d) A_step = A_MF + 32 N + (m + 1) 256 N + 32 N m + 160 N over all ranks
        "step_roofline": {"algorithmic_bytes": a_step, "ms": t_step * 1e3, "lanczos_m": m_avg,
                          "frac_of_hbm_peak": a_step / t_step / 1e9 / (HBM_PEAK_GBS * world)},
        "lanczos_m": m_avg, "lanczos_matvecs_per_step": info["lanczos_matvecs"],
        "neighbor_list": nl_note,
        "roofline": {"bound": "hbm", "kernel": names[dom], "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": alg[dom], "ms_per_launch": per_launch_ms[dom],
                     # what the kernel really moves, as a rate: how close it runs to the memory system on its own traffic
                     "traffic_rate": (traffic / (per_launch_ms[dom] * 1e-3) / 1e9) if traffic and per_launch_ms[dom] > 0 else None,
                     "traffic_rate_frac_of_peak": (traffic / (per_launch_ms[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic and per_launch_ms[dom] > 0 else None},
        "spread_plus_gather": {"ms": sg_ms, "algorithmic_bytes": sg_bytes,
                               "frac_of_hbm_peak": (sg_bytes / (sg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if sg_ms > 0 else 0.0,
                               "kernels_ms": sg_kernels_ms, "records_ms": rec_ms,
                               "kernels_frac_of_hbm_peak": (sg_bytes / (sg_kernels_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if sg_kernels_ms > 0 else 0.0},
        "phases_ms_per_step": {k[2:]: round(v, 4) for k, v in phases.items()},
        "phase_hbm_frac": {k[2:]: round(alg[k] / (per_launch_ms[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                           for k in alg if per_launch_ms[k] > 0},
        "step_share_ms": {k[2:]: round(v, 4) for k, v in weight.items()},
    }
    if world == 1 and not args.no_ref_grid and n == 1_000_000:
        # the same suspension on the grid the REFERENCE's parameter rule picks at xi = 0.5 (360^3, PSEv1/Stokes.cc:135-199;
        # BASELINE.md section 3): a separate engine, a few steps, reported beside the headline (never the headline)
        del sim
        torch.cuda.empty_cache()
        ref = pdist.make_simulation(n, (L, L, L, 0.0), xi=0.5, error=args.error, seed=1, grid=(0, 0, 0), world=1, rank=0)
        ref.load(pos, force, mass=1.0)
        mr = 2
        for it in range(3):
            mr = ref.step(args.kT, args.dt, it, lanczos_m=mr)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_ref = 10
        for it in range(n_ref):
            mr = ref.step(args.kT, args.dt, 3 + it, lanczos_m=mr)
        torch.cuda.synchronize()
        ri = ref.info()
        out["reference_rule_grid"] = {"grid": [ri["Nx"], ri["Ny"], ri["Nz"]], "xi": 0.5, "rcut": ri["rcut"], "P": ri["P"],
                                      "ms_per_step": (time.perf_counter() - t0) / n_ref * 1e3, "steps": n_ref, "lanczos_m": mr}
    if not args.no_cpu and world == 1:     # the CPU baseline is reported with the single-GPU line only
        out["cpu_baseline"] = cpu_baseline()
    print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
