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
    env = dict(os.environ, TMPDIR="/tmp", PSE_OVERLAP="-1")    # (every launch alone on one stream)
    py = os.path.realpath(sys.executable)
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="pse_pmc_", dir="/tmp")
        cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--", py, os.path.abspath(__file__),
               "--steps", "3", "--warmup", "1", "--no-cpu", "--no-ref-grid", "--no-cfg4", "--no-async", "--no-traffic", "--n", str(args.n), "--phi", str(args.phi),
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


def _free_port():
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    return port


def run_children(cmds_envs, timeout_s, json_of=0, abort=None):
    """Start one child per (cmd, env) and wait for all of them: returns (exit code, the last JSON line with a "metric" or "segment" key
    that child `json_of` printed, or None).  Everything else the children print goes to stderr.  The first non-zero exit code is the
    result; when a child fails -- or `abort()` says another supervisor's child did, or the deadline passes (code 124) -- the others
    (exactly those PIDs) are terminated, then killed.  This process never touches the GPU and never replaces itself."""
    procs = []
    for k, (cmd, env) in enumerate(cmds_envs):
        procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE if k == json_of else sys.stderr, text=True, env=env))
    found = []

    def relay():
        for out in procs[json_of].stdout:
            out = out.rstrip("\n")
            try:
                obj = json.loads(out)
                if isinstance(obj, dict) and ("metric" in obj or "segment" in obj):      # a bare JSON scalar ("0", "true") is just a line of output
                    found.append(out)
                    continue
            except Exception:   # noqa: BLE001  (whatever the child prints, this thread must keep draining its pipe)
                pass
            print(out, file=sys.stderr)

    reader = None
    if 0 <= json_of < len(procs):
        reader = threading.Thread(target=relay, daemon=True)      # rank 0 may be the one left waiting when another rank dies
        reader.start()
    rc = 0
    pending = list(procs)
    t_all = time.time() + timeout_s
    t_kill = None                                       # once children were told to end: when the survivors are killed
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
        if pending and t_kill is None and (now > t_all or (abort is not None and abort())):
            late = now > t_all
            rc = rc or (124 if late else 125)
            print("bench.py: " + ("the ranks did not finish within their deadline" if late else "a rank of this run failed elsewhere") + ": ending them",
                  file=sys.stderr)
            for q in pending:
                q.terminate()
            t_kill = now + 20.0
        elif pending and t_kill is not None and now > t_kill:
            for q in pending:                           # SIGTERM was not enough (stuck in a driver or RCCL call): exactly these PIDs
                q.kill()
            t_kill = now + 20.0
        if pending:
            time.sleep(0.05)
    if reader is not None:
        reader.join(timeout=30)
    return rc, (found[-1] if found else None)


def launch_ranks(args, argv, script=None):
    """--gpus N without a torch.distributed environment, ONE set of ranks (--replicated, and the stand-in ranks of the CPU tests): the N
    ranks as children (one process per GPU, the environment torch.distributed.run would give them: RANK, LOCAL_RANK, WORLD_SIZE,
    MASTER_ADDR = 127.0.0.1, a free MASTER_PORT); rank 0's line is relayed, the first non-zero exit code among them is ours."""
    port = _free_port()
    rest = [a for a in argv if a != "--dry-run"]
    cmd = [sys.executable, script or os.path.abspath(__file__)] + rest
    rank_env = {"WORLD_SIZE": str(args.gpus), "LOCAL_WORLD_SIZE": str(args.gpus), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)}
    if args.dry_run:
        print(json.dumps({"launch": cmd, "env": dict(rank_env, RANK="<r>", LOCAL_RANK="<r>"), "n_gpus": args.gpus}))
        return 0
    env = dict(os.environ, **rank_env)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    rc, line = run_children([(cmd, dict(env, RANK=str(r), LOCAL_RANK=str(r))) for r in range(args.gpus)],
                            float(os.environ.get("PSE_BENCH_LAUNCH_TIMEOUT", "3600")))
    if rc != 0:
        print(f"bench.py: the {args.gpus}-rank run failed (exit code {rc})", file=sys.stderr)
        return rc
    if line is None:
        print("bench.py: rank 0 printed no result line", file=sys.stderr)
        return 1
    print(line)
    return 0


# ---- --gpus N (owned-particle step): the run is a sequence of SEGMENTS, each a fresh set of rank processes ------------------------------
# "single": one process on GPU 0 -- the single-GPU step at the metric point and at BASELINE config 4 (what the speed-ups divide by);
# "one_stream" / "lanes": the N ranks with PSE_TEAM_LANES = 0 / 1 -- kernels and exchanges of a rank on ONE stream, or two compute lanes
# + a communication stream (the faster mode on paper; it has never run over RCCL: a hang there must not cost the line of the first).
# Every segment verifies itself against a single-GPU engine before it is timed; the line's `value` is the faster VERIFIED mode.
# "split" (two GPUs only): no spatial decomposition at all -- rank 0 the real-space half of the step (near field + Lanczos), rank 1 the
# wave-space half (spread, FFTs, gather), each on ALL particles, one all-reduce of the two velocity halves per step: at G = 2 the slab
# all-to-alls of the owned-particle step move 107 MB over the ONE link between the two GPUs twice per step.
SEGMENTS = ("single", "one_stream", "lanes", "split", "host_fallback")   # (host_fallback: only when no RCCL mode ended with a verified trajectory)


class Coordinator:
    """What the supervisors of the ranks agree on without touching the GPU: a fresh rendezvous port per segment, and word that a
    rank's child failed (so that the others end theirs instead of waiting in a collective for the deadline).  One supervisor (we
    launched the ranks ourselves) needs none of it; under torch.distributed.run every rank process is the supervisor of its own
    rank and they talk through a TCPStore -- the launcher's (TORCHELASTIC_USE_AGENT_STORE) or one that rank 0 hosts on MASTER_PORT."""

    def __init__(self, world, rank, own_all):
        self.world, self.rank, self.store = world, rank, None
        if own_all:
            self.ports = [_free_port() for _ in SEGMENTS]
            return
        base = int(os.environ.get("MASTER_PORT", "29533"))
        self.ports = [base + 1 + k for k in range(len(SEGMENTS))]         # the fallback when no store can be had
        try:
            from datetime import timedelta
            from torch.distributed import TCPStore          # (host-side only: importing torch does not initialise HIP)
            agent = os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "") == "True"
            self.store = TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), base, world_size=None, is_master=(rank == 0 and not agent),
                                  timeout=timedelta(seconds=180), wait_for_workers=False)
            if rank == 0:
                self.store.set("pse_bench/ports", json.dumps([_free_port() for _ in SEGMENTS]))
            self.ports = json.loads(self.store.get("pse_bench/ports").decode())
        except Exception as e:   # noqa: BLE001
            print(f"bench.py: no store between the rank supervisors ({e!r}): ports MASTER_PORT + 1 .., deadlines only", file=sys.stderr)
            self.store = None

    def failed(self, k):
        if self.store is None:
            return False
        try:
            return bool(self.store.check([f"pse_bench/failed/{k}"]))
        except Exception:   # noqa: BLE001
            return False

    def decision(self, key, value=None):
        """Rank 0 decides (value given), everyone learns: True / False.  Without a store between supervisors the answer is `value` for the
        one that decides and False for the others (a step only some ranks take would hang)."""
        if self.store is None:
            return bool(value)
        try:
            if value is not None:
                self.store.set(f"pse_bench/decision/{key}", "1" if value else "0")
                return bool(value)
            return self.store.get(f"pse_bench/decision/{key}").decode() == "1"
        except Exception:   # noqa: BLE001
            return False

    def say_failed(self, k):
        if self.store is not None:
            try:
                self.store.set(f"pse_bench/failed/{k}", "1")
            except Exception:   # noqa: BLE001
                pass


def supervise_segments(args, argv, script=None):
    """The owned-particle run of `bench.py --gpus N`: this process starts the rank processes of every segment (all N of them when it
    was started plainly, its own rank's when it is one of torch.distributed.run's workers), merges rank 0's lines and prints ONE line."""
    under_launcher = "WORLD_SIZE" in os.environ
    world = args.gpus
    if under_launcher and int(os.environ["WORLD_SIZE"]) != world:
        print(f"bench.py: --gpus {world} but the torch.distributed environment has WORLD_SIZE={os.environ['WORLD_SIZE']}", file=sys.stderr)
        return 2
    rank = int(os.environ.get("RANK", "0")) if under_launcher else 0
    rest = [a for a in argv if a != "--dry-run"]
    me = script or os.environ.get("PSE_BENCH_RANK_SCRIPT") or os.path.abspath(__file__)      # (the variable: stand-in ranks of the CPU tests)
    if args.dry_run:
        print(json.dumps({"launch": [sys.executable, me] + rest, "segments": list(SEGMENTS), "n_gpus": world,
                          "env": {"WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "<one per segment>", "RANK": "<r>", "LOCAL_RANK": "<r>"}}))
        return 0
    co = Coordinator(world, rank, own_all=not under_launcher)
    base_env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}      # (the children rendezvous among themselves)
    base_env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    base_env.setdefault("OMP_NUM_THREADS", "1")
    base_env["MASTER_ADDR"] = "127.0.0.1"
    deadline = float(os.environ.get("PSE_BENCH_SEGMENT_TIMEOUT", "900"))
    results, codes = {}, {}
    for k, seg in enumerate(SEGMENTS):
        if seg == "single" and (args.no_single or rank != 0):
            continue
        if seg == "lanes" and args.modes == "one_stream" or seg == "one_stream" and args.modes == "lanes":
            continue
        if seg == "split" and (world != 2 or args.modes not in ("both", "split")) or seg in ("one_stream", "lanes") and args.modes == "split":
            continue
        extra = []
        if seg == "host_fallback":
            # No RCCL mode ended with a verified trajectory (a team over RCCL has never run before the first multi-GPU node): the same
            # step once more with every exchange staged through host memory, so that the line still says whether the decomposition
            # itself is right on these GPUs -- a measurement of the fallback transport, labelled as such, never of the RCCL path.
            if args.transport == "host" or args.modes == "split":
                continue
            want = None
            if rank == 0:
                want = not any("error" not in r and (r.get("verify") or {}).get("ok", args.no_verify)
                               for m, r in results.items() if m in ("one_stream", "lanes", "split"))
                if want and os.environ.get("PSE_BENCH_FALLBACK_ANYWAY", "") != "1":
                    # ranks that would SHARE a GPU measure nothing about a multi-GPU node: no fallback there (the run ends with the
                    # refusal of the RCCL segments, exit code 3); the variable is for the test of this path on a one-GPU box
                    try:
                        import torch    # (device_count does not initialise the GPU; this process never does)
                        want = torch.cuda.device_count() >= world
                    except Exception:   # noqa: BLE001
                        want = False
            if not co.decision("host_fallback", want):
                continue
            extra = ["--transport", "host"]
        n_ranks = 1 if seg == "single" else world
        ranks = list(range(n_ranks)) if not under_launcher else [rank]
        env = dict(base_env, WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks), MASTER_PORT=str(co.ports[k]))
        if seg in ("one_stream", "lanes", "host_fallback"):
            env["PSE_TEAM_LANES"] = "1" if seg == "lanes" else "0"
        cmd = [sys.executable, me, "--segment", "one_stream" if seg == "host_fallback" else seg] + rest + extra
        t0 = time.time()
        rc, line = run_children([(cmd, dict(env, RANK=str(r), LOCAL_RANK=str(r))) for r in ranks], deadline,
                                json_of=0 if rank == 0 else -1, abort=(lambda k=k: co.failed(k)) if under_launcher and seg != "single" else None)
        codes[seg] = rc
        if rc != 0:
            co.say_failed(k)
            print(f"bench.py: segment {seg} failed on this supervisor (exit code {rc}) after {time.time() - t0:.0f} s", file=sys.stderr)
        if rank == 0:
            try:
                results[seg] = json.loads(line) if (rc == 0 and line) else {"error": f"exit code {rc}" if rc else "no result line", "seconds": round(time.time() - t0, 1)}
            except Exception:   # noqa: BLE001
                results[seg] = {"error": "unreadable result line"}
    if rank != 0:
        return 0 if any(codes.get(sg, 1) == 0 for sg in ("one_stream", "lanes", "split", "host_fallback")) else next((c for c in codes.values() if c), 1)
    return merge_segments(args, results)


def merge_segments(args, results):
    """ONE line from the segments' lines: `value` is the faster mode whose trajectory check against the single GPU passed."""
    modes = {m: results[m] for m in ("one_stream", "lanes", "split", "host_fallback") if m in results}
    good = {m: r for m, r in modes.items() if "error" not in r and (r.get("verify") or {}).get("ok", args.no_verify)}
    single = results.get("single")
    if not good:
        print("bench.py: no mode of the team finished with a verified trajectory: " +
              json.dumps({m: (r.get("error") or r.get("verify")) for m, r in modes.items()}), file=sys.stderr)
        return 3
    best = min(good, key=lambda m: good[m]["ms_per_step"])
    out = dict(good[best])
    out.pop("segment", None)
    out["mode"] = best
    if best == "host_fallback":
        out["mode_note"] = ("NO RCCL mode ended with a verified trajectory (modes.*.error / verify): this is the one-stream step with every exchange "
                            "staged through host memory -- the fallback transport, not the product's multi-GPU path")
    keep = ("ms_per_step", "value", "steps_per_s", "mf_evals_per_s", "lanczos_m", "lanczos_status", "lanczos_exchanges", "lanczos_extras_off", "verify",
            "exchanges_per_step", "exchange_us", "exchange_host_us", "exchange_bytes", "lanes_ms", "critical_path_ms", "device_flags", "particles_owned_sum",
            "config4", "error", "seconds")
    out["modes"] = {m: {k: r[k] for k in keep if k in r} for m, r in modes.items()}
    if single and "error" not in single:
        out["single_gpu"] = {"ms_per_step": single["ms_per_step"], "config4_ms_per_step": (single.get("config4_single_gpu") or {}).get("ms_per_step"),
                             "note": "one process on GPU 0 of this node, same suspension, measured in this run before the team"}
        out["speedup_vs_single"] = single["ms_per_step"] / out["ms_per_step"]
        c4 = out.get("config4")
        if c4 and "ms_per_step" in c4 and out["single_gpu"]["config4_ms_per_step"]:
            c4["speedup_vs_single"] = out["single_gpu"]["config4_ms_per_step"] / c4["ms_per_step"]
            for r in out["modes"].values():
                if isinstance(r.get("config4"), dict) and "ms_per_step" in r["config4"]:
                    r["config4"]["speedup_vs_single"] = out["single_gpu"]["config4_ms_per_step"] / r["config4"]["ms_per_step"]
    elif single:
        out["single_gpu"] = single
    sp = out.get("speedup_vs_single")
    out["north_star"] = {"speedup_at_this_gpu_count": [6.0 if args.gpus == 8 else None, sp],
                         "config4_speedup_at_this_gpu_count": [6.0 if args.gpus == 8 else None, (out.get("config4") or {}).get("speedup_vs_single")],
                         "note": "[target, measured]; the target is BASELINE.json's >= 6x at 8 GPUs"}
    print(json.dumps(out))
    return 0


def time_team_workload(args, n, phi, grid, steps, warmup, world, rank, host_transport, dist, torch, verify_steps, with_mf=True):
    """One workload on the owned-particle team of this segment: (verify against a single-GPU engine on rank 0,) warm-up -- the starting
    count of the Lanczos iteration settles and, once `settle` steps in a row have ended at it, the gated extra block is switched off
    (pse_amd.sharded.LanczosCount: the same decisions on every rank, from numbers every rank holds) --, then EXACTLY `steps` steps
    between two barriers, then five steps with every exchange bracketed by events.  Returns the dict of the workload (rank 0) or None."""
    from pse_amd.sharded import LanczosCount, LocalShardedSimulation
    pos, force, L = suspension(n, phi)
    box = (L, L, L, 0.0)
    xi = math.pi * grid / (2.0 * L * math.sqrt(-math.log(args.error)))      # xi from the fixed grid
    kw = dict(xi=xi, error=args.error, seed=1, grid=(grid,) * 3)
    sim = LocalShardedSimulation(n, box, world, rank, transport="host" if host_transport else "rccl", **kw)
    sim.load(pos, force, mass=1.0)
    # Before anything is timed: a few steps of the team next to the single-GPU engine on rank 0 (same suspension, same noise) -- the
    # first run on a multi-GPU node then says whether the exchanges moved the right bytes, not only how long they took.
    verify = None
    if verify_steps > 0:
        verify = verify_team_against_single_gpu(sim, pos, force, box, kw, args, world, rank, dist, torch, steps=verify_steps)
        sim.load(pos, force, mass=1.0)
    dev = "cpu" if host_transport else "cuda"

    def barrier():
        dist.barrier()
        torch.cuda.synchronize()

    def agree(v):     # (every rank holds the same m and status after a step -- they take the same decisions from the same sums; the
        t = torch.tensor([float(v)], dtype=torch.float64, device=dev)          # all-reduce makes a rank that does not hold them show)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return int(t[0])

    lc = LanczosCount(sim.team, m=2, settle=3, adaptive=not args.keep_extras)
    it = 0
    while it < max(warmup, 1) or (not lc.extras_off and not args.keep_extras and it < warmup + 12):
        sim.step(args.kT, args.dt, it, lanczos_m=lc.m)
        torch.cuda.synchronize()
        i = sim.engine.info()
        lc.seen(agree(i["lanczos_m"]), agree(i["lanczos_status"]))
        it += 1
    m = lc.m
    t_mf = None
    if with_mf:      # deterministic M.F (kT = 0, no update)
        barrier()
        n_mf = max(3, steps // 2)
        t0 = time.perf_counter()
        for k in range(n_mf):
            sim.step(0.0, args.dt, 0, integrate=False)
        barrier()
        t_mf = (time.perf_counter() - t0) / n_mf
    # headline: EXACTLY `steps` steps between two barriers; nothing is read back inside (the step only queues work)
    barrier()
    t0 = time.perf_counter()
    for k in range(steps):
        sim.step(args.kT, args.dt, it + k, lanczos_m=m)
    barrier()
    elapsed = time.perf_counter() - t0
    i = sim.engine.info()
    status, m_seen = agree(i["lanczos_status"]), agree(i["lanczos_m"])
    # a few more steps with every exchange bracketed by events
    sim.team.set_diag(True)
    diags = []
    for k in range(5):
        sim.step(args.kT, args.dt, it + steps + k, lanczos_m=m)
        diags.append(sim.team.diag())
    sim.team.set_diag(False)
    flags = sim.team.local_status()
    t = torch.tensor([elapsed, t_mf or 0.0], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, t_mf = float(t[0]), float(t[1])
    n_loc = torch.tensor([float(sim.s.n_local.item())], dtype=torch.float64, device=dev)
    dist.all_reduce(n_loc, op=dist.ReduceOp.SUM)
    info = sim.engine.info()
    lay, cap = sim.layout, sim.engine.params.n_max
    sim.team.close()
    del sim
    torch.cuda.empty_cache()
    if rank != 0:
        return None
    d = diags[-1]
    med = lambda key: {k: [round(float(np.median([x[key][k][j] for x in diags])), 2) for j in range(len(d[key][k]))] for k in d[key]}   # noqa: E731
    t_step = elapsed / steps
    return {
        "workload": f"random-sphere suspension N={n}, phi={phi}, cubic L={L:.2f}, grid {grid}^3, xi={xi:.4f}, rcut={info['rcut']:.3f}, P={info['P']}, "
                    f"error={args.error}, kT={args.kT}, dt={args.dt}",
        "layout": f"{lay['layers_per_rank']} of {lay['layers']} cell layers along x per rank + 2 ghost layers per side, row capacity {cap} per rank",
        "ms_per_step": t_step * 1e3, "value": n / t_step, "steps_per_s": 1.0 / t_step, "mf_evals_per_s": (1.0 / t_mf) if t_mf else None,
        "steps": steps, "lanczos_m": m_seen, "lanczos_status": status, "lanczos_exchanges": info["lanczos_exchanges"],
        "lanczos_extras_off": bool(lc.extras_off), "particles_owned_sum": int(n_loc[0]), "device_flags": flags,
        # what the first run on a multi-GPU node needs: per-exchange device time by kind (median of 5 steps), the host time spent issuing
        # each, bytes this rank sent, the spans of the two lanes
        "exchanges_per_step": d["exchanges_per_step"], "exchange_us": med("exchange_us"), "exchange_host_us": med("exchange_host_us"),
        "exchange_bytes": d["exchange_bytes"],
        "lanes_ms": {k: round(float(np.median([x["lanes_ms"][k] for x in diags])), 4) for k in ("main", "side")},
        "critical_path_ms": round(float(np.median([x["critical_path_ms"] for x in diags])), 4),
        "verify": verify,
    }


def run_owned_particle_team(args, world, rank, host_transport, dist, torch, segment):
    """One mode of `--gpus N` (a segment: this process is one of its N ranks): every rank owns the particles of its x slab
    (pse_team_step_local): migration + ghosts in one exchange of fixed-size messages, the whole step queue-only.  The metric point, then
    BASELINE config 4 (the size at which 8 ranks have the far field and the near field to themselves long enough for >= 6x to be within
    reach: DESIGN.md section 6)."""
    w = time_team_workload(args, args.n, args.phi, args.grid, args.steps, args.warmup, world, rank, host_transport, dist, torch,
                           0 if args.no_verify else 3)
    c4 = None
    if not args.no_cfg4:
        try:
            c4 = time_team_workload(args, args.cfg4_n, args.cfg4_phi, args.cfg4_grid, max(3, min(args.steps, 10)), max(1, min(args.warmup, 3)), world, rank,
                                    host_transport, dist, torch, 0 if args.no_verify else 2, with_mf=False)
        except Exception as e:   # noqa: BLE001  (the metric point's numbers are in hand: config 4 must not cost them)
            c4 = {"error": repr(e)[:300]}
            print(f"bench.py: config 4 failed on rank {rank}: {e!r}", file=sys.stderr)
    if rank != 0:
        return 0
    transport = ("HOST-STAGED transport (torch.distributed gloo, ranks may share a GPU): a functional run of the process-per-rank driver, "
                 "not a scaling number" if host_transport else "RCCL over xGMI")
    lanes = os.environ.get("PSE_TEAM_LANES", "default")
    out = {
        "segment": segment,
        "metric": "BD particle-steps/s (full PSE Brownian step: M.F + k-space noise + Lanczos M^1/2.psi + Euler), N=1e6, phi=0.1",
        "value": w["value"], "unit": "particle-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": w["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "dtype_note": "all arithmetic fp64; inside the Lanczos mat-vec the pair coefficients are read from 16-byte records (26-bit fixed / 22-bit block-floating point: single-precision accuracy) and, on one GPU, the neighbours' vector rows from 40-bit mantissas",
        "data": "synthetic",
        "config": {"workload": w["workload"],
                   "parallelism": f"{world} ranks, {transport}; owned-particle decomposition (pse_team_step_local): {w['layout']}, far-field grid in "
                                  f"{world} x-slabs (2 all-to-alls + 1 plane halo per step on the far-field lane), two Lanczos iterations per exchange; "
                                  f"PSE_TEAM_LANES={lanes} ({'two compute lanes + a communication stream' if lanes == '1' else 'one stream for kernels and exchanges'}); "
                                  f"the force provider of the bench re-gathers its fixed forces by tag after every step"},
        "roofline": None, "cpu_baseline": None,
    }
    out.update({k: w[k] for k in w if k not in ("workload", "layout", "value", "ms_per_step", "steps")})
    if c4 is not None:
        out["config4"] = c4 if "error" in c4 else {k: c4[k] for k in ("workload", "layout", "ms_per_step", "value", "steps", "lanczos_m", "lanczos_status",
                                                                         "lanczos_exchanges", "lanczos_extras_off", "exchange_us", "exchange_bytes", "lanes_ms",
                                                                         "critical_path_ms", "device_flags", "verify")}
    if w["verify"] is not None and not w["verify"]["ok"]:
        print(f"bench.py: the team's trajectory differs from the single GPU's: {w['verify']}", file=sys.stderr)
    print(json.dumps(out))
    return 0 if (w["verify"] is None or w["verify"]["ok"]) else 4


def run_functional_split(args, rank, host_transport, dist, torch):
    """The `split` segment of `--gpus 2` (pse_amd.sharded.SplitSimulation): rank 0 the real-space half of every step, rank 1 the wave-space
    half, both on all particles, one all-reduce per step; verified against a single-GPU engine on rank 0, then timed like the teams."""
    import pse_amd
    from pse_amd.sharded import SplitSimulation

    def workload(n, phi, grid, steps, warmup, verify_steps):
        pos, force, L = suspension(n, phi)
        box = (L, L, L, 0.0)
        xi = math.pi * grid / (2.0 * L * math.sqrt(-math.log(args.error)))
        kw = dict(xi=xi, error=args.error, seed=1, grid=(grid,) * 3)
        sim = SplitSimulation(n, box, rank, dist, **kw)
        verify = None
        if verify_steps > 0:
            sim.load(pos, force, mass=1.0)
            dt = 0.05
            worst, images_equal, m_equal = 0.0, True, True
            if rank == 0:
                ref = pse_amd.Engine(n, box, **kw)
                dpos, dF, vel = sim.pos.clone(), sim.force.clone(), sim.vel.clone()
                accel = torch.zeros((n, 3), dtype=torch.float64, device="cuda"); image = torch.zeros((n, 3), dtype=torch.int32, device="cuda")
            m = mr = 2
            for k in range(verify_steps):
                m = sim.step(args.kT, dt, 1000 + k, lanczos_m=m)
                if rank == 0:
                    mr = ref.step(dpos, vel, accel, image, dF, args.kT, dt, 1000 + k, lanczos_m=mr)
                    worst = max(worst, float((sim.pos[:, :3] - dpos[:, :3]).abs().max()))
                    images_equal = images_equal and bool(torch.equal(sim.image, image))
                    m_equal = m_equal and m == mr
            if rank == 0:
                del ref
                torch.cuda.empty_cache()
                verify = {"steps": verify_steps, "dt": dt, "max_abs_position_diff_vs_single_gpu": worst, "images_equal": images_equal,
                          "lanczos_m_equal": m_equal, "ok": bool(worst < 1e-3 and images_equal and m_equal)}
        sim.load(pos, force, mass=1.0)
        m = 2
        for it in range(max(warmup, 1)):
            m = sim.step(args.kT, args.dt, it, lanczos_m=m)
        dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(steps):
            m = sim.step(args.kT, args.dt, warmup + it, lanczos_m=m)
        dist.barrier(); torch.cuda.synchronize()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cpu" if host_transport else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        info = sim.engine.info()
        del sim
        torch.cuda.empty_cache()
        t_step = float(t[0]) / steps
        return {"workload": f"random-sphere suspension N={n}, phi={phi}, cubic L={L:.2f}, grid {grid}^3, xi={xi:.4f}, rcut={info['rcut']:.3f}, P={info['P']}, "
                            f"error={args.error}, kT={args.kT}, dt={args.dt}",
                "ms_per_step": t_step * 1e3, "value": n / t_step, "steps_per_s": 1.0 / t_step, "steps": steps, "lanczos_m": m, "lanczos_status": 0,
                "exchange_bytes": {"all_reduce": [32 * n]}, "verify": verify}

    w = workload(args.n, args.phi, args.grid, args.steps, args.warmup, 0 if args.no_verify else 3)
    c4 = None
    if not args.no_cfg4:
        try:
            c4 = workload(args.cfg4_n, args.cfg4_phi, args.cfg4_grid, max(3, min(args.steps, 10)), max(1, min(args.warmup, 3)), 0 if args.no_verify else 2)
        except Exception as e:   # noqa: BLE001
            c4 = {"error": repr(e)[:300]}
    if rank != 0:
        return 0
    transport = "torch.distributed gloo through host memory (ranks may share a GPU): a functional run, not a scaling number" if host_transport else "RCCL all-reduce over xGMI"
    out = {"segment": "split",
           "metric": "BD particle-steps/s (full PSE Brownian step: M.F + k-space noise + Lanczos M^1/2.psi + Euler), N=1e6, phi=0.1",
           "value": w["value"], "unit": "particle-steps/s", "n_gpus": 2, "steps": args.steps, "warmup": args.warmup, "ms_per_step": w["ms_per_step"],
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "dtype_note": "all arithmetic fp64; inside the Lanczos mat-vec the pair coefficients are read from 16-byte records (26-bit fixed / 22-bit block-floating point: single-precision accuracy) and, on one GPU, the neighbours' vector rows from 40-bit mantissas", "data": "synthetic",
           "config": {"workload": w["workload"],
                      "parallelism": f"2 ranks, {transport}; FUNCTIONAL split (pse_brownian_velocity_part): rank 0 the real-space half of every step (near field + "
                                     f"Lanczos), rank 1 the wave-space half (spread, transforms, k-space scaling and noise, gather), both on all N particles, ONE "
                                     f"all-reduce of 32 N bytes per step, both ranks integrate"},
           "roofline": None, "cpu_baseline": None}
    out.update({k: w[k] for k in ("steps_per_s", "lanczos_m", "lanczos_status", "exchange_bytes", "verify")})
    if c4 is not None:
        out["config4"] = c4
    print(json.dumps(out))
    return 0 if (w["verify"] is None or w["verify"]["ok"]) else 4


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
            # (1e-3, not 1e-9: the pair coefficients of the Lanczos mat-vecs are single precision -- a coefficient that rounds the other way
            # in one of the two runs moves a particle by ~1e-8 dt -- and a pair that then lies on the other side of rcut by ~3e-5 dt; a wrong
            # exchange shows as 1e-2 and more: tests/conftest.py TRAJ_TOL_BROWNIAN has the mechanisms and their sizes)
            "ok": bool(worst < 1e-3 and images_equal and m_equal and status_ok)}


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
    ap.add_argument("--modes", choices=["both", "one_stream", "lanes", "split"], default="both",
                    help="multi-rank runs: which lane modes of the owned-particle step are timed (default both, each as a fresh set of rank "
                         "processes; the line's value is the faster verified one)")
    ap.add_argument("--no-single", action="store_true", help="multi-rank runs: skip the single-GPU segment (no speedup_vs_single in the line)")
    ap.add_argument("--keep-extras", action="store_true",
                    help="multi-rank runs: keep the gated extra Lanczos block queued in every step (default: switched off once three steps in a row "
                         "ended at their starting count, back on at the first lanczos_status 1)")
    ap.add_argument("--no-async", action="store_true", help="single GPU: do not time the queue-only form of the step (pse_set_async) beside the host-checked one")
    ap.add_argument("--no-cfg4", action="store_true", help="skip the extra block at BASELINE config 4 (N = 4194304, phi = 0.3, 512^3)")
    ap.add_argument("--cfg4-n", type=int, default=4_194_304, help=argparse.SUPPRESS)      # (the tests shrink the block)
    ap.add_argument("--cfg4-phi", type=float, default=0.3, help=argparse.SUPPRESS)
    ap.add_argument("--cfg4-grid", type=int, default=512, help=argparse.SUPPRESS)
    ap.add_argument("--segment", choices=list(SEGMENTS), default=None, help=argparse.SUPPRESS)   # (set by supervise_segments for its children)
    args = ap.parse_args()
    argv = sys.argv[1:]
    if args.gpus > 1 and args.segment is None:
        if not args.replicated:      # the owned-particle run: a sequence of segments, each a fresh set of rank processes
            raise SystemExit(supervise_segments(args, argv))
        if "WORLD_SIZE" not in os.environ:
            raise SystemExit(launch_ranks(args, argv))
    if args.dry_run:
        print(json.dumps({"launch": None, "n_gpus": args.gpus}))
        return
    if args.segment == "single":     # one process on GPU 0: what the team's speed-ups divide by
        args.gpus, args.no_cpu, args.no_traffic, args.no_ref_grid = 1, True, True, True
        args.steps, args.warmup = min(args.steps, 20), min(args.warmup, 5)

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

    if world > 1 and args.segment == "split":
        rc = run_functional_split(args, rank, host_transport, dist, torch)
        dist.destroy_process_group()
        if rc:
            raise SystemExit(rc)
        return
    if world > 1 and not args.replicated:
        rc = run_owned_particle_team(args, world, rank, host_transport, dist, torch, args.segment)
        dist.destroy_process_group()
        if rc:
            raise SystemExit(rc)
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
    # The same K steps through the queue-only form of the SAME entry point (pse_set_async(1): pse_step only queues work, the Lanczos
    # convergence decision is taken on the device, gated extra iterations leave at once, nothing is read back -- what a host that
    # does not want to stall its stream every step calls; captured-graph and parity tests: tests/test_gpu_async.py).  The steps are
    # valid if every one ended with lanczos_status 0 at the count the host-checked steps use; the line's value is the faster form.
    modes = {"host_checked": {"ms_per_step": elapsed / args.steps * 1e3, "lanczos_m": float(np.mean(ms))}}
    mode = "host_checked"
    eng = getattr(sim, "engine", None)
    if world == 1 and eng is not None and not args.no_async:
        eng.set_async(True)
        m_q = int(round(float(np.max(ms))))
        # the steady state of a time-stepping loop (pse_amd.sharded.LanczosCount, the policy of the team bench): once `settle` steps in a
        # row have ended at their starting count with status 0 the gated extra iterations are not queued at all (pse_set_lanczos_extra(0):
        # ten launches per step that would leave at once); a step that then runs out says so -- pse_info.lanczos_open_calls is sticky,
        # and a timed loop in which it moved is not the line's value
        from pse_amd.sharded import LanczosCount
        lc = LanczosCount(eng, m=m_q, settle=3, adaptive=not args.keep_extras)
        it = 0
        while it < 3 or (not lc.extras_off and not args.keep_extras and it < 12):
            sim.step(args.kT, args.dt, 500000 + it, lanczos_m=lc.m)
            torch.cuda.synchronize()
            i_w = eng.info()
            lc.seen(i_w["lanczos_m"], i_w["lanczos_status"])
            it += 1
        m_q = lc.m
        open0 = eng.info()["lanczos_open_calls"]
        barrier()
        t0 = time.perf_counter()
        for it in range(args.steps):
            sim.step(args.kT, args.dt, 600000 + it, lanczos_m=m_q)
        barrier()
        el_q = time.perf_counter() - t0
        iq = eng.info()
        eng.set_lanczos_extra(-1)
        eng.set_async(False)
        modes["queue_only"] = {"ms_per_step": el_q / args.steps * 1e3, "lanczos_m": iq["lanczos_m"], "lanczos_status": iq["lanczos_status"],
                               "starting_count": m_q, "gated_extras_off": lc.extras_off, "steps_that_ran_out": int(iq["lanczos_open_calls"] - open0)}
        if iq["lanczos_status"] == 0 and iq["lanczos_m"] == m_q and iq["lanczos_open_calls"] == open0 and el_q < elapsed:
            mode, elapsed = "queue_only", el_q
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
           "n": n_pct, "form": "host-checked steps, each bracketed by device events"}
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
    # The dominant kernel's launch time.  The `matvec` of phases_ms_per_step is ONE pair of HIP events around one launch of the pair-list
    # mat-vec per step: it contains what an event pair costs by itself (two markers in a row are 4.5 us apart with nothing between them:
    # measured below, on the same stream) and a dispatch that nothing pipelines (the marker in front has drained the GPU) -- together 8 - 12 us
    # that a kernel inside an iteration does not pay and a profiler's per-kernel duration does not contain.  What the roofline divides by
    # is therefore twenty launches back to back between ONE pair of events (pse_debug_matvec_ms: the list, the vector and its mirror as
    # the last step left them), which is how the launches of an iteration follow one another; in the committed profile pair it is within
    # 2 - 3 % of rocprofv3's per-kernel average of the same build (114.0 against 111.9 us; 129.3 against 128.5 before the 16-byte records).
    # The bracketed launch and the empty pair are reported beside it.
    pair_ms, mv_warm_ms = 0.0, None
    if world == 1:
        lat = []
        for _ in range(40):
            ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ea.record(); eb.record()
            eb.synchronize()
            lat.append(ea.elapsed_time(eb))
        pair_ms = float(np.median(lat))
        if args.kT > 0 and hasattr(getattr(sim, "engine", None), "matvec_ms"):
            try:
                mv_warm_ms = sim.engine.matvec_ms(20)
            except Exception as e:   # noqa: BLE001  (no pair list, e.g. PSE_SKIN experiments)
                print(f"bench.py: pse_debug_matvec_ms: {e}", file=sys.stderr)
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
    mv_in_step_ms = per_launch_ms["t_matvec"]
    if mv_warm_ms:
        per_launch_ms["t_matvec"] = mv_warm_ms      # (see above; without the debug entry point the bracketed launch minus the empty pair)
    elif mv_in_step_ms > 2.0 * pair_ms:
        per_launch_ms["t_matvec"] = mv_in_step_ms - pair_ms
    # share of the step: the pair-list mat-vec runs once per Lanczos iteration except the first, whose M.psi is delivered by
    # the near-field pass that builds the list
    weight = dict(per_launch_ms)
    weight["t_matvec"] = per_launch_ms["t_matvec"] * max(1, info["lanczos_matvecs"] - 1)
    dom = max(weight, key=weight.get)
    own_fft = grid in (256, 512)      # (pse_capi.hip make_plans: the own z pass at Nz = 256 / 512 / 360 ..., the register y pass at 256 / 512)
    names = {"t_spread": "k_spread_tiles (spread, incl. binning + records)",
             "t_fft_fwd": "k_zfft_rows (real -> half spectrum along z) + k_yfft_regs (y)" if own_fft else "z and y transforms (own passes where they exist, else rocFFT)",
             "t_scale": "k_xfft_scale_cols (x FFT + k-space scale/noise + inverse x FFT)",
             "t_fft_inv": "k_yfft_regs (y) + k_zfft_rows (half spectrum -> real along z)" if own_fft else "y and z transforms (own passes where they exist, else rocFFT)",
             "t_gather": "k_gather_bins (gather)",
             "t_real": "k_mreal_cells (near-field M_real.F from the cell list, writes the pair list)",
             "t_matvec": "k_mreal_list (near-field mat-vec from the pair list, once per Lanczos iteration after the first)"}
    # what the SQ / TA counters of profiles/ say bounds each of them: `bound` stays the roofline the bytes are priced against
    limiters = {"t_matvec": "list stream (HBM) + gather round trips in flight (TA): profiles/r06_sq_counters.txt",
                "t_real": "texture-address path of the drain's dependent gathers (TA ~65 % busy, VALU ~14 %), not HBM",
                "t_spread": "VALU issue (13.6 of 64 lanes live per particle footprint)", "t_gather": "LDS-DMA region fill rate (4.3x halo-redundant regions)",
                "t_scale": "vector instructions of the k-space operator", "t_fft_fwd": "HBM (two passes)", "t_fft_inv": "HBM (two passes)"}
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
    placement = eng.grid_placement() if eng is not None and hasattr(eng, "grid_placement") else None   # what pse_create's placement planner did
    if eng is not None and hasattr(eng, "neighbor_stats"):
        rb, nb_, nr_ = eng.neighbor_stats()
        nl_note = {"r_buff": rb, "calls_that_built": nb_, "calls_that_reused": nr_,
                   "note": "mf_evals_per_s is measured at fixed positions and runs on the kept list; the steps of the headline do not"}
    out = {
        "metric": "BD particle-steps/s (full PSE Brownian step: M.F + k-space noise + Lanczos M^1/2.psi + Euler), "
                  "N=1e6, phi=0.1",
        "value": n / t_step, "unit": "particle-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": t_step * 1e3, "ms_per_step_percentiles": pct, "step_modes": modes, "step_mode": mode,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "dtype_note": "all arithmetic fp64; inside the Lanczos mat-vec the pair coefficients are read from 16-byte records (26-bit fixed / 22-bit block-floating point: single-precision accuracy) and, on one GPU, the neighbours' vector rows from 40-bit mantissas", "data": "synthetic",
        "config": {"workload": f"random-sphere suspension N={n}, phi={args.phi}, cubic L={L:.2f}, grid {grid}^3, "
                               f"xi={xi:.4f}, rcut={info['rcut']:.3f}, P={info['P']}, error={args.error}, kT={args.kT}, "
                               f"dt={args.dt}", "parallelism": sim.describe() + ("; steps through the queue-only form of pse_step (pse_set_async: device-side "
                                                                                  "Lanczos decision, nothing read back)" if mode == "queue_only" else "")},
        "steps_per_s": 1.0 / t_step, "mf_evals_per_s": 1.0 / t_mf, "mf_particle_evals_per_s": n / t_mf,
        "mf_evals_per_s_moving": (1.0 / t_mf_moving) if t_mf_moving else None,
        # the whole step against the roofline: SURVEY.md 8(d) A_step = A_MF + 32 N + (m + 1) 256 N + 32 N m + 160 N over all ranks
        "step_roofline": {"algorithmic_bytes": a_step, "ms": t_step * 1e3, "lanczos_m": m_avg,
                          "frac_of_hbm_peak": a_step / t_step / 1e9 / (HBM_PEAK_GBS * world)},
        "lanczos_m": m_avg, "lanczos_matvecs_per_step": info["lanczos_matvecs"],
        "neighbor_list": nl_note,
        "grid_placement": placement,
        "roofline": {"bound": "hbm", "limiter": limiters.get(dom), "kernel": names[dom], "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": alg[dom], "ms_per_launch": per_launch_ms[dom],
                     "ms_per_launch_how": ("20 launches back to back between one pair of HIP events on the engine's stream, on the state the last timed "
                                           "step left (pse_debug_matvec_ms)" if (dom == "t_matvec" and mv_warm_ms) else
                                           "HIP event pair around one launch per step on the stream it is launched on, minus an empty pair's latency"),
                     "ms_per_launch_bracketed_in_step": mv_in_step_ms if dom == "t_matvec" else None, "event_pair_latency_ms": pair_ms,
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
                                      "ms_per_step": (time.perf_counter() - t0) / n_ref * 1e3, "steps": n_ref, "lanczos_m": mr,
                                      "grid_placement": ref.engine.grid_placement() if hasattr(getattr(ref, "engine", None), "grid_placement") else None}
    if world == 1 and not args.no_cfg4:
        # BASELINE config 4 (N = 4194304, phi = 0.3, 512^3, xi from the grid) on this ONE GPU: what a multi-GPU line's config-4 block is
        # divided by (its own engine, a few steps; never the headline)
        try:
            ref = None
            if "sim" in dir():
                del sim
            torch.cuda.empty_cache()
            n4, g4 = args.cfg4_n, args.cfg4_grid
            pos4, force4, L4 = suspension(n4, args.cfg4_phi)
            xi4 = math.pi * g4 / (2.0 * L4 * math.sqrt(-math.log(args.error)))
            c4 = pdist.make_simulation(n4, (L4, L4, L4, 0.0), xi=xi4, error=args.error, seed=1, grid=(g4,) * 3, world=1, rank=0)
            c4.load(pos4, force4, mass=1.0)
            m4 = 2
            for it in range(3):
                m4 = c4.step(args.kT, args.dt, it, lanczos_m=m4)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n_c4 = 8
            for it in range(n_c4):
                m4 = c4.step(args.kT, args.dt, 3 + it, lanczos_m=m4)
            torch.cuda.synchronize()
            t4 = (time.perf_counter() - t0) / n_c4
            out["config4_single_gpu"] = {"workload": f"N={n4}, phi={args.cfg4_phi}, grid {g4}^3, xi={xi4:.4f}", "ms_per_step": t4 * 1e3, "steps": n_c4,
                                         "lanczos_m": m4, "particle_steps_per_s": n4 / t4,
                                         "grid_placement": c4.engine.grid_placement() if hasattr(getattr(c4, "engine", None), "grid_placement") else None}
            out["config4_single_gpu_ms"] = t4 * 1e3
            del c4, pos4, force4
            torch.cuda.empty_cache()
        except Exception as e:   # noqa: BLE001  (a smaller GPU: the headline must not depend on this block)
            out["config4_single_gpu"] = {"error": repr(e)[:300]}
    # the three clauses of BASELINE.json's north star, [target, measured], read off one line
    sg = out["spread_plus_gather"]["frac_of_hbm_peak"]
    out["north_star"] = {"steps_per_s": [50, out["steps_per_s"]], "spread_gather_frac_of_hbm": [0.40, sg],
                         "speedup_8gpu": [6, None], "note": "[target, measured]; the 8-GPU clause is measured by `bench.py --gpus 8` (speedup_vs_single there)"}
    if not args.no_cpu and world == 1:     # the CPU baseline is reported with the single-GPU line only
        out["cpu_baseline"] = cpu_baseline()
    print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
