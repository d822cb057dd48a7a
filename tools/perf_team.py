#!/usr/bin/env python3
"""Per-rank cost of the slab-decomposed step from an in-process team (all G ranks on one GPU, copies instead of RCCL): the
kernels of the G ranks run one after another, so (time of a team step) / G is the per-rank COMPUTE time of a G-GPU run; link
time is added on paper.  python3 tools/perf_team.py [--ranks 8] [--n 1000000] [--grid 256] [--steps 5]"""
import argparse
import math
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--phi", type=float, default=0.1)
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--solo", type=int, default=-1, help="after the full calls: time Brownian evaluations that queue the work of this "
                    "rank only (both lanes, its share of the copies): one rank's critical path with the GPU to itself")
    ap.add_argument("--local", action="store_true", help="owned-particle team (pse_team_step_local): every rank holds only its slab's particles")
    ap.add_argument("--m", type=int, default=0, help="--local: starting count of the Lanczos iteration (default: found by warm-up steps)")
    ap.add_argument("--extra", type=int, default=-1, help="--local: pse_team_set_lanczos_extra for the timed steps (0: the steady state of a "
                    "time-stepping loop -- no gated block queued; -1: the default)")
    a = ap.parse_args()
    import torch
    from conftest import make_suspension
    if a.local:
        return main_local(a)
    from pse_amd.sharded import LoopbackSimulation
    pos, force, box = make_suspension(a.n, phi=a.phi)
    xi = math.pi * a.grid / (2.0 * box[0] * math.sqrt(-math.log(1e-3)))
    sim = LoopbackSimulation(a.n, box, a.ranks, xi=xi, error=1e-3, seed=1, grid=(a.grid,) * 3)
    sim.load(pos, force)
    m = 2
    for it in range(3):
        m = sim.step(1.0, 1e-3, it, lanczos_m=m)
    torch.cuda.synchronize(); t0 = time.time()
    for it in range(a.steps):
        m = sim.step(1.0, 1e-3, 10 + it, lanczos_m=m)
    torch.cuda.synchronize(); t = (time.time() - t0) / a.steps
    print(f"team of {a.ranks}: {t * 1e3:.3f} ms per team step -> {t * 1e3 / a.ranks:.3f} ms per rank (compute only), m={m}")
    torch.cuda.synchronize(); t0 = time.time()
    for it in range(a.steps):
        sim.mobility()
    torch.cuda.synchronize(); t = (time.time() - t0) / a.steps
    print(f"team of {a.ranks}: M.F {t * 1e3:.3f} ms per team eval -> {t * 1e3 / a.ranks:.3f} ms per rank")
    if a.solo >= 0:
        # a full Brownian evaluation at fixed positions leaves every member's buffers in place; then only rank `solo` works
        vels, m = sim.brownian_velocity(1.0, 1e-3, 50, lanczos_m=m)
        sim.team.debug_solo(a.solo)
        for it in range(3):
            sim.brownian_velocity(1.0, 1e-3, 50, lanczos_m=m)
        torch.cuda.synchronize()
        ts = []
        for it in range(max(a.steps, 20)):
            t0 = time.perf_counter()
            sim.brownian_velocity(1.0, 1e-3, 50, lanczos_m=m)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        i = sim.engines[a.solo].info()
        print(f"solo rank {a.solo} of {a.ranks}: Brownian evaluation (no Euler update) {ts[len(ts) // 2] * 1e3:.3f} ms median, "
              f"{ts[0] * 1e3:.3f} min, {ts[-1] * 1e3:.3f} max over {len(ts)} calls; m = {m}, exchanges = {i['lanczos_exchanges']}, "
              f"mat-vecs = {i['lanczos_matvecs']}")
        sim.team.debug_solo(-1)
        return
    for e in sim.engines[:1]:
        e.set_timing(True)
    sim.engines[0].set_timing(True)
    m = sim.step(1.0, 1e-3, 99, lanczos_m=m)
    i = sim.engines[0].info()
    print("rank 0 phases ms:", {k: round(v, 4) for k, v in i.items() if k.startswith("t_") and v > 0})
    print(f"Lanczos: m = {i['lanczos_m']}, near-field mat-vecs = {i['lanczos_matvecs']}, exchanges = {i['lanczos_exchanges']}")


def main_local(a):
    import torch
    from conftest import make_suspension
    from pse_amd.sharded import LocalLoopbackSimulation
    pos, force, box = make_suspension(a.n, phi=a.phi)
    xi = math.pi * a.grid / (2.0 * box[0] * math.sqrt(-math.log(1e-3)))
    sim = LocalLoopbackSimulation(a.n, box, a.ranks, xi=xi, error=1e-3, seed=1, grid=(a.grid,) * 3)
    sim.load(pos, force)
    print("layout", sim.layout, "row capacity per rank", sim.engines[0].params.n_max, "device GB per rank %.2f" % (sim.engines[0].info()["device_bytes"] / 1e9))
    S = sim.s
    args = lambda: ([s.pos for s in S], [s.vel for s in S], [s.accel for s in S], [s.image for s in S], [s.force for s in S],   # noqa: E731
                    [s.tag for s in S], [s.n_local for s in S])
    m = a.m or 2   # the starting count grows until a step converges within its queue (a queue-only step never waits for more)
    for it in range(0 if a.m else 8):
        sim.step(1.0, 1e-3, it, lanczos_m=m)
        torch.cuda.synchronize()
        i0 = sim.engines[0].info()
        m = max(i0["lanczos_m"], 2)
        if i0["lanczos_status"] == 0:
            break
    print("m =", m, "status", [e.info()["lanczos_status"] for e in sim.engines], "n_local", [int(s.n_local.item()) for s in S])
    sim.team.set_lanczos_extra(a.extra)
    torch.cuda.synchronize(); t0 = time.time()
    for it in range(a.steps):
        sim.team.step_local(*args(), 1.0, 1e-3, 10 + it, lanczos_m=m)
    torch.cuda.synchronize(); t = (time.time() - t0) / a.steps
    print(f"local team of {a.ranks}: {t * 1e3:.3f} ms per team step -> {t * 1e3 / a.ranks:.3f} ms per rank (compute only), m={m}")
    sim.team.local_status()
    if a.solo >= 0:
        # The solo rank re-executes ITS part of one full team step: its arrays are put back to what they were before that step every
        # time (the messages of the frozen neighbours are those of that step: replaying them on any other state would add the same
        # arrivals again and again)
        snap = S[a.solo].snapshot()
        sim.team.step_local(*args(), 1.0, 1e-3, 49, lanczos_m=m)
        torch.cuda.synchronize()
        sim.team.debug_solo(a.solo)
        for it in range(3):
            S[a.solo].restore(snap)
            sim.team.step_local(*args(), 1.0, 1e-3, 49, lanczos_m=m)
        torch.cuda.synchronize()
        ts = []
        for it in range(max(a.steps, 30)):
            S[a.solo].restore(snap)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sim.team.step_local(*args(), 1.0, 1e-3, 49, lanczos_m=m)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        i = sim.engines[a.solo].info()
        print(f"solo rank {a.solo} of {a.ranks} (owned particles): full step incl. Euler update {ts[len(ts) // 2] * 1e3:.3f} ms median, "
              f"{ts[0] * 1e3:.3f} min, {ts[-1] * 1e3:.3f} max over {len(ts)} calls; m = {i['lanczos_m']}, exchanges = {i['lanczos_exchanges']}, "
              f"mat-vecs = {i['lanczos_matvecs']}")
        torch.cuda.synchronize(); t0 = time.perf_counter()
        nrep = max(a.steps, 30)
        th = 0.0
        for it in range(nrep):
            S[a.solo].restore(snap)
            h0 = time.perf_counter()
            sim.team.step_local(*args(), 1.0, 1e-3, 49, lanczos_m=m)
            th += time.perf_counter() - h0
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / nrep
        print(f"solo rank {a.solo}: {t * 1e3:.3f} ms per step with the calls queued back to back (no wait between steps; incl. seven small "
              f"copies that put the rank's arrays back); the host needs {th / nrep * 1e3:.3f} ms to queue one")
        # the same step as ONE hipGraph: the whole call only queues work, so it can be captured (both lanes, every exchange)
        st = torch.cuda.Stream()
        for e in sim.engines:
            e.set_stream(st.cuda_stream)
        word = torch.zeros(1, dtype=torch.int32, device="cuda")
        for e in sim.engines:
            e.set_timestep_offset(word)
        S[a.solo].restore(snap)
        with torch.cuda.stream(st):
            sim.team.step_local(*args(), 1.0, 1e-3, 49, lanczos_m=m)
        st.synchronize()
        S[a.solo].restore(snap)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st, capture_error_mode="thread_local"):
            sim.team.step_local(*args(), 1.0, 1e-3, 49, lanczos_m=m)
        ts = []
        for it in range(nrep + 3):
            S[a.solo].restore(snap)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            g.replay()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        ts = sorted(ts[3:])
        i = sim.engines[a.solo].info()
        print(f"solo rank {a.solo}: the step as one replayed hipGraph {ts[len(ts) // 2] * 1e3:.3f} ms median, {ts[0] * 1e3:.3f} min; m = {i['lanczos_m']} "
              f"status {i['lanczos_status']}")
        for e in sim.engines:
            e.set_stream(0)
            e.set_timestep_offset(None)
        sim.team.debug_solo(-1)


if __name__ == "__main__":
    main()
