"""pse_set_async: a deterministic evaluation only queues work -- so the caller can capture it into a hipGraph and replay it
(include/pse_amd.h).  The replay must be the computation itself: new positions and forces written into the
same arrays give the result an eager call gives for them."""
import numpy as np
import pytest

from conftest import make_suspension, to4

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


@pytest.mark.parametrize("n,grid,xy", [(4000, (0, 0, 0), 0.0), (65536, (64, 64, 64), 0.2)])
def test_mobility_captured_into_a_graph_replays_the_computation(n, grid, xy):
    import math
    import torch
    import pse_amd
    pos, force, box = make_suspension(n, phi=0.1, xy=xy)
    kw = dict(error=1e-3, seed=4, grid=grid)
    kw["xi"] = 0.5 if grid[0] == 0 else math.pi * grid[0] / (2.0 * box[0] * math.sqrt(-math.log(1e-3)))
    ref = pse_amd.Engine(n, box, **kw)                     # eager, default mode (keeps its neighbour list)
    eng = pse_amd.Engine(n, box, **kw)
    eng.set_async(True)
    s = torch.cuda.Stream()
    eng.set_stream(s.cuda_stream)
    dpos, dF, vel = to4(pos), to4(force), to4(np.zeros((n, 3)), 7.0)
    with torch.cuda.stream(s):
        eng.mobility(dpos, dF, vel=vel)                    # warm-up outside the capture
    s.synchronize()
    u0 = ref.mobility(to4(pos), to4(force)).cpu().numpy()[:, :3]
    assert rel(vel.cpu().numpy()[:, :3], u0) < 1e-12
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
        eng.mobility(dpos, dF, vel=vel)
    rng = np.random.default_rng(5)
    for trial in range(3):
        # other positions (every particle moved by more than any skin, some across the periodic boundary) and other forces,
        # written into the captured arrays
        pos2 = pos + rng.uniform(-1.5, 1.5, pos.shape)
        f2 = rng.normal(size=force.shape)
        dpos.copy_(to4(pos2)); dF.copy_(to4(f2)); vel[:, :3] = 0.0
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        u = ref.mobility(to4(pos2), to4(f2)).cpu().numpy()[:, :3]
        out = vel.cpu().numpy()
        assert rel(out[:, :3], u) < 1e-12, trial
        assert np.all(out[:, 3] == 7.0)                    # vel.w is preserved


@pytest.mark.parametrize("xy", [0.0, 0.25])
def test_device_side_list_decision_in_a_captured_graph(xy):
    """Asynchronous mode keeps the neighbour list and decides on the device whether a call may reuse it: ONE captured graph replays
    small moves (reuse chain), moves beyond r_buff / 2 (rebuild chain) and small moves again, each time equal to an eager call."""
    import torch
    import pse_amd
    n = 20000
    pos, force, box = make_suspension(n, phi=0.15, xy=xy)
    kw = dict(xi=0.5, error=1e-3, seed=2)
    ref = pse_amd.Engine(n, box, **kw)
    ref.set_neighbor_skin(0.0)                             # the checker rebuilds every call
    eng = pse_amd.Engine(n, box, **kw)
    eng.set_async(True)
    s = torch.cuda.Stream()
    eng.set_stream(s.cuda_stream)
    dpos, dF, vel = to4(pos), to4(force), to4(np.zeros((n, 3)), 1.0)
    with torch.cuda.stream(s):
        eng.mobility(dpos, dF, vel=vel)                    # builds the list (eager, ungated)
        assert eng.debug_last_gate() == -1
        eng.mobility(dpos, dF, vel=vel)                    # two chains, eager: same positions -> reuse
        assert eng.debug_last_gate() == 0
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
        eng.mobility(dpos, dF, vel=vel)
    rng = np.random.default_rng(8)
    cur = pos.copy()
    for trial, (amp, want_rebuild) in enumerate([(0.05, False), (0.05, False), (1.0, True), (0.03, False), (0.9, True), (0.0, False)]):
        cur = cur + rng.uniform(-amp, amp, cur.shape)      # r_buff / 2 = 0.2: 0.05 per axis stays inside, 1.0 does not
        f2 = rng.normal(size=force.shape)
        dpos.copy_(to4(cur)); dF.copy_(to4(f2)); vel[:, :3] = 0.0
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        gate = eng.debug_last_gate()
        assert (gate != 0) == want_rebuild, (trial, gate)
        u = ref.mobility(to4(cur), to4(f2)).cpu().numpy()[:, :3]
        assert rel(vel.cpu().numpy()[:, :3], u) < 1e-12, trial


def test_async_mode_never_reads_back():
    """With asynchronous submission on nothing is read back: the host-side counters cannot tell reuse from rebuild (the device-side
    gate can); with it off again the list is kept and reused the host-side way."""
    import pse_amd
    n = 3000
    pos, force, box = make_suspension(n, phi=0.1)
    eng = pse_amd.Engine(n, box, xi=0.5, error=1e-3, seed=1)
    dpos, dF = to4(pos), to4(force)
    eng.set_async(True)
    u = [eng.mobility(dpos, dF).cpu().numpy()[:, :3] for _ in range(3)]
    _, builds, reuses = eng.neighbor_stats()
    assert builds == 3 and reuses == 0                     # host-side counters: which chain ran is a device-side fact ...
    assert eng.debug_last_gate() == 0                      # ... and it was the reuse chain
    assert rel(u[2], u[0]) < 1e-12
    eng.set_async(False)
    for _ in range(3):
        v = eng.mobility(dpos, dF).cpu().numpy()[:, :3]
    _, builds2, reuses2 = eng.neighbor_stats()
    assert reuses2 >= 1 and rel(v, u[0]) < 1e-12


@pytest.mark.parametrize("n,xy,err", [(6000, 0.0, 1e-3), (20000, 0.2, 1e-3), (3000, 0.0, 1e-6)])
def test_queue_only_brownian_call_takes_the_host_decision_on_the_device(n, xy, err):
    """Asynchronous mode, Brownian calls: the tridiagonal square roots and the step-norm test run in one workgroup on the device
    (k_lz_decide), extra iterations are gated on its outcome.  Same m and same velocities as the host-checked driver, from
    starting counts below, at and above the converged one."""
    import torch
    import pse_amd
    pos, force, box = make_suspension(n, phi=0.15, xy=xy)
    kw = dict(xi=0.5, error=err, seed=11)
    ref = pse_amd.Engine(n, box, **kw)
    eng = pse_amd.Engine(n, box, **kw)
    eng.set_async(True)
    dpos, dF = to4(pos), to4(force)
    u_ref, m_ref = ref.brownian_velocity(dpos, dF, 1.0, 1e-3, 5, lanczos_m=2)
    u_ref = u_ref.cpu().numpy()[:, :3]
    for m_in in (m_ref, m_ref - 1, m_ref - 2, m_ref + 2):
        if m_in < 1:
            continue
        vel, _ = eng.brownian_velocity(dpos, dF, 1.0, 1e-3, 5, lanczos_m=m_in)
        torch.cuda.synchronize()
        i = eng.info()
        # the host-checked driver from the same starting count (it never goes below the count it starts from)
        u2, m2 = ref.brownian_velocity(dpos, dF, 1.0, 1e-3, 5, lanczos_m=m_in)
        assert i["lanczos_status"] == 0 and i["lanczos_m"] == m2, (m_in, i["lanczos_m"], m2, i["lanczos_status"])
        assert rel(vel.cpu().numpy()[:, :3], u2.cpu().numpy()[:, :3]) < 1e-12, m_in
        assert abs(i["lanczos_stepnorm"] - ref.info()["lanczos_stepnorm"]) < 1e-9
    # too few iterations queued: the call says so instead of waiting (status 1), and the result is the one of the last size
    if m_ref >= 6:
        vel, _ = eng.brownian_velocity(dpos, dF, 1.0, 1e-3, 5, lanczos_m=m_ref - 3)
        torch.cuda.synchronize()
        i = eng.info()
        assert i["lanczos_status"] == 1 and i["lanczos_m"] == m_ref - 1
        assert rel(vel.cpu().numpy()[:, :3], u_ref) < 50 * err
    # the starting count the next call should use reaches the host lazily
    _, m_next = eng.brownian_velocity(dpos, dF, 1.0, 1e-3, 5, lanczos_m=m_ref)
    assert m_next >= 1


@pytest.mark.parametrize("xy", [0.0, 0.3])
def test_step_captured_into_a_graph_follows_the_eager_trajectory(xy):
    """pse_step, captured ONCE into a hipGraph and replayed for ten steps with the timestep advanced through a device word
    (pse_set_timestep_offset), against the host-checked engine stepping eagerly: positions, images, equal m."""
    import torch
    import pse_amd
    n = 8000
    pos, force, box = make_suspension(n, phi=0.12, xy=xy)
    kw = dict(xi=0.5, error=1e-3, seed=3)
    ref = pse_amd.Engine(n, box, **kw)
    eng = pse_amd.Engine(n, box, **kw)
    eng.set_async(True)
    s = torch.cuda.Stream()
    eng.set_stream(s.cuda_stream)
    word = torch.zeros(1, dtype=torch.int32, device="cuda")
    eng.set_timestep_offset(word)

    def state():
        return (to4(pos), to4(np.zeros((n, 3)), 1.0), torch.zeros((n, 3), dtype=torch.float64, device="cuda"),
                torch.zeros((n, 3), dtype=torch.int32, device="cuda"), to4(force))
    p1, v1, a1, i1, f1 = state()
    p0, v0, a0, i0, f0 = state()
    kT, dt, ts0, rate = 1.0, 2e-3, 100, 0.4
    # the starting count of the captured call: the converged one of this suspension (an eager call finds it)
    _, m = ref.brownian_velocity(p0, f0, kT, dt, ts0, lanczos_m=2)
    with torch.cuda.stream(s):                                 # warm-up outside the capture, on scratch copies
        q = state()
        eng.step(q[0], q[1], q[2], q[3], q[4], kT, dt, ts0, shear_rate=rate, lanczos_m=m)
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
        eng.step(p1, v1, a1, i1, f1, kT, dt, ts0, shear_rate=rate, lanczos_m=m)
    for k in range(10):
        word.fill_(k)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        mk = ref.step(p0, v0, a0, i0, f0, kT, dt, ts0 + k, shear_rate=rate, lanczos_m=m)
        info = eng.info()
        assert info["lanczos_m"] == mk and info["lanczos_status"] == 0, (k, info["lanczos_m"], mk)
        assert rel(v1.cpu().numpy()[:, :3], v0.cpu().numpy()[:, :3]) < 1e-12, k
        assert np.abs(p1.cpu().numpy()[:, :3] - p0.cpu().numpy()[:, :3]).max() < 1e-12, k
        assert torch.equal(i1, i0)


def test_queue_only_steps_without_the_gated_extras_and_the_sticky_count_of_steps_that_ran_out():
    """pse_set_lanczos_extra(0) (the steady state of a time-stepping loop, pse_amd.sharded.LanczosCount): a queue-only call queues no
    gated iteration; from the converged starting count the result is the one of the default form, bit for bit, one Lanczos mat-vec
    and no decision more is launched than needed; from too low a count the call says so (status 1) and pse_info.lanczos_open_calls,
    which is sticky, counts it -- also when pse_info is read only after later, converged calls."""
    import torch
    import pse_amd
    from pse_amd.sharded import LanczosCount
    n = 20000
    pos, force, box = make_suspension(n, phi=0.15)
    eng = pse_amd.Engine(n, box, xi=0.5, error=1e-3, seed=11)
    eng.set_async(True)
    dpos, dF = to4(pos), to4(force)
    lc = LanczosCount(eng, m=2, settle=3)
    for it in range(12):                                   # the policy finds the count, then switches the extras off
        eng.brownian_velocity(dpos, dF, 1.0, 1e-3, 5, lanczos_m=lc.m)
        torch.cuda.synchronize()
        i = eng.info()
        lc.seen(i["lanczos_m"], i["lanczos_status"])
        if lc.extras_off:
            break
    assert lc.extras_off and lc.m >= 4
    open0 = eng.info()["lanczos_open_calls"]
    v_off, _ = eng.brownian_velocity(dpos, dF, 1.0, 1e-3, 5, lanczos_m=lc.m)
    torch.cuda.synchronize()
    i_off = eng.info()
    v_off = v_off.clone()
    eng.set_lanczos_extra(-1)
    v_on, _ = eng.brownian_velocity(dpos, dF, 1.0, 1e-3, 5, lanczos_m=lc.m)
    torch.cuda.synchronize()
    i_on = eng.info()
    assert i_off["lanczos_status"] == 0 and i_off["lanczos_m"] == lc.m == i_on["lanczos_m"] and i_off["lanczos_open_calls"] == open0
    assert rel(v_off.cpu().numpy()[:, :3], v_on.cpu().numpy()[:, :3]) < 1e-12
    assert i_off["lanczos_matvecs"] <= i_on["lanczos_matvecs"]
    # too low a count without extras: status 1, counted; two converged calls later the count still says so
    eng.set_lanczos_extra(0)
    eng.brownian_velocity(dpos, dF, 1.0, 1e-3, 5, lanczos_m=lc.m - 2)
    torch.cuda.synchronize()
    assert eng.info()["lanczos_status"] == 1
    eng.set_lanczos_extra(-1)
    for _ in range(2):
        eng.brownian_velocity(dpos, dF, 1.0, 1e-3, 5, lanczos_m=lc.m)
    torch.cuda.synchronize()
    i = eng.info()
    assert i["lanczos_status"] == 0 and i["lanczos_open_calls"] == open0 + 1
