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
