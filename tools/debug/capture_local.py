import sys, os, math
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from conftest import make_suspension
from pse_amd.sharded import LocalLoopbackSimulation
n, grid, world = int(os.environ.get("N", 60000)), int(os.environ.get("GRID", 128)), 8
pos, force, box = make_suspension(n, phi=0.1)
xi = math.pi * grid / (2.0 * box[0] * math.sqrt(-math.log(1e-3)))
sim = LocalLoopbackSimulation(n, box, world, xi=xi, error=1e-3, seed=1, grid=(grid,) * 3)
sim.load(pos, force)
S = sim.s
args = lambda: ([s.pos for s in S], [s.vel for s in S], [s.accel for s in S], [s.image for s in S], [s.force for s in S], [s.tag for s in S], [s.n_local for s in S])
m = 8
for it in range(3):
    sim.team.step_local(*args(), 1.0, 1e-3, it, lanczos_m=m)
torch.cuda.synchronize(); print("eager ok", sim.team.local_status())
if os.environ.get("SOLO"):
    sim.team.debug_solo(3)
st = torch.cuda.Stream()
for e in sim.engines:
    e.set_stream(st.cuda_stream)
word = torch.zeros(1, dtype=torch.int32, device="cuda")
if os.environ.get("TSOFF"):
    for e in sim.engines:
        e.set_timestep_offset(word)
with torch.cuda.stream(st):
    sim.team.step_local(*args(), 1.0, 1e-3, 200, lanczos_m=m)
st.synchronize(); print("stream ok")
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=st, capture_error_mode="thread_local"):
    sim.team.step_local(*args(), 1.0, 1e-3, 200, lanczos_m=m)
print("captured")
for it in range(30):
    word.fill_(it + 1)
    g.replay(); torch.cuda.synchronize()
    print("replay", it, "n_local", [int(s.n_local.item()) for s in S], "m", sim.engines[3].info()["lanczos_m"], flush=True)
print("replays ok")
for it in range(30):
    g.replay()
torch.cuda.synchronize(); print("back-to-back replays ok")
print(sim.team.local_status())
