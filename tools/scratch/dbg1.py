import numpy as np, math, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
from conftest import make_suspension, to4
from oracle import pse_port as pp
import pse_amd, torch
def rel(a,b): return np.linalg.norm(a-b)/np.linalg.norm(b)
n=1000
pos, force, box = make_suspension(n, phi=0.1, xy=0.0)
seed, ts, kT, dt = 424242, 11, 1.0, 1e-3
eng = pse_amd.Engine(n, box, xi=0.5, error=1e-3, seed=seed)
p = pp.select_params(box, 0.5, 1e-3, 0.5)
print(p['grid'], p['P'])
zero = np.zeros_like(force)
vel, m = eng.brownian_velocity(to4(pos), to4(zero), kT, dt, ts)
ub = vel.cpu().numpy()[:, :3]
# HIP Lanczos piece
psi = eng.random_psi(n, ts)
lz, m2 = eng.sqrt_mreal(to4(pos), psi, tol=1e-3)
lz = lz.cpu().numpy()[:, :3]*math.sqrt(2*kT/dt)
hip_wave_noise = ub - lz
# port pieces
nk = pp.noise_k(box, p, kT, dt, seed, ts)
ug = np.fft.irfftn(nk, s=p['grid'], axes=(1,2,3), norm='forward')
port_wave_noise = pp.gather(ug, pos, box, p)
psi_p = pp.psi_particles(n, seed, ts)
mv = lambda v: pp.mobility_real(pos, np.ascontiguousarray(v), box, p['xi'], p['rcut'])
lzp, mp = pp.lanczos_sqrt(mv, psi_p, 2, 1e-3); lzp = lzp*math.sqrt(2*kT/dt)
print('m', m, m2, mp)
print('lanczos piece rel', rel(lz, lzp))
print('wave noise piece rel', rel(hip_wave_noise, port_wave_noise), np.linalg.norm(port_wave_noise), np.linalg.norm(lzp))
# compare grids after inverse FFT
g = eng.debug_grid()
print('grid rel', rel(g, ug), np.abs(g-ug).max(), np.abs(ug).max())
d = np.abs(g-ug)
print('where max', np.unravel_index(d.argmax(), d.shape))
# spectrum of the difference
D = np.fft.rfftn(g-ug, axes=(1,2,3))
a = np.abs(D[0]); idx = np.argsort(a.ravel())[::-1][:10]
for t in idx: print(np.unravel_index(t, a.shape), a.ravel()[t])
