"""Thin Python owner of a pse_handle: torch tensors in, torch tensors out, everything through the C-ABI.

Arrays follow HOOMD's layout with Scalar = double: pos/vel/force are (N,4) float64 CUDA tensors
(x,y,z,type|mass|energy), accel (N,3) float64, image (N,3) int32, group_members (N,) int32/uint32.
"""
import ctypes

from . import _lib
from ._lib import pse_info, pse_params


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _chk4(t, name, n=None):
    import torch
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float64 and t.dim() == 2
            and t.shape[1] == 4 and t.is_contiguous()):
        raise ValueError(f"{name} must be a contiguous (N,4) float64 CUDA tensor")
    if n is not None and t.shape[0] < n:
        raise ValueError(f"{name} has fewer than {n} rows")


def _chk_group(g, name="group"):
    import torch
    if g is None:
        return
    if not (isinstance(g, torch.Tensor) and g.is_cuda and g.dtype in (torch.int32, torch.uint32) and g.dim() == 1
            and g.is_contiguous()):
        raise ValueError(f"{name} must be a contiguous 1-D int32/uint32 CUDA tensor (the C-ABI reads unsigned int indices)")


def _chk_arr(t, name, cols, dtype, n):
    import torch
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == dtype and t.dim() == 2 and t.shape[1] == cols
            and t.is_contiguous() and t.shape[0] >= n):
        raise ValueError(f"{name} must be a contiguous (N>={n},{cols}) {dtype} CUDA tensor")


def host_select_params(box, xi=0.5, error=1e-3, max_strain=0.5, grid=(0, 0, 0), P=0, rcut=0.0):
    """Parameter rule of Stokes::setParams (PSEv1/Stokes.cc:129-236,319), host only."""
    lib = _lib.load()
    p = pse_params(n_max=1, Lx=box[0], Ly=box[1], Lz=box[2], xy=box[3] if len(box) > 3 else 0.0, xi=xi, error=error,
                   max_strain=max_strain, seed=0, Nx=grid[0], Ny=grid[1], Nz=grid[2], P=P, rcut=rcut, device=-1,
                   n_slabs=1, slab_rank=0)
    info = pse_info()
    _lib.check(lib.pse_host_select_params(ctypes.byref(p), ctypes.byref(info)))
    return info.as_dict()


def host_lanczos_sqrt_e1(alpha, beta):
    """t = T^{1/2} e_1 (host); alpha[0..m), beta[0..m] with beta[0] unused."""
    import numpy as np
    lib = _lib.load()
    a = np.ascontiguousarray(alpha, dtype=np.float64)
    b = np.ascontiguousarray(beta, dtype=np.float64)
    t = np.zeros(len(a))
    dp = ctypes.POINTER(ctypes.c_double)
    _lib.check(lib.pse_host_lanczos_sqrt_e1(len(a), a.ctypes.data_as(dp), b.ctypes.data_as(dp), t.ctypes.data_as(dp)))
    return t


class Engine:
    """One PSE engine instance == one `Stokes` object's device state (PSEv1/Stokes.h:128-150)."""

    def __init__(self, n_max, box, xi=0.5, error=1e-3, max_strain=0.5, seed=0, grid=(0, 0, 0), P=0, rcut=0.0,
                 device=-1, n_slabs=1, slab_rank=0, local_rows=0):
        self._lib = _lib.load()
        self._h = ctypes.c_void_p()
        box = tuple(float(b) for b in box) + ((0.0,) if len(box) == 3 else ())
        self.params = pse_params(n_max=int(n_max), Lx=box[0], Ly=box[1], Lz=box[2], xy=box[3], xi=xi, error=error,
                                 max_strain=max_strain, seed=int(seed) & 0xFFFFFFFF, Nx=grid[0], Ny=grid[1],
                                 Nz=grid[2], P=P, rcut=rcut, device=device, n_slabs=n_slabs, slab_rank=slab_rank,
                                 local_rows=int(local_rows))
        _lib.check(self._lib.pse_create(ctypes.byref(self.params), ctypes.byref(self._h)))
        self.box = box

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.pse_destroy(self._h)
            self._h = ctypes.c_void_p()

    __del__ = close

    def info(self):
        out = pse_info()
        _lib.check(self._lib.pse_get_info(self._h, ctypes.byref(out)))
        return out.as_dict()

    def local_layout(self):
        """Row capacities of an owned-particle handle (local_rows=1): dict rows_own (capacity of the caller's arrays), rows_ghost,
        records (per neighbour message), layers (cell layers along x), layers_per_rank."""
        v = [ctypes.c_int() for _ in range(5)]
        _lib.check(self._lib.pse_local_layout(self._h, *[ctypes.byref(x) for x in v]))
        return dict(zip(("rows_own", "rows_ghost", "records", "layers", "layers_per_rank"), [x.value for x in v]))

    def set_box(self, Lx, Ly, Lz, xy):
        _lib.check(self._lib.pse_set_box(self._h, Lx, Ly, Lz, xy))
        self.box = (Lx, Ly, Lz, xy)

    def set_timing(self, on=True):
        _lib.check(self._lib.pse_set_timing(self._h, 1 if on else 0))

    def set_async(self, on=True):
        """Deterministic evaluations only queue work (no read-back, capturable into a hipGraph); reuse or rebuild of the kept
        neighbour list is decided on the device."""
        _lib.check(self._lib.pse_set_async(self._h, 1 if on else 0))

    def set_timestep_offset(self, word):
        """word: a 1-element int32/uint32 CUDA tensor (or None): Brownian calls draw their noise at timestep + word[0], read on the
        device -- what lets a captured step replay with fresh noise.  The tensor must outlive the registration."""
        self._ts_word = word
        _lib.check(self._lib.pse_set_timestep_offset(self._h, _ptr(word)))

    def debug_last_gate(self):
        """0: the last asynchronous evaluation reused the kept list, != 0: it rebuilt, -1: it did not take the two-chain path."""
        g = ctypes.c_int(-2)
        _lib.check(self._lib.pse_debug_last_gate(self._h, ctypes.byref(g)))
        return g.value

    def set_neighbor_skin(self, r_buff):
        """r_buff of the neighbour list kept across calls (HOOMD's nlist r_buff, PSEv1/integrate.py:60); 0 rebuilds every call."""
        _lib.check(self._lib.pse_set_neighbor_skin(self._h, float(r_buff)))

    def neighbor_stats(self):
        """(r_buff, builds, reuses) of the kept neighbour list."""
        r, b, u = ctypes.c_double(), ctypes.c_ulonglong(), ctypes.c_ulonglong()
        _lib.check(self._lib.pse_neighbor_stats(self._h, ctypes.byref(r), ctypes.byref(b), ctypes.byref(u)))
        return r.value, b.value, u.value

    def set_stream(self, stream_ptr):
        _lib.check(self._lib.pse_set_stream(self._h, ctypes.c_void_p(stream_ptr)))

    # -- hot path -------------------------------------------------------------------------------------------
    def mobility(self, pos, force, vel=None, group=None, parts=3):
        import torch
        n = pos.shape[0] if group is None else group.shape[0]
        _chk4(pos, "pos"); _chk4(force, "force"); _chk_group(group)
        if vel is None:
            vel = torch.zeros_like(pos)
        _chk4(vel, "vel")
        _lib.check(self._lib.pse_mobility(self._h, _ptr(pos), _ptr(force), _ptr(vel), _ptr(group), n, parts))
        return vel

    def brownian_velocity(self, pos, force, kT, dt, timestep, vel=None, group=None, lanczos_m=2):
        import torch
        n = pos.shape[0] if group is None else group.shape[0]
        _chk4(pos, "pos"); _chk4(force, "force"); _chk_group(group)
        if vel is None:
            vel = torch.zeros_like(pos)
        _chk4(vel, "vel")
        m = ctypes.c_int(int(lanczos_m))
        _lib.check(self._lib.pse_brownian_velocity(self._h, _ptr(pos), _ptr(force), _ptr(vel), _ptr(group), n,
                                                   float(kT), float(dt), int(timestep), ctypes.byref(m)))
        return vel, m.value

    def brownian_velocity_part(self, pos, force, kT, dt, timestep, parts, vel=None, group=None, lanczos_m=2):
        """One half of brownian_velocity (pse_brownian_velocity_part): parts = 1 real space + Lanczos noise, 2 wave space + k-space noise."""
        import torch
        n = pos.shape[0] if group is None else group.shape[0]
        _chk4(pos, "pos"); _chk4(force, "force"); _chk_group(group)
        if vel is None:
            vel = torch.zeros_like(pos)
        _chk4(vel, "vel")
        m = ctypes.c_int(int(lanczos_m))
        _lib.check(self._lib.pse_brownian_velocity_part(self._h, _ptr(pos), _ptr(force), _ptr(vel), _ptr(group), n, float(kT), float(dt),
                                                        int(timestep), int(parts), ctypes.byref(m)))
        return vel, m.value

    def integrate(self, pos, vel, accel, image, force, dt, shear_rate=0.0, group=None):
        """The Euler update + wrap alone (pse_integrate) for velocities the caller has put together."""
        import torch
        n = pos.shape[0] if group is None else group.shape[0]
        _chk4(pos, "pos"); _chk4(vel, "vel"); _chk4(force, "force"); _chk_group(group)
        _chk_arr(accel, "accel", 3, torch.float64, pos.shape[0]); _chk_arr(image, "image", 3, torch.int32, pos.shape[0])
        _lib.check(self._lib.pse_integrate(self._h, _ptr(pos), _ptr(vel), _ptr(accel), _ptr(image), _ptr(force), _ptr(group), n, float(dt),
                                           float(shear_rate)))

    def step(self, pos, vel, accel, image, force, kT, dt, timestep, shear_rate=0.0, group=None, lanczos_m=2):
        n = pos.shape[0] if group is None else group.shape[0]
        import torch
        _chk4(pos, "pos"); _chk4(vel, "vel"); _chk4(force, "force"); _chk_group(group)
        _chk_arr(accel, "accel", 3, torch.float64, pos.shape[0]); _chk_arr(image, "image", 3, torch.int32, pos.shape[0])
        m = ctypes.c_int(int(lanczos_m))
        _lib.check(self._lib.pse_step(self._h, _ptr(pos), _ptr(vel), _ptr(accel), _ptr(image), _ptr(force),
                                      _ptr(group), n, float(kT), float(dt), int(timestep), float(shear_rate),
                                      ctypes.byref(m)))
        return m.value

    def sqrt_mreal(self, pos, psi, tol=1e-3, group=None, lanczos_m=2):
        import torch
        n = pos.shape[0] if group is None else group.shape[0]
        _chk4(pos, "pos"); _chk4(psi, "psi"); _chk_group(group)
        out = torch.zeros_like(psi)
        m = ctypes.c_int(int(lanczos_m))
        _lib.check(self._lib.pse_sqrt_mreal(self._h, _ptr(pos), _ptr(psi), _ptr(out), _ptr(group), n, float(tol),
                                            ctypes.byref(m)))
        return out, m.value

    def pair_repulsion(self, pos, force, k, sigma=2.0, group=None, accumulate=True):
        """Soft repulsion k (sigma - r) r_hat for r < sigma added to (or stored in) `force`."""
        n = pos.shape[0] if group is None else group.shape[0]
        _chk4(pos, "pos"); _chk4(force, "force"); _chk_group(group)
        _lib.check(self._lib.pse_pair_repulsion(self._h, _ptr(pos), _ptr(force), _ptr(group), n, float(k), float(sigma),
                                                1 if accumulate else 0))
        return force

    def random_psi(self, n, timestep, group=None):
        import torch
        rows = n if group is None else int(group.max().item()) + 1
        psi = torch.zeros((rows, 4), dtype=torch.float64, device="cuda")
        _lib.check(self._lib.pse_random_psi(self._h, _ptr(psi), _ptr(group), n, int(timestep)))
        return psi

    # -- introspection ---------------------------------------------------------------------------------------
    def eval_realspace(self, r):
        import numpy as np
        r = np.ascontiguousarray(r, dtype=np.float64)
        f = np.zeros_like(r); g = np.zeros_like(r)
        dp = ctypes.POINTER(ctypes.c_double)
        _lib.check(self._lib.pse_eval_realspace(self._h, r.ctypes.data_as(dp), len(r), f.ctypes.data_as(dp),
                                                g.ctypes.data_as(dp)))
        return f, g

    def debug_spread(self, pos, force, group=None):
        """The three force grids right after the spread (3, Nx, Ny, Nz), host array."""
        n = pos.shape[0] if group is None else group.shape[0]
        _chk4(pos, "pos"); _chk4(force, "force"); _chk_group(group)
        _lib.check(self._lib.pse_debug_spread(self._h, _ptr(pos), _ptr(force), _ptr(group), n))
        return self.debug_grid()

    def debug_kvector(self, ijk):
        """(n, 5): kx, ky, kz, w sinc^2, sqrt(w) sinc of the grid nodes ijk (n, 3) as the k-space kernels evaluate them."""
        import numpy as np
        ijk = np.ascontiguousarray(ijk, dtype=np.int32)
        out = np.zeros((len(ijk), 5))
        _lib.check(self._lib.pse_debug_kvector(self._h, len(ijk), ijk.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
                                               out.ctypes.data_as(ctypes.POINTER(ctypes.c_double))))
        return out

    def set_lanczos_extra(self, extra):
        """Gated iterations a queue-only Brownian call queues beyond its starting count (pse_set_lanczos_extra; -1: the default, 0: none --
        for a time-stepping loop whose steps end at their starting count: pse_amd.sharded.LanczosCount)."""
        _lib.check(self._lib.pse_set_lanczos_extra(self._h, int(extra)))

    def matvec_ms(self, reps=20):
        """Milliseconds per launch of the pair-list mat-vec of a Lanczos iteration, `reps` launches back to back between one pair of
        events (pse_debug_matvec_ms): right after a Brownian call of a single-GPU engine."""
        ms = ctypes.c_float(0)
        _lib.check(self._lib.pse_debug_matvec_ms(self._h, int(reps), ctypes.byref(ms)))
        return ms.value

    def grid_placement(self):
        """What pse_create's grid-placement planner did: {"tried": pairs timed (0: off or not applicable), "ms_first", "ms_kept": the
        x pass + inverse y + z passes on the first pair allocated and on the one kept}."""
        n, a, b = ctypes.c_int(0), ctypes.c_float(0), ctypes.c_float(0)
        _lib.check(self._lib.pse_debug_grid_placement(self._h, ctypes.byref(n), ctypes.byref(a), ctypes.byref(b)))
        return {"tried": n.value, "ms_first": a.value, "ms_kept": b.value}

    def debug_grid(self):
        import numpy as np
        i = self.info()
        out = np.zeros((3, i["Nx"] // max(1, self.params.n_slabs), i["Ny"], i["Nz"]))
        _lib.check(self._lib.pse_debug_copy_grid(self._h, 0, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double))))
        return out


class Team:
    """A pse_team: the local slab ranks bound to a transport (RCCL across processes, or in-process loopback)."""

    def __init__(self, engines, unique_id=None, transport=None):
        """transport: an object with exchange(xfers) and allreduce_sum(array) (see pse_amd.sharded.TorchTransport): the
        host-staged third transport of include/pse_amd.h, one member per process."""
        self._lib = _lib.load()
        self.engines = list(engines)
        self._t = ctypes.c_void_p()
        if transport is not None:
            self._transport = transport              # the callbacks must outlive the team
            self._cb = _lib.pse_transport(None, _lib.EXCHANGE_FN(self._exchange), _lib.ALLREDUCE_FN(self._allreduce))
            _lib.check(self._lib.pse_team_create_transport(self.engines[0]._h, ctypes.byref(self._cb), ctypes.byref(self._t)))
            return
        arr = (ctypes.c_void_p * len(self.engines))(*[e._h for e in self.engines])
        idbuf = ctypes.create_string_buffer(bytes(unique_id), 128) if unique_id is not None else None
        _lib.check(self._lib.pse_team_create(arr, len(self.engines), idbuf, ctypes.byref(self._t)))

    def _exchange(self, _user, n, xfers):
        import numpy as np
        try:
            ops = []
            for q in range(n):
                x = xfers[q]
                send = np.ctypeslib.as_array(x.send, shape=(x.send_count,)) if x.send_count else None
                recv = np.ctypeslib.as_array(x.recv, shape=(x.recv_count,)) if x.recv_count else None
                ops.append((send, x.send_to, recv, x.recv_from))
            self._transport.exchange(ops)
            return 0
        except Exception as e:   # noqa: BLE001  (an exception must not unwind through the C caller)
            print("pse_amd transport: exchange failed:", repr(e))
            return 1

    def _allreduce(self, _user, buf, n):
        import numpy as np
        try:
            self._transport.allreduce_sum(np.ctypeslib.as_array(buf, shape=(n,)))
            return 0
        except Exception as e:   # noqa: BLE001
            print("pse_amd transport: all-reduce failed:", repr(e))
            return 1

    @staticmethod
    def unique_id():
        buf = ctypes.create_string_buffer(128)
        _lib.check(_lib.load().pse_team_unique_id(buf))
        return buf.raw

    def debug_solo(self, slab_rank):
        """Developer switch: queue the work of one member only (see include/pse_amd.h); -1 switches it off."""
        _lib.check(self._lib.pse_team_debug_solo(self._t, int(slab_rank)))

    def close(self):
        if getattr(self, "_t", None) is not None and self._t.value:
            self._lib.pse_team_destroy(self._t)
            self._t = ctypes.c_void_p()

    __del__ = close

    @staticmethod
    def _ptrs(tensors):
        return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])

    @staticmethod
    def _chk(group, *lists):
        _chk_group(group)
        for name, ts in lists:
            for t in ts:
                _chk4(t, name)

    def mobility(self, pos, force, vel, group=None, parts=3):
        n = pos[0].shape[0] if group is None else group.shape[0]
        self._chk(group, ("pos", pos), ("force", force), ("vel", vel))
        _lib.check(self._lib.pse_team_mobility(self._t, self._ptrs(pos), self._ptrs(force), self._ptrs(vel), _ptr(group),
                                               n, parts))
        return vel

    def brownian_velocity(self, pos, force, vel, kT, dt, timestep, group=None, lanczos_m=2):
        n = pos[0].shape[0] if group is None else group.shape[0]
        self._chk(group, ("pos", pos), ("force", force), ("vel", vel))
        m = ctypes.c_int(int(lanczos_m))
        _lib.check(self._lib.pse_team_brownian_velocity(self._t, self._ptrs(pos), self._ptrs(force), self._ptrs(vel),
                                                        _ptr(group), n, float(kT), float(dt), int(timestep),
                                                        ctypes.byref(m)))
        return vel, m.value

    def step_local(self, pos, vel, accel, image, force, tag, n_local, kT, dt, timestep, shear_rate=0.0, integrate=True, lanczos_m=2):
        """Owned-particle step (pse_team_step_local): per member, rows [0, n_local[0]) of the arrays are the particles the rank owns;
        tag (int32, global indices) and n_local (1-element int32) are CUDA tensors, rewritten with the arrays.  Queue-only."""
        import torch
        for t in list(pos) + list(vel) + list(force):
            _chk4(t, "pos/vel/force")
        self._check_local_arrays(pos, vel, force, accel, image, tag)
        for a_, im_, tg, nl, p_ in zip(accel, image, tag, n_local, pos):
            _chk_arr(a_, "accel", 3, torch.float64, p_.shape[0]); _chk_arr(im_, "image", 3, torch.int32, p_.shape[0])
            if not (tg.is_cuda and tg.dtype == torch.int32 and tg.is_contiguous() and tg.shape[0] >= p_.shape[0]):
                raise ValueError("tag must be a contiguous int32 CUDA tensor with a row per particle slot")
            if not (nl.is_cuda and nl.dtype == torch.int32 and nl.numel() == 1):
                raise ValueError("n_local must be a 1-element int32 CUDA tensor")
        m = ctypes.c_int(int(lanczos_m))
        _lib.check(self._lib.pse_team_step_local(self._t, self._ptrs(pos), self._ptrs(vel), self._ptrs(accel), self._ptrs(image),
                                                 self._ptrs(force), self._ptrs(tag), self._ptrs(n_local), float(kT), float(dt),
                                                 int(timestep), float(shear_rate), 1 if integrate else 0, ctypes.byref(m)))
        return m.value

    DIAG_KINDS = {0: "migrate_ghosts", 1: "lanczos", 2: "all_to_all", 3: "halo", 4: "all_gather", 5: "ghost_rows"}

    def set_diag(self, on=True):
        """Bracket every exchange of the team's calls with events (pse_team_set_diag)."""
        _lib.check(self._lib.pse_team_set_diag(self._t, 1 if on else 0))

    def diag(self):
        """The exchanges of the last call (waits for it): dict with exchanges_per_step, exchange_us {kind: [device us, ...]},
        exchange_host_us, exchange_bytes, lanes_ms, critical_path_ms."""
        d = _lib.pse_team_diag()
        _lib.check(self._lib.pse_team_get_diag(self._t, ctypes.byref(d)))
        out = {"exchanges_per_step": d.n_exchanges, "exchange_us": {}, "exchange_host_us": {}, "exchange_bytes": {},
               "lanes_ms": {"main": d.main_lane_ms, "side": d.side_lane_ms}, "critical_path_ms": d.critical_path_ms}
        for q in range(d.n_exchanges):
            k = self.DIAG_KINDS.get(d.kind[q], str(d.kind[q]))
            out["exchange_us"].setdefault(k, []).append(round(d.device_us[q], 2))
            out["exchange_host_us"].setdefault(k, []).append(round(d.host_us[q], 2))
            out["exchange_bytes"].setdefault(k, []).append(int(d.bytes[q]))
        return out

    def redistribute_local(self, pos, vel, accel, image, force, tag, n_local):
        """After set_box has taken every member's tilt through a Lees-Edwards flip: every particle to the rank that owns it under the
        new box (pse_team_redistribute_local; arrays as for step_local, force rewritten too).  Waits for the streams twice."""
        self._check_local_arrays(pos, vel, force, accel, image, tag)
        _lib.check(self._lib.pse_team_redistribute_local(self._t, self._ptrs(pos), self._ptrs(vel), self._ptrs(accel), self._ptrs(image),
                                                         self._ptrs(force), self._ptrs(tag), self._ptrs(n_local)))

    def _check_local_arrays(self, pos, vel, force, accel, image, tag):
        if getattr(self, "_rows_own", None) is None:     # a step writes up to rows_own rows back (particles migrate in): the arrays must hold them
            self._rows_own = [e.local_layout()["rows_own"] for e in self.engines]
        for k, ts in enumerate(zip(pos, vel, force, accel, image, tag)):
            if any(t.shape[0] < self._rows_own[k] for t in ts):
                raise ValueError(f"member {k}: pos / vel / force / accel / image / tag need {self._rows_own[k]} rows (rows_own of pse_local_layout), "
                                 f"not the current particle count")

    def set_lanczos_extra(self, extra):
        """Iterations an owned-particle step queues beyond its starting count (pse_team_set_lanczos_extra; -1: the default, gated
        on the device-side decision; 0: none -- for loops whose steps keep ending at their starting count).  Same on every rank."""
        _lib.check(self._lib.pse_team_set_lanczos_extra(self._t, int(extra)))

    def local_status(self):
        """Synchronises; raises if a member's step failed on the device (capacity exceeded, a particle moved too far)."""
        flags = (ctypes.c_int * len(self.engines))()
        _lib.check(self._lib.pse_team_local_status(self._t, flags))
        return list(flags)

    def step(self, pos, vel, accel, image, force, kT, dt, timestep, shear_rate=0.0, group=None, lanczos_m=2):
        n = pos[0].shape[0] if group is None else group.shape[0]
        import torch
        self._chk(group, ("pos", pos), ("vel", vel), ("force", force))
        for a_, im_ in zip(accel, image):
            _chk_arr(a_, "accel", 3, torch.float64, n); _chk_arr(im_, "image", 3, torch.int32, n)
        m = ctypes.c_int(int(lanczos_m))
        _lib.check(self._lib.pse_team_step(self._t, self._ptrs(pos), self._ptrs(vel), self._ptrs(accel),
                                           self._ptrs(image), self._ptrs(force), _ptr(group), n, float(kT), float(dt),
                                           int(timestep), float(shear_rate), ctypes.byref(m)))
        return m.value
