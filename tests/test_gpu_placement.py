"""pse_create's grid-placement planner (place_grids, pse_capi.hip; PSE_PLACE_TRIALS): for far-field grids large enough to matter the
pair (real grid, spectra) is allocated several times and the pair on which the x pass and the inverse y + z passes run fastest is kept.  The choice
must not change any result (beyond the 1e-14 by which two calls of ONE engine differ: the order of the particles inside a far-field
bin is that of an atomic counter), must leave exactly one pair allocated, and must be off where it is said to be off."""
import math
import os

import pytest

pytestmark = pytest.mark.gpu


def _engine(trials, grid, n, box):
    import pse_amd
    old = os.environ.get("PSE_PLACE_TRIALS")
    if trials is None:
        os.environ.pop("PSE_PLACE_TRIALS", None)
    else:
        os.environ["PSE_PLACE_TRIALS"] = str(trials)
    try:      # (the developer switches are read in pse_create)
        xi = math.pi * grid / (2.0 * box[0] * math.sqrt(-math.log(1e-3)))
        return pse_amd.Engine(n, box, xi=xi, error=1e-3, seed=3, grid=(grid,) * 3)
    finally:
        if old is None:
            os.environ.pop("PSE_PLACE_TRIALS", None)
        else:
            os.environ["PSE_PLACE_TRIALS"] = old


def test_placement_planner_keeps_one_pair_and_changes_no_result():
    import torch
    from conftest import make_suspension, to4
    n, grid = 20_000, 180                                # spectra 3 x 180 x 180 x 91 x 16 B = 141 MB: above the planner's threshold (96 MB)
    pos, force, box = make_suspension(n, L=60.0)         # (h = 1/3: a grid the rule would pick for a box of this size)
    dpos, dF = to4(pos, 1.0), to4(force)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    off = _engine(0, grid, n, box)
    free_off = torch.cuda.mem_get_info()[0]
    on = _engine(None, grid, n, box)                     # the default: up to ten trials
    free_on = torch.cuda.mem_get_info()[0]
    p_off, p_on = off.grid_placement(), on.grid_placement()
    assert p_off == {"tried": 0, "ms_first": 0.0, "ms_kept": 0.0}
    assert 2 <= p_on["tried"] <= 10 and 0.0 < p_on["ms_kept"] <= p_on["ms_first"]
    # the candidates that were not kept are freed: both engines hold the same amount of device memory (2 MB allocation granules)
    assert abs((free0 - free_off) - (free_off - free_on)) < (64 << 20), (free0, free_off, free_on)
    assert off.info()["device_bytes"] == on.info()["device_bytes"]
    u_off, u_on = torch.zeros_like(dpos), torch.zeros_like(dpos)
    off.mobility(dpos, dF, vel=u_off)
    on.mobility(dpos, dF, vel=u_on)
    assert (u_off - u_on).abs().max().item() < 1e-12 * u_off.abs().max().item()
    u_near_off, u_near_on = torch.zeros_like(dpos), torch.zeros_like(dpos)
    off.mobility(dpos, dF, vel=u_near_off, parts=1)      # (the near field has no such freedom: equal to the bit)
    on.mobility(dpos, dF, vel=u_near_on, parts=1)
    assert torch.equal(u_near_off, u_near_on)
    v_off, v_on = torch.zeros_like(dpos), torch.zeros_like(dpos)
    _, m0 = off.brownian_velocity(dpos, dF, 1.0, 1e-3, 7, vel=v_off)
    _, m1 = on.brownian_velocity(dpos, dF, 1.0, 1e-3, 7, vel=v_on)
    assert m0 == m1 and (v_off - v_on).abs().max().item() < 1e-11 * v_off.abs().max().item()
    small = _engine(None, 64, 2000, (24.0, 24.0, 24.0, 0.0))       # a small grid: nothing to plan
    assert small.grid_placement()["tried"] == 0


def test_packed_vector_rows_round_trip_within_their_bound():
    """vq_pack / vq_unpack (pse_device.h): the 16-byte form in which the pair-list mat-vec gathers its neighbours' rows of the Lanczos
    vector -- three 40-bit mantissas under the exponent of the row's largest component: |error| <= 2^-39 of that component, for any
    scale of the row, components of either sign, zeros, and a largest component just below a power of two (the mantissa clamp)."""
    import ctypes
    import numpy as np
    from pse_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(5)
    rows = rng.standard_normal((20000, 3)) * 10.0 ** rng.uniform(-25, 25, (20000, 1))
    rows[:2000] *= 10.0 ** rng.uniform(-6, 0, (2000, 3))          # wide ratios inside a row
    rows[2000:2100, 1] = 0.0
    rows[2100] = 0.0
    rows[2101] = (np.nextafter(1.0, 0.0), -np.nextafter(1.0, 0.0), 0.5)
    rows[2102] = (np.nextafter(2.0 ** -30, 0.0), 2.0 ** -31, -2.0 ** -69)
    rows[2103] = (1.0, -1.0, 2.0 ** -38)
    out = np.zeros_like(rows)
    dp = ctypes.POINTER(ctypes.c_double)
    assert lib.pse_debug_vq_roundtrip(len(rows), rows.ctypes.data_as(dp), out.ctypes.data_as(dp)) == 0
    bound = 2.0 ** -39 * np.abs(rows).max(axis=1, keepdims=True)
    assert (np.abs(out - rows) <= bound).all()
    assert (out[2100] == 0.0).all() and (out[2000:2100, 1] == 0.0).all()
    assert out[2103, 2] == 2.0 ** -38                                 # one unit of the last place of the mantissa (1.0 = 0.5 x 2^1: scale 2^38)


def _vq_model(rows):
    """NumPy restatement of vq_pack followed by vq_unpack (pse_device.h): exponent e of the row's largest magnitude (m = f 2^e, 0.5 <= f < 1),
    clamped to [-100, 127]; mantissas rint(component x 2^(39 - e)) clamped to +-(2^39 - 1); value = mantissa x 2^(e - 39)."""
    import numpy as np
    m = np.abs(rows).max(axis=1)
    e = np.where(m > 0, np.frexp(m)[1], 0)
    e = np.clip(e, -100, 127)
    lim = 2.0 ** 39 - 1
    mant = np.clip(np.rint(np.ldexp(rows, (39 - e)[:, None])), -lim, lim)
    return np.ldexp(mant, (e - 39)[:, None])


def test_packed_vector_rows_are_the_stated_format_to_the_bit():
    import ctypes
    import numpy as np
    from pse_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(11)
    rows = rng.standard_normal((50000, 3)) * 10.0 ** rng.uniform(-12, 3, (50000, 1))
    rows[:5000] *= 10.0 ** rng.uniform(-14, 0, (5000, 3))
    rows[5000] = (np.nextafter(1.0, 0.0), 0.25, -0.125)            # rounds up to 2^39: clamped
    out = np.zeros_like(rows)
    dp = ctypes.POINTER(ctypes.c_double)
    assert lib.pse_debug_vq_roundtrip(len(rows), rows.ctypes.data_as(dp), out.ctypes.data_as(dp)) == 0
    assert np.array_equal(out, _vq_model(rows))


def test_the_mat_vec_alone_can_be_timed_after_a_brownian_call_only():
    """pse_debug_matvec_ms (what bench.py's roofline divides by): the pair-list mat-vec back to back between one pair of events; it needs
    the pair list of a Brownian call and leaves the engine usable (the next call's results are what they were)."""
    import torch
    from conftest import make_suspension, to4
    import pse_amd
    n = 40_000
    pos, force, box = make_suspension(n, phi=0.1)
    eng = pse_amd.Engine(n, box, xi=0.5, error=1e-3, seed=9)
    dpos, dF = to4(pos, 1.0), to4(force)
    with pytest.raises(RuntimeError, match="pair list"):
        eng.matvec_ms(3)                                             # nothing has run yet
    v0, m0 = eng.brownian_velocity(dpos, dF, 1.0, 1e-3, 5)
    v0 = v0.clone()
    ms = eng.matvec_ms(10)
    assert 0.0 < ms < 5.0
    v1, m1 = eng.brownian_velocity(dpos, dF, 1.0, 1e-3, 5)
    assert m1 == m0 and (v1 - v0).abs().max().item() < 1e-11 * v0.abs().max().item()
    eng.mobility(dpos, dF)                                           # a deterministic call: no pair list afterwards
    with pytest.raises(RuntimeError, match="pair list"):
        eng.matvec_ms(3)
