"""What of the REFERENCE itself can be compiled in this image is compiled -- its shear-function classes
(PSEv1/SpecificShearFunction.h, PSEv1/ShearFunction.h: they need pybind11 only) -- by `make -C oracle ref` from the sources where
they lie, into oracle/_ref/libpse_ref_shear.so (git-ignored; built in the build container, travels to the GPU box as a built file).
Here the COMPILED reference is the judge of three things: (1) the fixture that round 2 produced by evaluating the text of those
getters (tests/golden/reference_arithmetic.json, "as_written") -- so the text-evaluation route is itself checked against a real
build; (2) the oracle's restatement; (3) the product's C++ host classes.  ("what's missing" 3.)"""
import ctypes
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "_ref", "libpse_ref_shear.so")
FIX = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_arithmetic.json")))
KIND = {"sine": 0, "sine_offset": 0, "steady": 1, "chirp": 2, "tukey": 3}


@pytest.fixture(scope="module")
def ref():
    if not os.path.exists(LIB):
        if not os.path.isdir("/root/reference"):
            pytest.skip("oracle/_ref/libpse_ref_shear.so has not been built (needs /root/reference: `make -C oracle ref`)")
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], stdout=subprocess.DEVNULL)
    L = ctypes.CDLL(LIB)
    L.pse_ref_shear.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.c_uint, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
    L.pse_ref_shear_windowed.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.POINTER(ctypes.c_double),
                                         ctypes.c_uint, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]

    def args(a):
        return (ctypes.c_double * len(a))(*[float(x) for x in a])

    def value(name, t, what):
        f = FIX["shear"]["functions"]
        out = ctypes.c_double()
        if name == "windowed_chirp_tukey":
            rc = L.pse_ref_shear_windowed(2, args(f["chirp"]["args"]), 3, args(f["tukey"]["args"]), t, what, ctypes.byref(out))
        else:
            rc = L.pse_ref_shear(KIND[name], args(f[name]["args"]), t, what, ctypes.byref(out))
        assert rc == 0
        return out.value
    return value


def test_text_evaluated_fixture_equals_the_compiled_reference(ref):
    """The "as_written" rows were produced by executing the TEXT of the reference's getters in Python (float32 log emulated); the
    reference compiled by g++ gives the same numbers: to the last bits of a double, where libm's and numpy's float32 log agree."""
    f = FIX["shear"]["functions"]
    worst = 0.0
    for name, rows in f.items():
        for t, r, s in zip(rows["timesteps"], rows["as_written"]["shear_rate"], rows["as_written"]["strain"]):
            for got, want in ((ref(name, t, 0), r), (ref(name, t, 1), s)):
                err = abs(got - want) / max(1.0, abs(want))
                worst = max(worst, err)
                assert err < 1e-12, (name, t, got, want)
    assert worst < 1e-12


def test_oracle_and_product_shear_against_the_compiled_reference(ref, oracle):
    """The oracle's restatement and the product's host classes use exact pi and a double-precision log: they differ from
    the reference AS BUILT by what those two constants move (chirp phase: ~1e-7 relative; everything else 1e-10)."""
    from pse_amd import build
    build.build_all()
    from pse_amd import _PSEv1 as mod
    f = FIX["shear"]["functions"]

    def mk_oracle(name):
        a = f[name]["args"]
        return {"sine": oracle.SinShear, "sine_offset": oracle.SinShear, "steady": oracle.SteadyShear, "chirp": oracle.ChirpShear,
                "tukey": oracle.TukeyWindow}[name](*a)

    def mk_cpp(name):
        a = list(f[name]["args"])
        off = {"sine": 2, "sine_offset": 2, "steady": 1, "chirp": 4, "tukey": 2}[name]
        a[off] = int(a[off])
        return {"sine": mod.SinShearFunction, "sine_offset": mod.SinShearFunction, "steady": mod.SteadyShearFunction,
                "chirp": mod.ChirpShearFunction, "tukey": mod.TukeyWindowFunction}[name](*a)
    objs_o = {n: mk_oracle(n) for n in KIND}
    objs_c = {n: mk_cpp(n) for n in KIND}
    objs_o["windowed_chirp_tukey"] = oracle.Windowed(objs_o["chirp"], objs_o["tukey"])
    objs_c["windowed_chirp_tukey"] = mod.WindowedFunction(objs_c["chirp"], objs_c["tukey"])
    for name, rows in f.items():
        tol = 2e-6 if "chirp" in name else 1e-9
        for t in rows["timesteps"]:
            r_ref, s_ref = ref(name, t, 0), ref(name, t, 1)
            for label, rate, strain in (("oracle", objs_o[name].shear_rate(t), objs_o[name].strain(t)),
                                        ("product", objs_c[name].getShearRate(t), objs_c[name].getStrain(t))):
                assert abs(rate - r_ref) <= tol * max(1.0, abs(r_ref)), (label, name, t, rate, r_ref)
                assert abs(strain - s_ref) <= tol * max(1.0, abs(s_ref)), (label, name, t, strain, s_ref)
        assert int(ref(name, 0, 2)) == objs_c[name].getOffset()
