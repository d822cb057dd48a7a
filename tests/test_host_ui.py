"""CPU tests of the C++ host classes (module _PSEv1) and the mirrored Python UI: shear functions against the oracle's
restatement of PSEv1/SpecificShearFunction.h, the wrapped-strain variant, validation behaviour, the seed hash."""
import math

import pytest


@pytest.fixture(scope="module")
def mod():
    from pse_amd import build
    build.build_all()
    from pse_amd import _PSEv1
    return _PSEv1


def test_cpp_shear_functions_match_oracle(mod, oracle):
    dt = 1e-3
    pairs = [
        (mod.SinShearFunction(1.3, 0.7, 10, dt), oracle.SinShear(1.3, 0.7, 10, dt)),
        (mod.SteadyShearFunction(0.4, 5, dt), oracle.SteadyShear(0.4, 5, dt)),
        (mod.ChirpShearFunction(0.1, 1.0, 20.0, 3.0, 0, dt), oracle.ChirpShear(0.1, 1.0, 20.0, 3.0, 0, dt)),
        (mod.TukeyWindowFunction(2.0, 0.4, 100, dt), oracle.TukeyWindow(2.0, 0.4, 100, dt)),
    ]
    pairs.append((mod.WindowedFunction(pairs[0][0], pairs[3][0]), oracle.Windowed(pairs[0][1], pairs[3][1])))
    for cpp, ref in pairs:
        for t in (0, 10, 11, 100, 101, 499, 500, 1234, 2099, 2100, 5000):
            assert abs(cpp.getShearRate(t) - ref.shear_rate(t)) < 1e-12 * max(1, abs(ref.shear_rate(t))), (type(ref), t)
            assert abs(cpp.getStrain(t) - ref.strain(t)) < 1e-12 * max(1, abs(ref.strain(t))), (type(ref), t)
        assert cpp.getOffset() == ref.offset


def test_variant_wraps_strain(mod, oracle):
    f = mod.SteadyShearFunction(1.0, 5, 1e-2)
    v = mod.VariantShearFunction(f, 1000, -0.5, 0.5)
    ref = oracle.SteadyShear(1.0, 5, 1e-2)
    for t in (0, 4, 5, 30, 55, 56, 155, 1004, 1005, 5000):
        assert abs(v.getValue(t) - oracle.variant_value(ref, t, 1000, -0.5, 0.5)) < 1e-12
        assert -0.5 <= v.getValue(t) < 0.5


def test_python_subclass_can_override(mod):
    class Mine(mod.ShearFunction):
        def __init__(self):
            mod.ShearFunction.__init__(self)

        def getShearRate(self, t):
            return 2.5

        def getStrain(self, t):
            return 0.5 * t

    base, win = Mine(), mod.SteadyShearFunction(0.0, 0, 1.0)
    w = mod.WindowedFunction(base, win)
    assert w.getStrain(4) == 0.0 and base.getShearRate(1) == 2.5
    assert mod.ShearFunction().getShearRate(3) == 0.0 and mod.ShearFunction().getOffset() == 0


def test_ui_validation_matches_reference_messages(mod, capsys):
    from pse_amd import shear_function, variant
    with pytest.raises(RuntimeError, match="Error creating shear function"):
        shear_function.sine(dt=1e-3, shear_rate=0.0, shear_freq=1.0)
    assert "Shear rate must be positive" in capsys.readouterr().err
    with pytest.raises(RuntimeError, match="Error creating shear function"):
        shear_function.sine(dt=1e-3, shear_rate=1.0, shear_freq=-1.0)
    with pytest.raises(RuntimeError, match="Tukey"):
        shear_function.tukey_window(dt=1e-3, periodT=1.0, tukey_param=1.5)
    with pytest.raises(RuntimeError, match="Error creating shear function"):
        shear_function.steady(dt=1e-3, shear_rate=1.0, zero=-1)
    with pytest.raises(RuntimeError, match="Error creating shear function"):
        shear_function.steady(dt=1e-3, shear_rate=1.0, zero=10)     # zero in the future (current step is 0)
    s = shear_function.sine(dt=1e-3, shear_rate=1.0, shear_freq=1.0)
    assert s.get_offset() == 0 and abs(s.get_shear_rate(0) - 1.0) < 1e-15
    assert abs(s.get_strain(250) - 1 / (2 * math.pi)) < 1e-15
    w = shear_function.windowed(s, shear_function.tukey_window(dt=1e-3, periodT=1.0, tukey_param=0.5))
    assert abs(w.get_strain(125) - 0.5 * s.get_strain(125)) < 1e-15
    with pytest.raises(RuntimeError, match="Error creating variant"):
        variant.shear_variant(s, 0)
    v = variant.shear_variant(shear_function.steady(dt=1e-2, shear_rate=1.0), 1000)
    assert abs(v.get_value(70) - (-0.3)) < 1e-12


def test_seed_hash_matches_reference_arithmetic(mod, oracle):
    for seed in (0, 1, 2, 12345, 0xFFFFFFFF):
        s = mod.Stokes(10, 20.0, 20.0, 20.0, 0.0, mod.VariantConst(1.0), seed, 0.5, 1e-3, 1e-3)
        assert s.hashedSeed() == oracle.hash_seed(seed)


def test_hoomd_shim_covers_the_reference_example():
    """Every `hoomd....` name the reference's examples/run.py touches resolves on the stand-in package (the script itself
    needs a GPU; it is not copied into this repo).  Skipped where the reference tree is not mounted."""
    import ast
    import importlib
    import os
    import sys
    ref = "/root/reference/examples/run.py"
    if not os.path.exists(ref):
        pytest.skip("reference tree not present")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "compat"))
    try:
        tree = ast.parse(open(ref).read())
        chains, imports = set(), set()
        for node in ast.walk(tree):
            if isinstance(node, ast.Import):
                imports.update(a.name for a in node.names if a.name.startswith("hoomd"))
            elif isinstance(node, ast.ImportFrom) and node.module and node.module.startswith("hoomd"):
                imports.update(node.module + "." + a.name for a in node.names)
            elif isinstance(node, ast.Attribute):
                parts, cur = [], node
                while isinstance(cur, ast.Attribute):
                    parts.append(cur.attr); cur = cur.value
                if isinstance(cur, ast.Name) and cur.id == "hoomd":
                    chains.add(tuple(reversed(parts)))
        assert chains and imports
        for name in sorted(imports):
            try:
                importlib.import_module(name)
            except ImportError:
                mod, _, attr = name.rpartition(".")
                assert hasattr(importlib.import_module(mod), attr), name
        hoomd = importlib.import_module("hoomd")
        importlib.import_module("hoomd.PSEv1")
        for chain in sorted(chains):
            obj = hoomd
            for a in chain:
                assert hasattr(obj, a), "hoomd." + ".".join(chain)
                obj = getattr(obj, a)
        # keyword arguments the script passes to the integrator and the shear function
        import inspect
        sig = inspect.signature(hoomd.PSEv1.integrate.PSEv1.__init__).parameters
        for kw in ("group", "seed", "T", "xi", "error", "function_form"):
            assert kw in sig, kw
        sig = inspect.signature(hoomd.PSEv1.shear_function.sine.__init__).parameters
        for kw in ("dt", "shear_rate", "shear_freq"):
            assert kw in sig, kw
        # dry run: execute the script top to bottom with the two GPU-touching entry points replaced by recorders
        import runpy
        import tempfile
        from pse_amd import context as pctx, integrate as pint, system as psys
        calls = {}

        class FakeSystem:
            def __init__(self, a, n):
                self.n, self.dt, self.timestep, self.integrators = n ** 3, None, 0, []
                self.box = (n * a, n * a, n * a, 0.0)
                calls["lattice"] = (a, n)
                pctx.current = self

            def all(self):
                return ("group", self)

            def run(self, nsteps):
                calls["run"] = nsteps

        def fake_pse(**kw):
            calls["pse"] = kw
            return object()

        real_create, real_pse = psys.System.create_lattice_sc, pint.PSEv1
        psys.System.create_lattice_sc = classmethod(lambda cls, a, n, dt=1e-3: FakeSystem(a, n))
        pint.PSEv1 = fake_pse
        cwd = os.getcwd()
        try:
            with tempfile.TemporaryDirectory() as tmp:
                os.chdir(tmp)
                os.environ.pop("PSE_EXAMPLE_STEPS", None)
                runpy.run_path(ref, run_name="__main__")
        finally:
            os.chdir(cwd)
            psys.System.create_lattice_sc, pint.PSEv1 = real_create, real_pse
            pctx.current = None
        assert calls["lattice"] == (6.4, 10) and calls["run"] == 1000
        assert pctx.current is None and calls["pse"]["xi"] == 0.5 and calls["pse"]["error"] == 1e-3 and calls["pse"]["seed"] == 1
        assert type(calls["pse"]["function_form"]).__name__ == "sine"
    finally:
        sys.path.pop(0)


def test_host_parameter_rule_refuses_what_the_device_refuses():
    """pse_host_select_params applies the same validity check as pse_create (gaussian_fits, csrc/pse_host_api.cpp): an override
    whose coarsest grid spacing takes the spreading Gaussian out of the double range over its support is refused on the host
    too -- and under the sanitizer build, whose pse_create stand-in calls the same function."""
    import pytest
    from pse_amd import PSEError, host_select_params
    ok = host_select_params((24.0, 24.0, 24.0, 0.0), xi=0.5, error=1e-3, grid=(64, 64, 64), P=4)
    assert ok["P"] == 4
    with pytest.raises(PSEError, match="too unequal"):
        host_select_params((24.0, 24.0, 24.0, 0.0), xi=0.5, error=1e-3, grid=(256, 16, 16), P=4)
