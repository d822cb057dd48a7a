"""The C-subset interpreter behind the kernel fixture (tests/golden/cmini.py) must have C's semantics where they differ from
Python's: the fixture is only as good as the interpreter.  And, where the reference tree is mounted (the build container), the
committed fixtures must be exactly what the generators produce from it."""
import gzip
import json
import math
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import cmini                      # noqa: E402
from cmini import Machine, Vec    # noqa: E402

REF = os.environ.get("PSE_REFERENCE", "/root/reference")


def run(body, **env):
    m = Machine({"sqrtf": math.sqrt, "make_scalar2": lambda *a: Vec("Scalar2", *a)})
    return m.run(m.parse(body), env), m


def test_integer_division_truncates_and_remainder_follows_the_dividend():
    out, _ = run("int a = -7 / 2; int b = 7 / -2; int c = -7 % 3; int d = 7 % -3; Scalar e = -7 / 2; Scalar f = -7 / 2.0; r[0] = a; r[1] = b; r[2] = c; r[3] = d; s[0] = e; s[1] = f;",
                 r=[0] * 4, s=[0.0] * 2)
    assert out["r"] == [-3, -3, -1, 1] and out["s"] == [-3.0, -3.5]


def test_typed_declarations_coerce():
    out, _ = run("int i = 2.9; int j = -2.9; Scalar x = 3; bool b = 7; unsigned int u = 5; r[0] = i; r[1] = j; r[2] = b; r[3] = u / 2; s[0] = x / 2;", r=[0] * 4, s=[0.0])
    assert out["r"] == [2, -2, 1, 2] and out["s"] == [1.5]


def test_ternary_logic_and_comparisons_are_ints():
    out, _ = run("int P = 5; Scalar f = 0.25; int a = 3 - (P % 2) * ( f < 0.5 ); int b = (a > 1 && !(a == 3)) ? 10 : 20; int c = (0 || a) + (1 && 0); r[0] = a; r[1] = b; r[2] = c;",
                 r=[0] * 3)
    assert out["r"] == [2, 10, 1]


def test_casts_structs_pointers_and_shared_arrays():
    body = """
    __shared__ Scalar3 shared[2];
    Scalar3 *first = shared;
    Scalar3 *second = &shared[1];
    if (tid == 0) { first[0].x = 1.5; second[0].y = (Scalar)n / 4; }
    Scalar2 v = make_scalar2(3.0, 4.0);
    Scalar norm = sqrtf( v.x*v.x + v.y*v.y );
    int3 t; t.x = 2.7; t.y = int( norm ) + (int)1.9;
    out[0] = shared[0].x + shared[1].y + norm; out[1] = t.x + t.y;
    """
    shared = [Vec("Scalar3"), Vec("Scalar3")]
    out, _ = run(body, tid=0, n=3, out=[0.0, 0.0], shared=shared)
    assert out["out"] == [1.5 + 0.75 + 5.0, 2 + 6]
    out2, _ = run(body, tid=1, n=3, out=[0.0, 0.0], shared=shared)          # a later thread of the block sees what thread 0 stored
    assert out2["out"][0] == 1.5 + 0.75 + 5.0


def test_for_loops_compound_assignment_and_return():
    out, m = run("int s = 0; for (int i = 0; i < 5; ++i) { s += i; if (i == 3) { acc[0] = s; } } acc[1] = s; Scalar q = 10; q /= 4; q *= 2; acc[2] = q; return; acc[0] = -1;", acc=[0, 0, 0.0])
    assert out["acc"] == [6, 10, 5.0]
    m2 = Machine({})
    m2.run(m2.parse("return 3 * (x + 1);"), {"x": 4})
    assert m2.returned == 15
    assert Machine({"pow": math.pow}).evaluate("pow(2.0, 3) / 4 + 7 / 2", {}) == 2.0 + 3


def test_host_code_constructs():
    """What the Lanczos driver needs beyond kernel bodies: while / break / continue, shifts, string and character literals (with
    comment markers inside strings), casts to pointer types, sizeof as an element count that remembers its type, &variable."""
    from cmini import SizeOf
    seen = []

    def alloc(ref, count):
        seen.append((int(count), count.tname))
        ref.put([0.0] * int(count))
    m = Machine({"printf": lambda *a: seen.append(a[0]), "cudaMalloc": alloc, "malloc": lambda c: [float("nan")] * int(c)})
    out = m.run(m.parse("""
    float *a; a = (float *)malloc( 6*sizeof(float) );
    Scalar *d; cudaMalloc( (void**)&d, n_el*sizeof(Scalar) );
    while ( n < 10 ) { n++; if (n == 3) { continue; } if (n > 6) { break; } s += n; a[n-1] = n / 2.0; d[0] += 1; }
    int offs = 512 >> 1; offs >>= 2;
    printf("text with // and /* inside */ %i \\n", n); /* a comment */ // another
    res[0] = 'I'; res[1] = offs; res[2] = a[5]; res[3] = d[0]; res[4] = 1 << 4;
    """), {"res": [None] * 5, "n": 0, "s": 0, "n_el": 3})
    assert (out["n"], out["s"]) == (7, 18)
    assert out["res"] == ["I", 64, 3.0, 5.0, 16]
    assert seen[0] == (3, "Scalar") and seen[1].startswith("text with // and /* inside */")
    assert isinstance(SizeOf(1, "float") * 4, SizeOf) and int(3 * SizeOf(1, "float")) == 3


def test_unknown_names_are_errors_not_python():
    m = Machine({})
    with pytest.raises(cmini.CError):
        m.run(m.parse("int a = __import__(1);"), {})
    with pytest.raises(cmini.CError):
        m.run(m.parse("int a = undefined_name + 1;"), {})


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "PSEv1")), reason="the reference tree is mounted in the build container only")
def test_committed_fixtures_are_what_the_generators_produce(tmp_path, monkeypatch):
    import importlib
    golden = os.path.join(HERE, "golden")
    mk = importlib.import_module("make_kernel_fixture")
    monkeypatch.setattr(mk, "OUT", str(tmp_path / "k.json.gz"))
    mk.main()
    with gzip.open(tmp_path / "k.json.gz") as f, gzip.open(os.path.join(golden, "reference_kernels.json.gz")) as g:
        assert json.load(f) == json.load(g)
    ml = importlib.import_module("make_lanczos_fixture")
    monkeypatch.setattr(ml, "OUT", str(tmp_path / "l.json"))
    ml.main()
    assert json.load(open(tmp_path / "l.json")) == json.load(open(os.path.join(golden, "reference_lanczos.json")))
    md = importlib.import_module("make_driver_fixture")
    monkeypatch.setattr(md, "OUT", str(tmp_path / "d.json.gz"))
    md.main()
    with gzip.open(tmp_path / "d.json.gz") as f, gzip.open(os.path.join(golden, "reference_driver.json.gz")) as g:
        assert json.load(f) == json.load(g)
    mr = importlib.import_module("make_reference_fixture")
    monkeypatch.setattr(mr, "OUT", str(tmp_path / "a.json"))
    mr.main()
    assert json.load(open(tmp_path / "a.json")) == json.load(open(os.path.join(golden, "reference_arithmetic.json")))
