#!/usr/bin/env python3
"""Pins the oracle to the reference's OWN arithmetic.

The reference cannot be compiled or imported here (CUDA + HOOMD), and ships no golden vectors.  What it does hold is
plain arithmetic written out as expression text.  This script -- run in the build container only, where
/root/reference is mounted -- READS that expression text at run time (it is not embedded here), evaluates it on
a grid of inputs and writes the numbers to tests/golden/reference_arithmetic.json.  The fixture is data: inputs and the
values the reference's expressions take.  Sources evaluated:

  PSEv1/Stokes.cc:319            m_self                              (self mobility)
  PSEv1/Stokes.cc:348-406        Imrr, rr  for r > 2a, r == 2a, r < 2a (real-space RPY-Ewald functions f, g)
  PSEv1/Stokes.cc:135-236        the parameter-rule expressions (rcut, kmax, N_raw, lambda, gaussm loop test, P, eta)
  PSEv1/Stokes.cc:102            seed hash
  PSEv1/Helper.cu:326            wave-space scaling factor gridk.w
  PSEv1/SpecificShearFunction.h  getShearRate / getStrain bodies of Sin, Steady, Chirp, TukeyWindow (+ Windowed product rule)
  PSEv1/VariantShearFunction.h:47  wrapValue

Each value is stored twice where the reference's text carries a deliberate-difference constant:
"as_written" (its truncated pi = 3.1415926536 / 3.1416926536, logf, pi12 = 1.77245385091) and "exact" (the same text with
those constants replaced by exact pi / log / sqrt(pi)); the closed forms are evaluated in 50-digit arithmetic ("exact": the
mathematical value of the reference's expression) and in IEEE double as the reference itself does ("fp64").

  python3 tests/golden/make_reference_fixture.py       # rewrites tests/golden/reference_arithmetic.json
"""
import json
import math
import os
import re
import sys

import mpmath as mp
import numpy as np

REF = os.environ.get("PSE_REFERENCE", "/root/reference")
# The reference tree is untrusted content.  An empty __builtins__ is NOT a sandbox (any callable handed to the text leads back to
# the real builtins through its __globals__), so whatever text is taken from the tree is first parsed and held to a whitelist:
# arithmetic, comparisons, calls of plain NAMES, numeric constants -- no attribute access, no subscripts, no lambdas, no
# comprehensions, no strings.  Only a tree that passes is compiled.  The class bodies built from SpecificShearFunction.h pass a
# second whitelist that also admits `self.m_*` / `self.get*` and the handful of statements a getter is made of.  (The
# kernel-level fixture, make_kernel_fixture.py, does not evaluate text at all: it interprets a parsed tree.)
import ast

NO_BUILTINS = {"__builtins__": {}}
CLASS_BUILTINS = {"__builtins__": {"__build_class__": __build_class__}, "__name__": "reference_fixture"}
_EXPR_NODES = (ast.Expression, ast.BinOp, ast.UnaryOp, ast.BoolOp, ast.Compare, ast.IfExp, ast.Call, ast.Name, ast.Load, ast.Constant,
               ast.Add, ast.Sub, ast.Mult, ast.Div, ast.Mod, ast.Pow, ast.FloorDiv, ast.USub, ast.UAdd, ast.Not, ast.And, ast.Or,
               ast.Lt, ast.LtE, ast.Gt, ast.GtE, ast.Eq, ast.NotEq, ast.BitXor, ast.BitAnd, ast.BitOr, ast.LShift, ast.RShift)
_STMT_NODES = (ast.Module, ast.ClassDef, ast.FunctionDef, ast.arguments, ast.arg, ast.Assign, ast.AugAssign, ast.Return, ast.If,
               ast.Expr, ast.Store, ast.Pass)


def _check_tree(tree, what, allow_self):
    for node in ast.walk(tree):
        if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name) and node.attr in ("x", "y", "z") and node.value.id != "self":
            continue   # a component of a Scalar3 (L.x): the struct handed in holds three numbers and nothing else
        if isinstance(node, ast.Attribute) and allow_self:
            chain = node   # self.m_x, self.getY, self.m_x.getY: rooted at `self`, every link a member or a getter
            while isinstance(chain, ast.Attribute) and re.fullmatch(r"(m_\w+|get\w+)", chain.attr):
                chain = chain.value
            if not (isinstance(chain, ast.Name) and chain.id == "self"):
                raise ValueError(f"{what}: attribute access other than self.m_* / self.get* in reference text")
            continue
        if isinstance(node, ast.Constant):
            if not isinstance(node.value, (int, float)) or isinstance(node.value, bool):
                raise ValueError(f"{what}: non-numeric constant {node.value!r} in reference text")
            continue
        if isinstance(node, ast.Call):
            ok = isinstance(node.func, ast.Name) or (allow_self and isinstance(node.func, ast.Attribute))
            if not ok or node.keywords:
                raise ValueError(f"{what}: call of something that is not a plain name in reference text")
            continue
        if isinstance(node, ast.Name) and node.id.startswith("__"):
            raise ValueError(f"{what}: dunder name {node.id} in reference text")
        if not isinstance(node, _EXPR_NODES + (_STMT_NODES if allow_self else ())):
            raise ValueError(f"{what}: {type(node).__name__} is not allowed in reference text")


def safe_compile(text, what):
    """Expression text from the reference tree -> code object, after the whitelist."""
    tree = ast.parse(text.strip(), what, "eval")
    _check_tree(tree, what, allow_self=False)
    return compile(tree, what, "eval")


def safe_eval(code, env):
    if isinstance(code, str):
        code = safe_compile(code, "reference expression")
    return eval(code, dict(NO_BUILTINS, **env))


def safe_exec_class(text, ns, what):
    tree = ast.parse(text, what, "exec")
    _check_tree(tree, what, allow_self=True)
    exec(compile(tree, what, "exec"), ns)


OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_arithmetic.json")


def lines(path, lo, hi):
    with open(os.path.join(REF, path)) as f:
        return "".join(f.readlines()[lo - 1:hi])


def c_expr(text):
    """C arithmetic expression text -> Python expression text (same tokens: pow/exp/erfc/sin/cos are supplied as names)."""
    text = re.sub(r"\s+", " ", text.strip())
    text = re.sub(r"(?<![\w.])(\d+)\.(?![\d\w])", r"\1.0", text)           # "2." -> "2.0"
    text = re.sub(r"\b(Scalar|double|float)\s*\(", "(", text)               # casts
    text = re.sub(r"\blambda\b", "lambda_", text)                          # a Python keyword
    return text


# ---------------------------------------------------------------------------------------------- real-space closed forms
def realspace():
    src = lines("PSEv1/Stokes.cc", 344, 408)
    found = re.findall(r"\b(Imrr|rr)\s*=\s*(.*?);", src, flags=re.S)
    found = [(n, e) for n, e in found if e.strip() not in ("0", "0, rr = 0")]
    assert [n for n, _ in found] == ["Imrr", "rr"] * 3, [n for n, _ in found]
    exprs = {"gt": (found[0][1], found[1][1]), "eq": (found[2][1], found[3][1]), "lt": (found[4][1], found[5][1])}
    code = {k: tuple(safe_compile(c_expr(e), f"Stokes.cc:{k}") for e in v) for k, v in exprs.items()}
    self_src = re.search(r"m_self\s*=\s*(.*?);", lines("PSEv1/Stokes.cc", 315, 320), flags=re.S).group(1)
    self_code = safe_compile(c_expr(self_src), "Stokes.cc:319")

    def branch(r):
        return "gt" if r > 2.0 else ("eq" if r == 2.0 else "lt")

    env64 = {"pow": math.pow, "exp": math.exp, "erfc": math.erfc, "Pi": 3.141592653589793, "a": 1.0}
    mp.mp.dps = 50
    envmp = {"pow": mp.power, "exp": mp.exp, "erfc": mp.erfc, "Pi": mp.pi, "a": mp.mpf(1)}
    xis = [0.2733, 0.3, 0.441, 0.5, 0.546, 0.788, 0.8]
    rs = [0.001, 0.01, 0.05, 0.125, 0.2, 0.249, 0.251, 0.3, 0.5, 0.75, 1.0, 1.5, 1.9, 1.999, 2.0, 2.001, 2.1, 2.5, 3.0, 3.7,
          4.0, 4.81, 5.0, 5.26, 5.97, 6.0, 7.5, 9.0, 9.6]
    rows = []
    for xi in xis:
        for r in rs:
            b = branch(r)
            e64 = dict(env64, xi=xi, r=r)
            emp = dict(envmp, xi=mp.mpf(xi), r=mp.mpf(r))
            f64, g64 = safe_eval(code[b][0], e64), safe_eval(code[b][1], e64)
            fmp, gmp = safe_eval(code[b][0], emp), safe_eval(code[b][1], emp)
            rows.append({"xi": xi, "r": r, "branch": b, "Imrr_fp64": f64, "rr_fp64": g64,
                         "Imrr_exact": float(fmp), "rr_exact": float(gmp)})
    selfs = []
    for xi in xis:
        as_written = safe_eval(self_code, {"exp": math.exp, "erfc": math.erfc, "pi12": 1.77245385091, "axi": xi, "axi2": xi * xi, "aa": 1.0})
        exact = safe_eval(self_code, {"exp": mp.exp, "erfc": mp.erfc, "pi12": mp.sqrt(mp.pi), "axi": mp.mpf(xi),
                                 "axi2": mp.mpf(xi) ** 2, "aa": mp.mpf(1)})
        selfs.append({"xi": xi, "as_written": as_written, "exact": float(exact)})
    return {"source": "PSEv1/Stokes.cc:319,348-406", "rows": rows, "self": selfs}


# ---------------------------------------------------------------------------------------------- parameter rule
def parameter_rule():
    src = lines("PSEv1/Stokes.cc", 129, 237)

    def rhs(name):
        m = re.search(r"(?:\b(?:Scalar|int|double)\s+)?" + re.escape(name) + r"\s*=\s*(.*?);", src, flags=re.S)
        assert m, name
        return safe_compile(c_expr(m.group(1)), "Stokes.cc:" + name)
    e_cut, e_kmax, e_nx = rhs("m_ewald_cut"), rhs("kmax"), rhs("m_Nx")
    e_lambda, e_P, e_w, e_eta = rhs("lambda"), rhs("m_gaussP"), rhs("w"), rhs("m_eta")
    m = re.search(r"while\s*\((.*?)\)\s*\{\s*m_gaussm\s*=\s*(.*?);", src, flags=re.S)
    e_while, e_step = safe_compile(c_expr(m.group(1)), "while"), safe_compile(c_expr(m.group(2)), "step")
    mlist = sorted({2 ** a * 3 ** b * 5 ** c for a in range(13) for b in range(8) for c in range(6)
                    if 8 <= 2 ** a * 3 ** b * 5 ** c <= 4096})     # the loop bounds of Stokes.cc:153-175

    class V:   # struct with .x .y .z
        def __init__(s, x, y, z): s.x, s.y, s.z = x, y, z
    fns = {"sqrtf": math.sqrt, "logf": math.log, "erfcf": math.erfc, "int": int}   # the fp64 restatement
    rows = []
    for (L, xi, err, ms) in [(64.0, 0.5, 1e-3, 0.5), (64.0, 0.5, 1e-3, 0.0), (43.756, 0.5, 1e-3, 0.5), (347.29, 0.5, 1e-3, 0.5),
                             (280.04, 0.546, 1e-3, 0.5), (30.0, 0.8, 1e-6, 0.3), (25.0, 0.7, 1e-9, 0.5), (100.0, 0.3, 1e-4, 0.1)]:
        env = dict(fns, m_error=err, m_xi=xi, m_max_strain=ms, L=V(L, L, L))
        env["m_ewald_cut"] = safe_eval(e_cut, env)
        env["kmax"] = safe_eval(e_kmax, env)
        n_raw = safe_eval(e_nx, env)
        N = next(q for q in mlist if n_raw <= q)
        env["gamma"] = ms; env["gamma2"] = ms * ms
        env["lambda_"] = lam = safe_eval(e_lambda, env)
        env["m_gaussm"] = 1.0
        # integer counter: the reference accumulates +0.01 in Scalar; the fp64 restatement counts steps
        steps = 0
        while safe_eval(e_while, env):
            steps += 1
            env["m_gaussm"] = 1.0 + 0.01 * steps
            assert abs(safe_eval(e_step, dict(env, m_gaussm=1.0 + 0.01 * (steps - 1))) - env["m_gaussm"]) < 1e-12
        P = min(safe_eval(e_P, env), N)
        env["m_gaussP"] = P; env["m_gridh"] = V(L / N, L / N, L / N); env["m_Nx"] = N
        env["w"] = safe_eval(e_w, env); env["xisq"] = xi * xi
        rows.append({"L": L, "xi": xi, "error": err, "max_strain": ms, "rcut": env["m_ewald_cut"], "kmax": env["kmax"],
                     "N_raw": n_raw, "N": N, "lambda": lam, "gaussm": env["m_gaussm"], "P": P, "eta": safe_eval(e_eta, env)})
    return {"source": "PSEv1/Stokes.cc:135-236 (expressions as text, evaluated in fp64; pi as written 3.1415926536)", "rows": rows}


def seed_hash():
    src = lines("PSEv1/Stokes.cc", 102, 102)
    stmts = [s.strip() for s in src.split(";") if "m_seed" in s]
    out = []
    for seed in (0, 1, 2, 7, 12345, 0xFFFFFFFF):
        v = seed
        for s in stmts:
            m = re.match(r"m_seed\s*(\^=|\*=|=)\s*(.*)", s)
            val = safe_eval(m.group(2).replace("m_seed", str(v)), {}) & 0xFFFFFFFF
            v = {"=": val, "^=": v ^ val, "*=": (v * val) & 0xFFFFFFFF}[m.group(1)]
        out.append({"seed": seed, "hashed": v})
    return {"source": "PSEv1/Stokes.cc:102", "rows": out}


def wave_scale():
    src = lines("PSEv1/Helper.cu", 326, 326)
    expr = re.search(r"gridk_value\.w\s*=\s*(.*);", src).group(1).replace("Scalar(", "float(")
    rows = []
    for (k2, xi, eta, ng) in [(0.01, 0.5, 0.5054, 64 ** 3), (0.37, 0.441, 0.72, 256 ** 3), (2.5, 0.8, 0.6, 45 ** 3), (9.0, 0.3, 0.3, 100 ** 3)]:
        n = round(ng ** (1 / 3))
        env = {"k2": k2, "xisq": xi * xi, "eta": eta, "Nx": n, "Ny": n, "Nz": n, "float": float}
        as_written = safe_eval(c_expr(expr), dict(env, expf=math.exp))
        exact = safe_eval(c_expr(expr.replace("3.1415926536", "pi")), dict(env, expf=math.exp, pi=math.pi))
        rows.append({"k2": k2, "xi": xi, "eta": eta, "Ng": n ** 3, "as_written": as_written, "exact_pi": exact})
    return {"source": "PSEv1/Helper.cu:326", "rows": rows}


# ---------------------------------------------------------------------------------------------- shear functions
def method_to_python(body, members):
    """Body of a C++ getter (declarations, if / else if / else, return) -> Python source."""
    body = re.sub(r"//[^\n]*", "", body)
    out, depth = [], 0
    tokens = re.findall(r"else\s+if\s*\(.*?\)\s*\{|if\s*\(.*?\)\s*\{|else\s*\{|\}|[^;{}]+;", body, flags=re.S)
    for t in tokens:
        t = t.strip()
        if t == "}":
            depth -= 1
        elif t.startswith("else if"):
            out.append("    " * depth + "elif " + c_expr(re.search(r"\((.*)\)\s*\{", t, flags=re.S).group(1)) + ":"); depth += 1
        elif t.startswith("if"):
            out.append("    " * depth + "if " + c_expr(re.search(r"\((.*)\)\s*\{", t, flags=re.S).group(1)) + ":"); depth += 1
        elif t.startswith("else"):
            out.append("    " * depth + "else:"); depth += 1
        elif t.startswith("return"):
            out.append("    " * depth + "return " + c_expr(t[6:-1]))
        else:
            out.append("    " * depth + c_expr(re.sub(r"^\s*(double|int|unsigned int)\s+", "", t[:-1])))
    src = re.sub(r"\s*->\s*", ".", "\n".join(out).replace("||", " or ").replace("&&", " and "))
    for m in members:
        src = re.sub(r"\b" + m + r"\b", "self." + m, src)
    return src


def shear_functions():
    src = open(os.path.join(REF, "PSEv1/SpecificShearFunction.h")).read()

    def klass(name, exact):
        text = src[src.index("class " + name):]
        text = text[:text.index("\n};")]
        if exact:   # the deliberate differences of SURVEY 2.4: exact pi, double-precision log
            text = text.replace("3.1415926536", "3.14159265358979323846").replace("logf", "log")
        members = sorted(set(re.findall(r"\bm_\w+", text)))
        ctor_args = re.search(name + r"\((.*?)\)\s*:", text, flags=re.S).group(1)
        args = [a.strip().split()[-1] for a in ctor_args.split(",")]
        c0 = text.index(name + "(")
        inits = re.findall(r"(m_\w+)\((\w+)\)", text[c0:text.index("{", c0)])
        ctor_body = re.search(r"\)\s*\{(.*?)\}", text[text.index("ShearFunction()"):], flags=re.S).group(1)
        py = [f"class {name}:", "    def __init__(self, " + ", ".join(args) + "):"]
        py += [f"        self.{m} = {a}" for m, a in inits]
        pi = re.search(r"m_pi\s*=\s*([\d.]+)", text)
        if pi:
            py.append(f"        self.m_pi = {pi.group(1)}")
        for st in [s.strip() for s in ctor_body.split(";") if s.strip()]:
            py.append("        " + re.sub(r"\b(m_\w+)\b", r"self.\1", c_expr(st)))
        for meth in re.finditer(r"(?:double|unsigned int)\s+(get\w+)\(([^)]*)\)\s*\{", text):
            start, depth, i = meth.end(), 1, meth.end()
            while depth:
                depth += {"{": 1, "}": -1}.get(text[i], 0); i += 1
            arg = meth.group(2).split()[-1] if meth.group(2).strip() else ""
            body = method_to_python(text[start:i - 1], members)
            body = re.sub(r"(?<![.\w])(get\w+)\(", r"self.\1(", body)
            py.append(f"    def {meth.group(1)}(self{', ' + arg if arg else ''}):")
            py += ["        " + ln for ln in body.split("\n")]
        return "\n".join(py)

    def build(exact):
        f32log = (lambda x: float(np.log(np.float32(x)))) if not exact else math.log
        ns = dict(CLASS_BUILTINS, cos=math.cos, sin=math.sin, exp=math.exp, logf=f32log, log=math.log)
        for k in ("SinShearFunction", "SteadyShearFunction", "ChirpShearFunction", "TukeyWindowFunction", "WindowedFunction"):
            safe_exec_class(klass(k, exact), ns, "SpecificShearFunction.h:" + k)
        return ns
    dt = 1e-3
    cases = {
        "sine": ("SinShearFunction", (1.0, 1.0, 0, dt)),
        "sine_offset": ("SinShearFunction", (0.7, 2.5, 100, dt)),
        "steady": ("SteadyShearFunction", (0.35, 10, dt)),
        "chirp": ("ChirpShearFunction", (0.2, 1.0, 40.0, 2.0, 0, dt)),
        "tukey": ("TukeyWindowFunction", (2.0, 0.4, 0, dt)),
    }
    steps = [0, 1, 10, 100, 101, 137, 250, 399, 400, 401, 500, 999, 1000, 1500, 1600, 1601, 1999, 2000, 2001, 2500]
    out = {}
    for variant, exact in (("as_written", False), ("exact_constants", True)):
        ns = build(exact)
        objs = {k: ns[c](*a) for k, (c, a) in cases.items()}
        objs["windowed_chirp_tukey"] = ns["WindowedFunction"](objs["chirp"], objs["tukey"])
        for k, o in objs.items():
            ts = [t for t in steps if t >= o.getOffset()]   # the reference's unsigned subtraction wraps below the offset
            out.setdefault(k, {"args": list(cases[k][1]) if k in cases else "windowed(chirp, tukey)", "timesteps": ts})
            out[k][variant] = {"shear_rate": [o.getShearRate(t) for t in ts], "strain": [o.getStrain(t) for t in ts]}
    wrap_src = re.search(r"double wrapValue\(double functionValue\)\s*\{\s*return (.*?);", open(os.path.join(REF, "PSEv1/VariantShearFunction.h")).read(), flags=re.S).group(1)
    wrap = []
    for v in (-2.3, -0.5, -0.49999, 0.0, 0.2, 0.4999, 0.5, 0.77, 1.0, 3.21):
        wrap.append({"value": v, "min": -0.5, "range": 1.0,
                     "wrapped": safe_eval(c_expr(wrap_src), {"functionValue": v, "m_min_value": -0.5, "m_value_range": 1.0, "floor": math.floor})})
    return {"source": "PSEv1/SpecificShearFunction.h:16-223, PSEv1/VariantShearFunction.h:46-48", "dt": dt, "functions": out, "wrapValue": wrap}


def main():
    if not os.path.isdir(os.path.join(REF, "PSEv1")):
        sys.exit(f"{REF}/PSEv1 not found: this script only runs where the reference tree is mounted")
    fixture = {"generator": "tests/golden/make_reference_fixture.py", "realspace": realspace(), "parameter_rule": parameter_rule(),
               "seed_hash": seed_hash(), "wave_scale": wave_scale(), "shear": shear_functions()}
    with open(OUT, "w") as f:
        json.dump(fixture, f, indent=0)
    print("written", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
