"""A small interpreter for the C subset the reference's CUDA kernels -- and the host function that drives its Lanczos iteration -- are
written in: test infrastructure of the fixture generators (tests/golden/make_kernel_fixture.py, make_lanczos_fixture.py), nothing
else imports it.

Why: the reference cannot be compiled here (CUDA + HOOMD headers absent) and ships no golden vectors, but the per-thread
arithmetic of its kernels is plain C.  The generator READS a kernel's text from /root/reference at generation time, this
module executes that text one thread at a time with C semantics (truncating integer division, C remainder, typed
declarations, ?:, && / ||, shifts, structs with .x .y .z .w, pointers into arrays, casts, sizeof, for / while / break / if / else,
string and character literals), and the numbers it produces are
what gets committed.  No reference text is stored; nothing is passed to eval()/exec(): the text is tokenised, parsed into a
tree of a fixed set of node kinds and walked, and every name it may call comes from a table the caller supplies.

What the caller supplies (the pieces of CUDA / HOOMD that are not in the reference tree): blockIdx/threadIdx/blockDim, the
HOOMD BoxDim helpers (restated in the generator, as SURVEY.md 8 a15 says they must be), math functions, the RNG stand-in.
"""
import math
import re

INT_TYPES = {"int", "unsigned", "unsigned int", "bool", "size_t"}
FLOAT_TYPES = {"Scalar", "float", "double"}
VEC_TYPES = {"Scalar2": ("xy", float), "Scalar3": ("xyz", float), "Scalar4": ("xyzw", float), "int3": ("xyz", int), "uint3": ("xyz", int),
             "CUFFTCOMPLEX": ("xy", float), "cufftComplex": ("xy", float), "float2": ("xy", float), "dim3": ("xyz", int)}


class CError(Exception):
    pass


def c_int(v):
    return int(v)          # Python int() truncates toward zero like a C conversion


class Vec:
    """A CUDA vector type: named fields of one scalar type."""
    __slots__ = ("tname", "f")

    def __init__(self, tname, *vals):
        names, conv = VEC_TYPES[tname]
        self.tname = tname
        vals = list(vals) + [0] * (len(names) - len(vals))
        self.f = {n: conv(v) for n, v in zip(names, vals)}

    def copy(self):
        v = Vec(self.tname)
        v.f = dict(self.f)
        return v

    def get(self, name):
        return self.f[name]

    def set(self, name, val):
        self.f[name] = VEC_TYPES[self.tname][1](val) if VEC_TYPES[self.tname][1] is float else c_int(val)

    def _bin(self, o, fn):
        r = Vec(self.tname)
        for n in self.f:
            r.f[n] = fn(self.f[n], o.f[n] if isinstance(o, Vec) else o)
        return r

    def __repr__(self):
        return f"{self.tname}({', '.join(repr(v) for v in self.f.values())})"


class Ptr:
    """Pointer into a Python list (array base + element offset)."""
    __slots__ = ("arr", "off")

    def __init__(self, arr, off=0):
        self.arr, self.off = arr, off


class Ref:
    """An lvalue: something that can be read and assigned."""
    def __init__(self, get, put):
        self.get, self.put = get, put


# ------------------------------------------------------------------------------------------------ tokens
TOKEN = re.compile(r"""
    (?P<str>"(?:[^"\\]|\\.)*")
  | (?P<chr>'(?:[^'\\]|\\.)')
  | (?P<num>(?:\d+\.\d*|\.\d+|\d+)(?:[eE][+-]?\d+)?[fFuUlL]*)
  | (?P<id>[A-Za-z_]\w*(?:::[A-Za-z_]\w*)*)
  | (?P<op>>>=|<<=|<<|>>|\+\+|--|->|\+=|-=|\*=|/=|==|!=|<=|>=|&&|\|\||[-+*/%<>=!&?:;,.(){}\[\]])
  | (?P<ws>\s+)
""", re.X)


def tokenize(text):
    text = re.sub(r'''("(?:[^"\\]|\\.)*")|/\*.*?\*/|//[^\n]*''', lambda m: m.group(1) or " ", text, flags=re.S)
    out, i = [], 0
    while i < len(text):
        m = TOKEN.match(text, i)
        if not m:
            raise CError(f"cannot tokenise at {text[i:i + 30]!r}")
        i = m.end()
        if m.lastgroup == "ws":
            continue
        out.append((m.lastgroup, m.group(m.lastgroup)))
    return out


# ------------------------------------------------------------------------------------------------ parser
class Parser:
    def __init__(self, tokens, type_names):
        self.t, self.i = tokens, 0
        self.types = set(type_names) | INT_TYPES | FLOAT_TYPES | set(VEC_TYPES)

    def peek(self, k=0):
        return self.t[self.i + k] if self.i + k < len(self.t) else ("eof", "")

    def next(self):
        tok = self.peek()
        self.i += 1
        return tok

    def accept(self, val):
        if self.peek()[1] == val and self.peek()[0] != "num":
            self.i += 1
            return True
        return False

    def expect(self, val):
        if not self.accept(val):
            raise CError(f"expected {val!r}, found {self.peek()!r} (token {self.i})")

    # -- types
    def at_type(self):
        k, v = self.peek()
        if k != "id":
            return False
        if v in ("const", "__shared__", "unsigned", "extern", "static"):
            return True
        return v in self.types

    def parse_type(self):
        words = []
        while self.peek()[1] in ("const", "__shared__", "extern", "static"):
            self.next()
        if self.accept("unsigned"):
            if self.peek()[1] == "int":
                self.next()
            words.append("unsigned int")
        else:
            words.append(self.next()[1])
        while self.peek()[1] == "const":
            self.next()
        return words[0]

    # -- statements
    def parse_block_body(self):
        out = []
        while self.peek()[0] != "eof" and self.peek()[1] != "}":
            out.append(self.statement())
        return ("block", out)

    def statement(self):
        k, v = self.peek()
        if v == "{":
            self.next()
            b = self.parse_block_body()
            self.expect("}")
            return b
        if v == ";":
            self.next()
            return ("block", [])
        if v == "if":
            self.next(); self.expect("(")
            c = self.expr(); self.expect(")")
            a = self.statement()
            b = None
            if self.accept("else"):
                b = self.statement()
            return ("if", c, a, b)
        if v == "for":
            self.next(); self.expect("(")
            init = self.declaration() if self.at_type() else (("expr", self.expr()) if self.peek()[1] != ";" else None)
            if init is None or init[0] == "expr":
                self.expect(";")
            cond = self.expr(); self.expect(";")
            step = self.expr(); self.expect(")")
            return ("for", init, cond, step, self.statement())
        if v == "return":
            self.next()
            e = None if self.peek()[1] == ";" else self.expr()
            self.expect(";")
            return ("return", e)
        if v == "while" and k == "id":
            self.next(); self.expect("(")
            c = self.expr(); self.expect(")")
            return ("while", c, self.statement())
        if v in ("break", "continue") and k == "id":
            self.next(); self.expect(";")
            return (v,)
        if self.at_type():
            # a type name followed by '(' is a function-style cast in an expression statement, not a declaration
            if not (self.peek(1)[1] == "(" and self.peek()[1] not in ("const", "unsigned", "__shared__")):
                return self.declaration()
        e = self.expr()
        self.expect(";")
        return ("expr", e)

    def declaration(self):
        tname = self.parse_type()
        decls = []
        while True:
            ptr = False
            while self.accept("*"):
                ptr = True
            name = self.next()[1]
            size = init = ctor = None
            if self.accept("["):
                size = self.expr(); self.expect("]")
            if self.accept("="):
                init = self.assign()
            elif self.peek()[1] == "(":                      # C++ constructor syntax: T name(args);
                self.next()
                ctor = self.args()
            decls.append((name, ptr, size, init, ctor))
            if not self.accept(","):
                break
        self.expect(";")
        return ("decl", tname, decls)

    # -- expressions (C precedence)
    def args(self):
        out = []
        if self.accept(")"):
            return out
        while True:
            out.append(self.assign())
            if self.accept(")"):
                return out
            self.expect(",")

    def expr(self):
        e = self.assign()
        while self.accept(","):
            e = ("comma", e, self.assign())
        return e

    def assign(self):
        left = self.ternary()
        k, v = self.peek()
        if k == "op" and v in ("=", "+=", "-=", "*=", "/=", ">>=", "<<="):
            self.next()
            return ("assign", v, left, self.assign())
        return left

    def ternary(self):
        c = self.binary(0)
        if self.accept("?"):
            a = self.assign(); self.expect(":")
            return ("cond", c, a, self.assign())
        return c

    LEVELS = [("||",), ("&&",), ("==", "!="), ("<", ">", "<=", ">="), ("<<", ">>"), ("+", "-"), ("*", "/", "%")]

    def binary(self, lvl):
        if lvl == len(self.LEVELS):
            return self.unary()
        e = self.binary(lvl + 1)
        while self.peek()[0] == "op" and self.peek()[1] in self.LEVELS[lvl]:
            op = self.next()[1]
            e = ("bin", op, e, self.binary(lvl + 1))
        return e

    def unary(self):
        k, v = self.peek()
        if k == "op" and v in ("-", "+", "!", "&", "*", "++", "--"):
            self.next()
            return ("un", v, self.unary())
        if v == "(" and self.peek(1)[0] == "id" and self.peek(1)[1] in self.types | {"unsigned", "void"}:
            j = 3 if (self.peek(1)[1] == "unsigned" and self.peek(2)[1] == "int") else 2
            stars = 0
            while self.peek(j + stars)[1] == "*":
                stars += 1
            if self.peek(j + stars)[1] == ")":                # (T) x, (unsigned int) x, (T *) p, (void **) &p
                self.next()
                tname = self.next()[1] if self.peek()[1] == "void" else self.parse_type()
                for _ in range(stars):
                    self.expect("*")
                self.expect(")")
                return ("cast", "ptr" if stars else tname, self.unary())
        if k == "id" and v == "sizeof" and self.peek(1)[1] == "(":
            self.next(); self.next()
            tname = self.parse_type()
            while self.accept("*"):
                tname = "ptr"
            self.expect(")")
            return ("sizeof", tname)
        return self.postfix()

    def postfix(self):
        k, v = self.next()
        if k == "num":
            txt = v.rstrip("fFuUlL")
            e = ("num", float(txt) if re.search(r"[.eE]", txt) else int(txt))
        elif k == "str":
            e = ("num", v[1:-1])
        elif k == "chr":
            e = ("num", v[1:-1])
        elif k == "id":
            e = ("name", v)
        elif v == "(":
            e = self.expr(); self.expect(")")
        else:
            raise CError(f"unexpected token {v!r}")
        while True:
            if self.accept("("):
                e = ("call", e, self.args())
            elif self.accept("["):
                i = self.expr(); self.expect("]")
                e = ("index", e, i)
            elif self.accept("."):
                e = ("member", e, self.next()[1])
            elif self.accept("->"):
                e = ("member", ("un", "*", e), self.next()[1])
            elif self.peek()[1] in ("++", "--") and self.peek()[0] == "op":
                e = ("post", self.next()[1], e)
            else:
                return e


class _Return(Exception):
    def __init__(self, value=None):
        self.value = value


class _Break(Exception):
    pass


class _Continue(Exception):
    pass


class SizeOf(int):
    """sizeof(T) counts ELEMENTS (value 1) and remembers T, so `n * sizeof(T)` handed to an allocator stub says what to allocate."""
    def __new__(cls, count, tname):
        o = int.__new__(cls, count)
        o.tname = tname
        return o

    def __mul__(self, other):
        return SizeOf(int(self) * int(other), self.tname) if isinstance(other, int) else NotImplemented
    __rmul__ = __mul__


# ------------------------------------------------------------------------------------------------ interpreter
class Machine:
    """Executes one parsed function body.  `builtins`: name -> Python callable (called with evaluated arguments) or value;
    `methods`: (type of object, method name) -> callable(obj, *args) for the objects the caller hands in."""

    def __init__(self, builtins, extra_types=()):
        self.builtins = dict(builtins)
        self.extra_types = set(extra_types)

    def parse(self, text):
        p = Parser(tokenize(text), self.extra_types)
        tree = p.parse_block_body()
        if p.peek()[0] != "eof":
            raise CError(f"trailing tokens at {p.peek()!r}")
        return tree

    def run(self, tree, variables):
        """Execute a parsed body; returns the outermost scope (parameters as the body left them) -- `returned` holds the value of
        a `return e;` if one ran."""
        self.scopes = [dict(variables)]
        self.types = [{}]
        self.returned = None
        try:
            self.exec(tree)
        except _Return as r:
            self.returned = r.value
        return self.scopes[0]

    def evaluate(self, expr_text, variables):
        """Value of one C expression."""
        p = Parser(tokenize(expr_text), self.extra_types)
        tree = p.expr()
        if p.peek()[0] != "eof":
            raise CError(f"trailing tokens at {p.peek()!r}")
        self.scopes = [dict(variables)]
        self.types = [{}]
        return self.val(tree)

    # -- variables
    def lookup(self, name):
        for s in reversed(self.scopes):
            if name in s:
                return s
        return None

    def declared_type(self, name):
        for t in reversed(self.types):
            if name in t:
                return t[name]
        return None

    @staticmethod
    def coerce(tname, v):
        if tname in INT_TYPES:
            return (1 if v else 0) if tname == "bool" else c_int(v)
        if tname in FLOAT_TYPES:
            return float(v)
        if tname in VEC_TYPES and isinstance(v, Vec):
            r = Vec(tname)
            for n in r.f:
                r.set(n, v.f[n])
            return r
        return v

    def default(self, tname):
        if tname in INT_TYPES:
            return 0
        if tname in FLOAT_TYPES:
            return float("nan")          # reading an uninitialised Scalar is a bug in the text, not in the interpreter: make it loud
        if tname in VEC_TYPES:
            v = Vec(tname)
            if VEC_TYPES[tname][1] is float:
                for n in v.f:
                    v.f[n] = float("nan")
            return v
        return None

    # -- statements
    def exec(self, s):
        kind = s[0]
        if kind == "block":
            self.scopes.append({}); self.types.append({})
            try:
                for x in s[1]:
                    self.exec(x)
            finally:
                self.scopes.pop(); self.types.pop()
        elif kind == "expr":
            self.val(s[1])
        elif kind == "decl":
            _, tname, decls = s
            for name, ptr, size, init, ctor in decls:
                if size is not None:
                    v = [self.default(tname) for _ in range(self.val(size))]
                    held = self.scopes[0].get(name)          # a __shared__ array the caller keeps across the threads of a block
                    if isinstance(held, list) and len(held) == len(v):
                        v = held
                elif ctor is not None:
                    v = self.builtins[tname](*[self.val(a) for a in ctor])
                elif init is not None:
                    v = self.val(init)
                    if isinstance(v, Vec):
                        v = v.copy()
                    if not ptr:
                        v = self.coerce(tname, v)
                else:
                    v = None if ptr else self.default(tname)
                self.scopes[-1][name] = v
                self.types[-1][name] = "ptr" if ptr or size is not None else tname
        elif kind == "if":
            if self.truth(self.val(s[1])):
                self.exec(s[2])
            elif s[3] is not None:
                self.exec(s[3])
        elif kind == "for":
            self.scopes.append({}); self.types.append({})
            try:
                if s[1] is not None:
                    self.exec(s[1])
                guard = 0
                while self.truth(self.val(s[2])):
                    depth = len(self.scopes)
                    try:
                        self.exec(s[4])
                    except _Break:
                        del self.scopes[depth:], self.types[depth:]
                        break
                    except _Continue:
                        del self.scopes[depth:], self.types[depth:]
                    self.val(s[3])
                    guard += 1
                    if guard > 10_000_000:
                        raise CError("runaway loop")
            finally:
                self.scopes.pop(); self.types.pop()
        elif kind == "while":
            guard = 0
            while self.truth(self.val(s[1])):
                depth = len(self.scopes)
                try:
                    self.exec(s[2])
                except _Break:
                    del self.scopes[depth:], self.types[depth:]
                    break
                except _Continue:
                    del self.scopes[depth:], self.types[depth:]
                guard += 1
                if guard > 10_000_000:
                    raise CError("runaway loop")
        elif kind == "break":
            raise _Break()
        elif kind == "continue":
            raise _Continue()
        elif kind == "return":
            raise _Return(None if s[1] is None else self.val(s[1]))
        else:
            raise CError(f"statement kind {kind}")

    @staticmethod
    def truth(v):
        return bool(v)

    # -- lvalues
    def ref(self, e):
        kind = e[0]
        if kind == "name":
            name = e[1]
            scope = self.lookup(name)
            if scope is None:
                raise CError(f"assignment to undeclared name {name}")
            tname = self.declared_type(name)

            def put(v, scope=scope, name=name, tname=tname):
                if isinstance(v, Vec):
                    v = v.copy()
                scope[name] = self.coerce(tname, v) if tname and tname != "ptr" else v
            return Ref(lambda scope=scope, name=name: scope[name], put)
        if kind == "member":
            obj = self.val(e[1])
            if not isinstance(obj, Vec):
                raise CError(f"member {e[2]} of a non-struct")
            return Ref(lambda: obj.get(e[2]), lambda v: obj.set(e[2], v))
        if kind == "index":
            base, i = self.val(e[1]), self.val(e[2])
            arr, off = (base.arr, base.off) if isinstance(base, Ptr) else (base, 0)

            def put(v, arr=arr, k=off + i):
                if isinstance(arr[k], Vec) and isinstance(v, Vec):
                    arr[k] = self.coerce(arr[k].tname, v)
                elif isinstance(arr[k], float):
                    arr[k] = float(v)
                else:
                    arr[k] = v
            return Ref(lambda arr=arr, k=off + i: arr[k], put)
        if kind == "un" and e[1] == "*":
            p = self.val(e[2])
            arr, off = (p.arr, p.off) if isinstance(p, Ptr) else (p, 0)
            return Ref(lambda: arr[off], lambda v: arr.__setitem__(off, v))
        raise CError(f"not an lvalue: {e[0]}")

    # -- expressions
    def val(self, e):
        kind = e[0]
        if kind == "num":
            return e[1]
        if kind == "name":
            scope = self.lookup(e[1])
            if scope is not None:
                return scope[e[1]]
            if e[1] in self.builtins:
                return self.builtins[e[1]]
            raise CError(f"unknown name {e[1]}")
        if kind == "member":
            obj = self.val(e[1])
            if isinstance(obj, Vec):
                return obj.get(e[2])
            return ("bound", obj, e[2])
        if kind == "index":
            return self.ref(e).get()
        if kind == "call":
            fn = e[1]
            args = [self.val(a) for a in e[2]]
            if fn[0] == "name" and self.lookup(fn[1]) is None:
                name = fn[1]
                if name in INT_TYPES or name in FLOAT_TYPES:
                    return self.coerce(name, args[0])
                if name not in self.builtins:
                    raise CError(f"call to unknown function {name}")
                return self.builtins[name](*args)
            f = self.val(fn)
            if isinstance(f, tuple) and f[0] == "bound":
                obj, meth = f[1], f[2]
                return getattr(obj, meth)(*args)
            raise CError("call of a non-function")
        if kind == "cast":
            return self.coerce(e[1], self.val(e[2]))
        if kind == "sizeof":
            return SizeOf(1, e[1])
        if kind == "cond":
            return self.val(e[2]) if self.truth(self.val(e[1])) else self.val(e[3])
        if kind == "comma":
            self.val(e[1])
            return self.val(e[2])
        if kind == "assign":
            r = self.ref(e[2])
            v = self.val(e[3])
            if e[1] != "=":
                v = self.arith(e[1][:-1], r.get(), v)
            r.put(v)
            return r.get()
        if kind == "post":
            r = self.ref(e[2])
            old = r.get()
            r.put(old + (1 if e[1] == "++" else -1))
            return old
        if kind == "un":
            op = e[1]
            if op in ("++", "--"):
                r = self.ref(e[2])
                r.put(r.get() + (1 if op == "++" else -1))
                return r.get()
            if op == "&":
                inner = e[2]
                if inner[0] == "index":
                    base, i = self.val(inner[1]), self.val(inner[2])
                    return Ptr(base.arr, base.off + i) if isinstance(base, Ptr) else Ptr(base, i)
                return self.ref(inner)                      # &(x.y): a reference the builtin (atomicAdd) writes through
            if op == "*":
                return self.ref(e).get()
            v = self.val(e[2])
            if op == "-":
                return v._bin(0, lambda a, _: -a) if isinstance(v, Vec) else -v
            if op == "+":
                return v
            return 0 if self.truth(v) else 1
        if kind == "bin":
            op = e[1]
            if op == "&&":
                return 1 if (self.truth(self.val(e[2])) and self.truth(self.val(e[3]))) else 0
            if op == "||":
                return 1 if (self.truth(self.val(e[2])) or self.truth(self.val(e[3]))) else 0
            return self.arith(op, self.val(e[2]), self.val(e[3]))
        raise CError(f"expression kind {kind}")

    @staticmethod
    def arith(op, a, b):
        if isinstance(a, Vec) or isinstance(b, Vec):
            fn = {"+": lambda x, y: x + y, "-": lambda x, y: x - y, "*": lambda x, y: x * y, "/": lambda x, y: x / y}[op]
            if isinstance(a, Vec):
                return a._bin(b, fn)
            return b._bin(a, lambda x, y: fn(y, x))          # scalar op Vec
        if isinstance(a, Ptr) and op in "+-":
            return Ptr(a.arr, a.off + (b if op == "+" else -b))
        if isinstance(a, bool):
            a = int(a)
        if isinstance(b, bool):
            b = int(b)
        both_int = isinstance(a, int) and isinstance(b, int)
        if op == "+":
            return a + b
        if op == "-":
            return a - b
        if op == "*":
            return a * b
        if op == "/":
            if both_int:
                q = abs(a) // abs(b)
                return q if (a >= 0) == (b >= 0) else -q    # truncation toward zero
            return a / b
        if op == "%":
            if not both_int:
                raise CError("% on non-integers")
            return int(math.fmod(a, b))                      # sign of the dividend
        if op in ("<<", ">>"):
            if not both_int:
                raise CError("shift of non-integers")
            return a << b if op == "<<" else a >> b
        if op == "<":
            return 1 if a < b else 0
        if op == ">":
            return 1 if a > b else 0
        if op == "<=":
            return 1 if a <= b else 0
        if op == ">=":
            return 1 if a >= b else 0
        if op == "==":
            return 1 if a == b else 0
        if op == "!=":
            return 1 if a != b else 0
        raise CError(f"operator {op}")


# ------------------------------------------------------------------------------------------------ source extraction
def function_source(text, name):
    """(parameter list text, body text) of the C function `name` in `text` (brace matching; comments kept for the tokenizer)."""
    m = re.search(r"\b" + re.escape(name) + r"\s*\(", text)
    if not m:
        raise CError(f"function {name} not found")
    i = m.end()
    depth, j = 1, i
    while depth:
        depth += {"(": 1, ")": -1}.get(text[j], 0)
        j += 1
    params = text[i:j - 1]
    k = text.index("{", j)
    depth, e = 1, k + 1
    while depth:
        depth += {"{": 1, "}": -1}.get(text[e], 0)
        e += 1
    return params, text[k + 1:e - 1]


def parameter_names(params):
    """[(type word, name)] of a C parameter list (pointers and references reduced to the name)."""
    params = re.sub(r"/\*.*?\*/", " ", params, flags=re.S)
    params = re.sub(r"//[^\n]*", " ", params)
    out = []
    for p in params.split(","):
        p = p.strip()
        if not p:
            continue
        words = re.findall(r"[A-Za-z_]\w*", p)
        out.append((" ".join(w for w in words[:-1] if w != "const"), words[-1]))
    return out
