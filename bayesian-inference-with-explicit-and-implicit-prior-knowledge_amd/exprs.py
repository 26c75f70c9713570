"""Model callables as data: a state-space model written against an array namespace `xp` (as pgas_amd.experiments writes the
reference's models: `model(xp) -> (transition_model, output_model)`) is traced ONCE with the symbolic namespace below into a short
register program over per-particle scalars, and pgas_m_expr_eval (csrc/pgas_marginal.hip.h, k_expr) runs that program for every
particle in ONE launch -- instead of the ~45 elementwise torch launches of an RK4 transition.  Not a compiler: the kernel is fixed,
the program is a descriptor it interprets (opcode, destination, two sources per instruction; inputs and constants preloaded).

What the namespace understands is what the reference's models use (src/SingleMassOscillator.py:32-48, src/Vehicle.py:62-131,
src/EMPS.py:158-198, src/Toy_Example.py:69-70): + - * / and unary minus with numbers and each other, `x[:, j]`, `x[:, a:b]`, `u[j]`,
`.reshape(-1)`, `.reshape(-1, 1)`, `.reshape(-1)[0]`, `xp.stack([...], 1)`, `xp.cos / sin / tan / tanh / arctan / sqrt / exp / sign`.
Anything else raises TypeError at trace time and the caller keeps the torch callables.
"""
from __future__ import annotations

import numpy as np

OPS = {"add": 1, "sub": 2, "mul": 3, "div": 4, "neg": 5, "cos": 6, "sin": 7, "tan": 8, "tanh": 9, "arctan": 10, "sqrt": 11, "exp": 12,
       "sign": 13, "mov": 14}
MAX_REG = 96   # PG_EX_MAXREG of the kernel


class _Trace:
    def __init__(self):
        self.nodes = []          # ("in", slot) | ("const", value) | (opname, a, b)
        self._memo = {}

    def node(self, *key):
        if key not in self._memo:
            self._memo[key] = len(self.nodes)
            self.nodes.append(key)
        return self._memo[key]

    def const(self, v):
        return self.node("const", float(v))


class Sym:
    """cols: node ids.  kind: "pp2" (N, k) per particle, "pp1" (N,) per particle, "u1" (k,) uniform vector, "u0" uniform scalar."""

    __array_priority__ = 1000
    __array_ufunc__ = None   # numpy scalars (np.float64 constants of a model) defer to the reflected operators below

    def __init__(self, tr, cols, kind):
        self.tr, self.cols, self.kind = tr, list(cols), kind

    # ---- shapes
    @property
    def shape(self):
        return {"pp2": (None, len(self.cols)), "pp1": (None,), "u1": (len(self.cols),), "u0": ()}[self.kind]

    def reshape(self, *shape):
        shape = shape[0] if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else shape
        if tuple(shape) == (-1,):
            if self.kind == "pp2" and len(self.cols) == 1:
                return Sym(self.tr, self.cols, "pp1")
            if self.kind in ("pp1", "u1"):
                return self
            if self.kind == "u0":
                return Sym(self.tr, self.cols, "u1")
        if tuple(shape) == (-1, 1) and self.kind in ("pp1", "pp2") and len(self.cols) == 1:
            return Sym(self.tr, self.cols, "pp2")
        raise TypeError(f"symbolic trace: reshape{tuple(shape)} of a {self.kind} value with {len(self.cols)} component(s) is not supported")

    def __getitem__(self, idx):
        if self.kind == "u1" and isinstance(idx, (int, np.integer)):
            return Sym(self.tr, [self.cols[idx]], "u0")
        if self.kind == "pp2" and isinstance(idx, tuple) and len(idx) == 2 and idx[0] == slice(None):
            j = idx[1]
            if isinstance(j, (int, np.integer)):
                return Sym(self.tr, [self.cols[j]], "pp1")
            if isinstance(j, slice):
                return Sym(self.tr, self.cols[j], "pp2")
        raise TypeError(f"symbolic trace: index {idx!r} on a {self.kind} value is not supported")

    # ---- arithmetic
    def _lift(self, other):
        if isinstance(other, Sym):
            return other
        if isinstance(other, (int, float, np.integer, np.floating)) or (isinstance(other, np.ndarray) and other.ndim == 0):
            return Sym(self.tr, [self.tr.const(float(other))], "u0")
        raise TypeError(f"symbolic trace: operand of type {type(other).__name__} is not supported")

    def _bin(self, op, other, swap=False):
        a, b = (self._lift(other), self) if swap else (self, self._lift(other))
        per = [x for x in (a, b) if x.kind in ("pp1", "pp2")]
        if a.kind in ("u0", "u1") and b.kind in ("u0", "u1"):
            kind = "u1" if "u1" in (a.kind, b.kind) else "u0"
        elif len(per) == 2 and per[0].kind != per[1].kind:
            raise TypeError("symbolic trace: (N,) with (N, k) operands are not broadcast (write x[:, j] explicitly)")
        else:
            kind = per[0].kind
        n = max(len(a.cols), len(b.cols))
        if len(a.cols) not in (1, n) or len(b.cols) not in (1, n):
            raise TypeError("symbolic trace: operand widths do not match")
        if (a.kind == "u1" and len(a.cols) > 1 and kind.startswith("pp")) or (b.kind == "u1" and len(b.cols) > 1 and kind.startswith("pp")):
            raise TypeError("symbolic trace: a uniform vector is not broadcast against particles (index it: u[j])")
        ac = a.cols * n if len(a.cols) == 1 else a.cols
        bc = b.cols * n if len(b.cols) == 1 else b.cols
        return Sym(self.tr, [self.tr.node(op, x, y) for x, y in zip(ac, bc)], kind)

    def __add__(self, o): return self._bin("add", o)
    def __radd__(self, o): return self._bin("add", o, True)
    def __sub__(self, o): return self._bin("sub", o)
    def __rsub__(self, o): return self._bin("sub", o, True)
    def __mul__(self, o): return self._bin("mul", o)
    def __rmul__(self, o): return self._bin("mul", o, True)
    def __truediv__(self, o): return self._bin("div", o)
    def __rtruediv__(self, o): return self._bin("div", o, True)
    def __neg__(self): return Sym(self.tr, [self.tr.node("neg", c, c) for c in self.cols], self.kind)


class SymNamespace:
    """The `xp` handed to model(xp) while tracing."""

    def __init__(self, tr):
        self.tr = tr

    def stack(self, parts, axis=1):
        if axis != 1 or not all(isinstance(p, Sym) and p.kind == "pp1" for p in parts):
            raise TypeError("symbolic trace: stack([...], 1) of per-particle (N,) values only")
        return Sym(self.tr, [p.cols[0] for p in parts], "pp2")

    def _un(self, op, x):
        if not isinstance(x, Sym):
            raise TypeError(f"symbolic trace: {op} of a non-symbolic value")
        return Sym(self.tr, [self.tr.node(op, c, c) for c in x.cols], x.kind)

    def cos(self, x): return self._un("cos", x)
    def sin(self, x): return self._un("sin", x)
    def tan(self, x): return self._un("tan", x)
    def tanh(self, x): return self._un("tanh", x)
    def arctan(self, x): return self._un("arctan", x)
    def sqrt(self, x): return self._un("sqrt", x)
    def exp(self, x): return self._un("exp", x)
    def sign(self, x): return self._un("sign", x)


class Program:
    """code (n_instr, 4) int32 = (opcode, dst, a, b); preload: registers [0, n_in) = inputs (state columns, input components, interface
    variable components, in that order), [n_in, n_in + n_const) = constants; out_regs: the registers holding the results."""

    def __init__(self, code, consts, n_in, n_reg, out_regs, widths):
        self.code = np.asarray(code, dtype=np.int32).reshape(-1, 4)
        self.consts = np.asarray(consts, dtype=np.float64)
        self.n_in, self.n_reg, self.out_regs, self.widths = int(n_in), int(n_reg), [int(r) for r in out_regs], widths


def trace(fn, nx, nu, iv_widths):
    """Trace fn(state (N, nx), input (nu,), *int_var (N, w_i)) -> (N, k) [or (N,)] into a Program, or raise TypeError."""
    tr = _Trace()
    slot = 0
    state = Sym(tr, [tr.node("in", slot + j) for j in range(nx)], "pp2"); slot += nx
    inp = Sym(tr, [tr.node("in", slot + j) for j in range(nu)], "u1"); slot += nu
    ivs = []
    for w in iv_widths:
        ivs.append(Sym(tr, [tr.node("in", slot + j) for j in range(w)], "pp2")); slot += w
    out = fn(state, inp, *ivs)
    if not isinstance(out, Sym) or out.kind not in ("pp1", "pp2"):
        raise TypeError("symbolic trace: the model must return a per-particle value built from its arguments")
    # ---- linearise the part of the graph the outputs need; registers by last use
    need, order = set(), []

    def visit(i):
        if i in need:
            return
        nd = tr.nodes[i]
        if nd[0] not in ("in", "const"):
            visit(nd[1]); visit(nd[2])
        need.add(i); order.append(i)

    for c in out.cols:
        visit(c)
    consts = [i for i in order if tr.nodes[i][0] == "const"]
    reg = {}
    for i in order:
        if tr.nodes[i][0] == "in":
            reg[i] = tr.nodes[i][1]
    for k, i in enumerate(consts):
        reg[i] = slot + k
    last = {}
    ops = [i for i in order if tr.nodes[i][0] not in ("in", "const")]
    for pos, i in enumerate(ops):
        for a in tr.nodes[i][1:3]:
            last[a] = pos
    for c in out.cols:
        last[c] = len(ops) + 1
    free, nxt, code = [], slot + len(consts), []
    for pos, i in enumerate(ops):
        op, a, b = tr.nodes[i]
        ra, rb = reg[a], reg[b]
        for src in {a, b}:   # a temporary dies at its last use; its register may be this instruction's destination
            if last.get(src) == pos and tr.nodes[src][0] not in ("in", "const"):
                free.append(reg[src])
        if free:
            rd = free.pop()
        else:
            rd, nxt = nxt, nxt + 1
        reg[i] = rd
        code.append((OPS[op], rd, ra, rb))
    out_regs = []
    for c in out.cols:   # an output that is an input / constant, or the same node twice, gets its own copy
        if tr.nodes[c][0] in ("in", "const") or reg[c] in out_regs:
            rd, nxt = nxt, nxt + 1
            code.append((OPS["mov"], rd, reg[c], reg[c]))
            out_regs.append(rd)
        else:
            out_regs.append(reg[c])
    if nxt > MAX_REG:
        raise TypeError(f"symbolic trace: the model needs {nxt} registers, the kernel has {MAX_REG}")
    return Program(code, [tr.nodes[i][1] for i in consts], slot, nxt, out_regs, (nx, nu, tuple(iv_widths)))


def run_numpy(prog, state, inp, ivs):
    """Reference interpreter (tests): the same program on NumPy arrays."""
    N = state.shape[0]
    r = np.zeros((prog.n_reg, N))
    cols = [state[:, j] for j in range(state.shape[1])] + [np.full(N, float(v)) for v in np.asarray(inp, dtype=np.float64).reshape(-1)]
    for v in ivs:
        v = np.asarray(v, dtype=np.float64).reshape(N, -1)
        cols += [v[:, j] for j in range(v.shape[1])]
    for k, c in enumerate(cols):
        r[k] = c
    for k, c in enumerate(prog.consts):
        r[prog.n_in + k] = c
    f = {1: np.add, 2: np.subtract, 3: np.multiply, 4: np.divide}
    g = {5: np.negative, 6: np.cos, 7: np.sin, 8: np.tan, 9: np.tanh, 10: np.arctan, 11: np.sqrt, 12: np.exp, 13: np.sign, 14: lambda x: x}
    for op, d, a, b in prog.code:
        r[d] = f[op](r[a], r[b]) if op in f else g[op](r[a])
    return np.stack([r[k] for k in prog.out_regs], 1)
