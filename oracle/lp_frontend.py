"""TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Pure-Python restatement of the reference's text front-end for the tableau
path: lexer/parser (reference parser.cc:3-297), expression algebra
(base.cc:196-286, base.h:200-231), ToStandardForm / ToSlackForm / ToTableau
(lp.cc:55-197), PrintTableau (lp.cc:199-219), solution extraction
(lp_simplex_tableau.cc:381-414) and the CLI output format (solver.cc:160-171).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.  It is checked against the reference's golden files by
oracle/pin_oracle.py and tests/test_oracle_goldens.py.
"""
import ctypes
import functools

import numpy as np

_libc = ctypes.CDLL(None)
_libc.strtof.restype = ctypes.c_float
_libc.strtof.argtypes = [ctypes.c_char_p, ctypes.c_void_p]

EPS32 = np.float32(1e-6)


def stof(text):
    """std::stof then widen to double (reference parser.cc:179,188)."""
    return float(_libc.strtof(text.encode(), None))


def is_zero(x):  # base.h:65, float-typed
    return bool(abs(np.float32(x)) < EPS32)


def var_less(a, b):
    """Variable::operator< for FLOAT variables (reference base.cc:11-29)."""
    if a.startswith("dual") and b.startswith("dual"):
        return int(a[4:]) < int(b[4:])
    if a.startswith("artificial") and b.startswith("artificial"):
        return int(a[10:]) < int(b[10:])
    return a < b


def _cmp(a, b):
    return -1 if var_less(a, b) else (1 if var_less(b, a) else 0)


var_key = functools.cmp_to_key(_cmp)


def sorted_vars(names):
    return sorted(names, key=var_key)


class Expr:
    """Expression = {variable: coefficient} + constant (base.h:200-231)."""

    def __init__(self, const=0.0):
        self.c = {}
        self.k = float(const)

    def copy(self):
        e = Expr(self.k)
        e.c = dict(self.c)
        return e

    def get(self, v):
        return self.c.get(v, 0.0)

    def set(self, v, x):  # SetCoeffOf, base.h:212-219: exact zeros are not stored
        self.c.pop(v, None)
        if x != 0.0:
            self.c[v] = x

    def iadd(self, o):  # base.cc:203-213
        for v in sorted_vars(o.c):
            x = o.c[v]
            l = self.get(v)
            self.set(v, l + x if l != 0.0 else x)
        self.k = self.k + o.k

    def isub(self, o):  # base.cc:220-230
        for v in sorted_vars(o.c):
            x = o.c[v]
            l = self.get(v)
            self.set(v, l - x if l != 0.0 else -1.0 * x)
        self.k = self.k - o.k

    def imul(self, mult):  # base.cc:237-252: float-zero products are erased
        if mult == 0.0:
            self.c = {}
            self.k = mult
        for v in list(self.c):
            self.c[v] = self.c[v] * mult
            if is_zero(self.c[v]):
                del self.c[v]
        self.k = self.k * mult

    def to_string(self):  # base.h:221-227
        s = ""
        for v in sorted_vars(self.c):
            s += "%f * %s + " % (self.c[v], v)
        return s + "%f" % self.k


def replace_var(expr, var, sub):  # base.cc:281-286
    coeff = expr.get(var)
    expr.set(var, 0.0)
    t = sub.copy()
    t.imul(coeff)
    expr.iadd(t)


# ---------------------------------------------------------------- lexer/parser
def scan(text):
    """Lexer::Scan (parser.cc:3-122) -> list of (type, lexeme)."""
    toks = []
    cur = None  # [type, start]

    def close(i):
        nonlocal cur
        if cur is not None:
            toks.append((cur[0], text[cur[1]:i]))
            cur = None

    i = 0
    n = len(text)
    while i < n:
        ch = text[i]
        if ch == "\n":
            close(i)
            toks.append(("EOL", "\n"))
        elif ch in "\0 \t":
            close(i)
        elif ch.isascii() and (ch.isalpha() or ch == "_"):
            if cur is None:
                cur = ["VAR", i]
            elif cur[0] == "NUM":
                raise RuntimeError("cannot concat num + var")
        elif ch.isascii() and ch.isdigit():
            if cur is None:
                cur = ["NUM", i]
        elif ch == ".":
            if cur is None:
                cur = ["NUM", i]
            else:
                assert cur[0] == "NUM"
        elif ch in "+-*><=,":
            close(i)
            if ch == ">":
                assert text[i + 1] == "="
                toks.append(("GE", ">="))
                i += 1
            elif ch == "<":
                assert text[i + 1] == "="
                toks.append(("LE", "<="))
                i += 1
            else:
                toks.append(({"+": "ADD", "-": "SUB", "*": "MUL", "=": "EQ", ",": "COMMA"}[ch], ch))
        else:
            raise RuntimeError("Lexing error: un-recognized symbol")
        i += 1
    close(n)
    out = []
    for t, lx in toks:
        if t == "VAR":
            if lx in ("MAX", "max"):
                t = "MAX"
            elif lx in ("MIN", "min"):
                t = "MIN"
            elif lx in ("ST", "st"):
                t = "ST"
        out.append((t, lx))
    return out


def parse_expression(tokens):  # parser.cc:168-219
    ret = Expr(0.0)
    sign = True
    group = []

    def unary(g):
        assert len(g) in (1, 3)
        if len(g) == 1:
            t, lx = g[0]
            assert t in ("NUM", "VAR")
            if t == "NUM":
                return Expr(stof(lx))
            e = Expr(0.0)
            e.c[lx] = 1.0
            return e
        assert g[0][0] == "NUM" and g[1][0] == "MUL" and g[2][0] == "VAR"
        e = Expr(0.0)
        e.set(g[2][1], stof(g[0][1]))
        return e

    for i, tk in enumerate(tokens):
        if tk[0] in ("ADD", "SUB"):
            if i != 0:
                ex = unary(group)
                ret.iadd(ex) if sign else ret.isub(ex)
            sign = tk[0] == "ADD"
            group = []
        else:
            group.append(tk)
    if group:
        ex = unary(group)
        ret.iadd(ex) if sign else ret.isub(ex)
    return ret


class Constraint:
    def __init__(self, expr, typ, compare=0.0):
        self.expr, self.typ, self.compare = expr, typ, compare

    def to_string(self):  # base.h:284-302
        return "%s %s %f" % (self.expr.to_string(), {"LE": "<=", "GE": ">=", "EQ": "="}[self.typ], self.compare)


def parse_constraint(tokens):  # parser.cc:141-167
    idx = [i for i, t in enumerate(tokens) if t[0] in ("EQ", "GE", "LE")]
    if len(idx) > 1:
        raise RuntimeError("More than one compare symbol found in one constraint")
    assert idx
    k = idx[0]
    e = parse_expression(tokens[:k])
    e.isub(parse_expression(tokens[k + 1:]))
    return Constraint(e, tokens[k][0])


class Model:
    """LPModel state that the tableau path needs (lp.h:264-279)."""

    def __init__(self):
        self.opt_type = "MIN"
        self.obj = Expr(0.0)
        self.cons = []
        self.base_vars = set()
        self.non_base_vars = set()
        self.opt_reverted = False
        self.raw_expr = {}
        self.n_base = 0
        self.n_subst = 0

    def to_string(self):  # lp.h:234-241
        s = ("min " if self.opt_type == "MIN" else "max ") + self.obj.to_string() + "\n"
        for c in self.cons:
            s += c.to_string() + "\n"
        return s


def parse(text):  # parser.cc:221-283
    toks = scan(text)
    lines, line = [], []
    for t in toks:
        if t[0] == "EOL":
            lines.append(line)
            line = []
        else:
            line.append(t)
    if line:
        lines.append(line)
    assert len(lines) >= 2 and len(lines[0]) >= 2
    assert len(lines[1]) == 1 and lines[1][0][0] == "ST"
    assert lines[0][0][0] in ("MAX", "MIN")
    if any(t[0] == "COMMA" for t in lines[-1]):
        raise NotImplementedError("integer-variable line: not on the tableau LP path")
    m = Model()
    m.opt_type = lines[0][0][0]
    m.obj = parse_expression(lines[0][1:])
    for ln in lines[2:]:
        m.cons.append(parse_constraint(ln))
    return m


def parse_file(path):  # parser.cc:285-297 (std::getline loop re-adds "\n" to every line)
    with open(path) as f:
        content = f.read()
    lines = content.split("\n")
    if lines and lines[-1] == "":
        lines.pop()
    return parse("".join(l + "\n" for l in lines))


def read_text(path):
    """File content exactly as Parser::Parse(std::ifstream&) hands it on."""
    with open(path) as f:
        content = f.read()
    lines = content.split("\n")
    if lines and lines[-1] == "":
        lines.pop()
    return "".join(l + "\n" for l in lines)


# ---------------------------------------------------------------- transforms
def to_standard_form(m):  # lp.cc:55-129
    if m.opt_type == "MIN":
        m.opt_reverted = True
        m.obj.imul(-1.0)
        m.opt_type = "MAX"
    extra = []
    for c in m.cons:
        if c.typ == "EQ":
            c.typ = "GE"
            extra.append(Constraint(c.expr.copy(), "LE", c.compare))
    m.cons.extend(extra)
    for c in m.cons:
        c.compare = c.compare - c.expr.k
        c.expr.k = 0.0
        if c.typ == "GE":
            c.expr.imul(-1.0)
            c.compare = c.compare * -1.0
            c.typ = "LE"
    for c in m.cons:
        m.non_base_vars.update(c.expr.c.keys())
    negative = set(m.non_base_vars)
    drop = []
    for i, c in enumerate(m.cons):
        if not _is_nonneg_constraint(c):
            continue
        negative.discard(next(iter(c.expr.c)))
        drop.append(i)
    for i in reversed(drop):
        del m.cons[i]
    for v in sorted_vars(negative):
        m.non_base_vars.discard(v)
        s1 = "subst%d" % m.n_subst
        s2 = "subst%d" % (m.n_subst + 1)
        m.n_subst += 2
        ex = Expr(0.0)
        ex.set(s1, 1.0)
        t = Expr(0.0)
        t.set(s2, 1.0)
        ex.isub(t)
        m.raw_expr[v] = ex
        for c in m.cons:
            replace_var(c.expr, v, ex)
        replace_var(m.obj, v, ex)
        m.non_base_vars.update((s1, s2))


def _is_nonneg_constraint(c):  # lp.cc:298-306
    if c.typ != "LE" or not is_zero(c.compare) or not is_zero(c.expr.k):
        return False
    if len(c.expr.c) != 1:
        return False
    return next(iter(c.expr.c.values())) == -1.0


def to_slack_form(m):  # lp.cc:135-145
    for c in m.cons:
        b = "base%d" % m.n_base
        m.n_base += 1
        e = Expr(c.compare)
        e.isub(c.expr)
        c.expr = e
        c.compare = 0.0
        c.typ = "EQ"
        t = Expr(0.0)
        t.set(b, -1.0)
        c.expr.iadd(t)
        m.base_vars.add(b)


class Tableau:
    """Dense view of lp.cc:147-197.  T is m x (N+2) (column N = RHS, column
    N+1 = phase-1 artificial, all zero while inactive)."""

    def __init__(self, m):
        names = sorted_vars(set(m.base_vars) | set(m.non_base_vars))
        self.names = names
        self.N = len(names)
        self.m = len(m.cons)
        idx = {v: j for j, v in enumerate(names)}
        self.index = idx
        self.T = np.zeros((self.m, self.N + 2), dtype=np.float64, order="F")
        self.obj = np.zeros(self.N + 2, dtype=np.float64)
        self.is_basic = np.zeros(self.N + 2, dtype=np.uint8)
        for v in m.base_vars:
            self.is_basic[idx[v]] = 1
        for i, c in enumerate(m.cons):
            for v, x in c.expr.c.items():
                if not is_zero(x):
                    self.T[i, idx[v]] = x
            self.T[i, self.N] = c.expr.k
        for v, x in m.obj.c.items():
            if v in idx and not is_zero(x):
                self.obj[idx[v]] = x
        if not is_zero(m.obj.k):
            self.obj[self.N] = m.obj.k
        self.opt_reverted = m.opt_reverted
        self.raw_expr = m.raw_expr


def build_tableau(text):
    m = parse(text)
    to_standard_form(m)
    to_slack_form(m)
    return Tableau(m)


def print_tableau(T, obj, names, N, art_name="artificial0"):  # lp.cc:199-219
    def row(vals):
        parts = []
        for j, x in enumerate(vals):
            if x == 0.0:
                continue
            s = "%f" % x
            if j != N:
                s += " * " + (names[j] if j < N else art_name)
            parts.append(s)
        return " + ".join(parts)

    out = row(obj) + "\n"
    for i in range(T.shape[0]):
        out += row(T[i, :]) + "\n"
    return out


def is_user_defined(name):  # lp.cc:3-8
    return not (name.startswith("base") or name.startswith("subst") or name.startswith("dual") or name.startswith("artificial"))


def format_solution(tab, values, obj_const):
    """solver.cc:160-164 + lp_simplex_tableau.cc:381-414."""
    opt = (-1 if tab.opt_reverted else 1) * obj_const
    all_sol = {v: float(values[j]) for j, v in enumerate(tab.names)}
    sol = {v: x for v, x in all_sol.items() if is_user_defined(v)}
    for raw, ex in tab.raw_expr.items():
        e = ex.copy()
        while e.c:
            v = sorted_vars(e.c)[0]
            replace_var(e, v, Expr(all_sol[v]))
        sol[raw] = e.k
    out = "%f\n" % opt
    for v in sorted_vars(sol):
        out += "%s = %f\n" % (v, sol[v])
    return out
