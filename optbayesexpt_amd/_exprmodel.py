"""Device models from a user's formula.

The reference takes the measurement model as an arbitrary Python callable.  For analytic
models the same formula can be given as an expression string; it is translated once to
(1) a HIP ``PluginModel`` struct with the interface the kernels are templated on — the
kernel sources are then compiled for that model into a plugin library by
``optbayesexpt_amd.build.build_plugin`` (hipcc, cached by content hash) — and (2) a NumPy
callable with the reference's ``model(settings, parameters, constants)`` signature for
user-side simulation.  With ``-ffp-contract=off`` the device evaluates the expression with
one correctly rounded operation per node, like NumPy does.
"""
import ast
import hashlib
import re

import numpy as np

# name in the expression -> (C function, NumPy function, arity)
_FUNCS = {
    "exp": ("exp", np.exp, 1), "log": ("log", np.log, 1), "log1p": ("log1p", np.log1p, 1),
    "expm1": ("expm1", np.expm1, 1), "sqrt": ("sqrt", np.sqrt, 1), "sin": ("sin", np.sin, 1),
    "cos": ("cos", np.cos, 1), "tan": ("tan", np.tan, 1), "tanh": ("tanh", np.tanh, 1),
    "sinh": ("sinh", np.sinh, 1), "cosh": ("cosh", np.cosh, 1), "arctan": ("atan", np.arctan, 1),
    "atan": ("atan", np.arctan, 1), "arctan2": ("atan2", np.arctan2, 2), "atan2": ("atan2", np.arctan2, 2),
    "hypot": ("hypot", np.hypot, 2), "abs": ("fabs", np.abs, 1), "fabs": ("fabs", np.abs, 1),
    "minimum": ("fmin", np.minimum, 2), "maximum": ("fmax", np.maximum, 2),
}
_CONSTS = {"pi": float(np.pi), "e": float(np.e)}


class _ToC(ast.NodeVisitor):
    def __init__(self, names):
        self.names = names

    def visit_Expression(self, node):
        return self.visit(node.body)

    def visit_Constant(self, node):
        if isinstance(node.value, bool) or not isinstance(node.value, (int, float)):
            raise ValueError(f"unsupported constant {node.value!r}")
        return repr(float(node.value))

    def visit_Name(self, node):
        if node.id in self.names:
            return f"u_{node.id}"
        if node.id in _CONSTS:
            return repr(_CONSTS[node.id])
        raise ValueError(f"unknown name {node.id!r} (not a setting, parameter, constant, pi or e)")

    def visit_UnaryOp(self, node):
        v = self.visit(node.operand)
        if isinstance(node.op, ast.USub):
            return f"(-{v})"
        if isinstance(node.op, ast.UAdd):
            return v
        raise ValueError("unsupported unary operator")

    def visit_BinOp(self, node):
        a, b = self.visit(node.left), self.visit(node.right)
        if isinstance(node.op, ast.Add):
            return f"({a} + {b})"
        if isinstance(node.op, ast.Sub):
            return f"({a} - {b})"
        if isinstance(node.op, ast.Mult):
            return f"({a} * {b})"
        if isinstance(node.op, ast.Div):
            return f"({a} / {b})"
        if isinstance(node.op, ast.Pow):
            if isinstance(node.right, ast.Constant) and node.right.value == 2:
                return f"sq({a})"                       # NumPy squares by multiplication
            if isinstance(node.right, ast.Constant) and node.right.value == 0.5:
                return f"sqrt({a})"
            return f"pow({a}, {b})"
        raise ValueError("unsupported binary operator")

    def visit_Call(self, node):
        if not isinstance(node.func, ast.Name) or node.func.id not in _FUNCS or node.keywords:
            raise ValueError("unsupported function call")
        cname, _, arity = _FUNCS[node.func.id]
        if len(node.args) != arity:
            raise ValueError(f"{node.func.id} takes {arity} argument(s)")
        return f"{cname}({', '.join(self.visit(a) for a in node.args)})"

    def generic_visit(self, node):
        raise ValueError(f"unsupported syntax: {type(node).__name__}")


# ---------------------------------------------------------------------------
# Sweep form.  The K1 kernel evaluates the model ns x n times, so the generated sweep code
# is a small compiler pass over the expression rather than a transliteration:
#   * every node has a level: C (constants only), S (settings), P (parameters, sqrt(w)), SP;
#   * maximal S-level subtrees are computed once per setting (prep_setting -> xs slots),
#     maximal C/P-level subtrees once per particle (pack -> pk slots of the packed record the sweep streams);
#   * products are distributed / re-associated when that moves work out of the SP level
#     ((x - x0)/d -> x/d - x0/d;  sw*(b + a*r) -> (sw*a)*r + sw*b);
#   * divisions by SP-level denominators are batched over the SPT settings a lane owns
#     (batch_rcp_guarded: one v_rcp_f64 per batch); a*b + c is emitted as fma.
# Only the sweep uses this form (utilities agree with the exact form to ~1e-13 relative);
# the Bayes update and eval_over_* use the exact one-operation-per-node `formula`.
_LC, _LS, _LP, _LSP = 0, 1, 2, 3
_FAST_CALLS = {"cos": "fast_cos", "sin": "fast_sin", "sqrt": "fast_sqrt", "hypot": "fast_hypot"}


class _Node:
    __slots__ = ("op", "args", "val", "level", "key")

    def __init__(self, op, args=(), val=None, level=None):
        self.op, self.args, self.val = op, tuple(args), val
        if level is None:
            level = 0
            for a in self.args:
                level |= a.level
        self.level = level
        self.key = (op, val, tuple(a.key for a in self.args))


def _num(v):
    return _Node("num", val=float(v), level=_LC)


def _mul(a, b):
    """a*b, pushing a non-SP factor into an SP-level sum / product / quotient when that
    takes at least one multiplication out of the SP level."""
    for x, f in ((a, b), (b, a)):
        if f.op == "num" and f.val == 1.0:
            return x
        if x.level != _LSP or f.level == _LSP:
            continue
        if x.op in ("add", "sub") and any((t.level | f.level) != _LSP for t in x.args):
            return _Node(x.op, (_mul(x.args[0], f), _mul(x.args[1], f)))
        if x.op == "neg":
            return _Node("neg", (_mul(x.args[0], f),))
        if x.op == "mul":
            p, q = x.args
            if (p.level | f.level) != _LSP:
                return _mul(_mul(p, f), q)
            if (q.level | f.level) != _LSP:
                return _mul(p, _mul(q, f))
        if x.op == "div" and (x.args[0].level | f.level) != _LSP:
            return _div(_mul(x.args[0], f), x.args[1])
    return _Node("mul", (a, b))


def _div(a, b):
    if (a.level | b.level) != _LSP or b.level == _LSP:
        return _Node("div", (a, b))
    return _mul(a, _Node("div", (_num(1.0), b)))      # reciprocal hoisted to b's level


class _SweepBuilder(ast.NodeVisitor):
    def __init__(self, settings, parameters, constants):
        self.sets = {n: k for k, n in enumerate(settings)}
        self.pars = {n: k for k, n in enumerate(parameters)}
        self.cons = {n: k for k, n in enumerate(constants)}

    def visit_Expression(self, node):
        return self.visit(node.body)

    def visit_Constant(self, node):
        if isinstance(node.value, bool) or not isinstance(node.value, (int, float)):
            raise ValueError(f"unsupported constant {node.value!r}")
        return _num(node.value)

    def visit_Name(self, node):
        if node.id in self.sets:
            return _Node("set", val=self.sets[node.id], level=_LS)
        if node.id in self.pars:
            return _Node("par", val=self.pars[node.id], level=_LP)
        if node.id in self.cons:
            return _Node("con", val=self.cons[node.id], level=_LC)
        if node.id in _CONSTS:
            return _num(_CONSTS[node.id])
        raise ValueError(f"unknown name {node.id!r}")

    def visit_UnaryOp(self, node):
        v = self.visit(node.operand)
        if isinstance(node.op, ast.USub):
            return _Node("neg", (v,))
        if isinstance(node.op, ast.UAdd):
            return v
        raise ValueError("unsupported unary operator")

    def visit_BinOp(self, node):
        a, b = self.visit(node.left), self.visit(node.right)
        if isinstance(node.op, ast.Add):
            return _Node("add", (a, b))
        if isinstance(node.op, ast.Sub):
            return _Node("sub", (a, b))
        if isinstance(node.op, ast.Mult):
            return _mul(a, b)
        if isinstance(node.op, ast.Div):
            return _div(a, b)
        if isinstance(node.op, ast.Pow):
            if b.op == "num" and b.val == 2.0:
                return _Node("sq", (a,))
            if b.op == "num" and b.val == 0.5:
                return _Node("call", (a,), val="sqrt")
            return _Node("call", (a, b), val="pow")
        raise ValueError("unsupported binary operator")

    def visit_Call(self, node):
        if not isinstance(node.func, ast.Name) or node.func.id not in _FUNCS or node.keywords:
            raise ValueError("unsupported function call")
        cname, _, arity = _FUNCS[node.func.id]
        if len(node.args) != arity:
            raise ValueError(f"{node.func.id} takes {arity} argument(s)")
        return _Node("call", [self.visit(a) for a in node.args], val=cname)

    def generic_visit(self, node):
        raise ValueError(f"unsupported syntax: {type(node).__name__}")


class _SweepEmitter:
    """C code for prep_setting (xs slots), pack (pk slots) and the phased sweep_eval body."""

    def __init__(self, safe=False, share=None):
        """``safe``: one guarded IEEE reciprocal per element instead of the branch-free batched
        inversions (the sweep_eval_safe twin); ``share``: the emitter of the fast form, whose
        xs / pk slot tables this one extends so that both forms read the same prepared data."""
        self.safe = safe
        if share is None:
            self.xs_slots, self.pk_slots = {}, {}     # node key -> slot index
            self.prep, self.pack = [], []             # C statements filling the slots
        else:
            self.xs_slots, self.pk_slots, self.prep, self.pack = share.xs_slots, share.pk_slots, share.prep, share.pack
        self.phases, self.memo = [], {}
        self.count = 0

    # exact (one operation per node) form of a hoisted subtree, inside prep_setting / pack
    def exact(self, n):
        a = [self.exact(c) for c in n.args]
        if n.op == "num":
            return repr(n.val)
        if n.op == "set":
            return f"x[{n.val}]"
        if n.op == "par":
            return f"th({n.val})"
        if n.op == "con":
            return f"m.consts[{n.val}]"
        if n.op == "sw":
            return "sw"
        if n.op == "neg":
            return f"(-{a[0]})"
        if n.op == "sq":
            return f"sq({a[0]})"
        if n.op == "call":
            return f"{n.val}({', '.join(a)})"
        return f"({a[0]} {dict(add='+', sub='-', mul='*', div='/')[n.op]} {a[1]})"

    def slot(self, n):
        if n.level == _LS:
            k = self.xs_slots.setdefault(n.key, len(self.xs_slots))
            if k == len(self.prep):
                self.prep.append(f"        xs[{k}] = {self.exact(n)};")
            return f"xs[j][{k}]"
        k = self.pk_slots.setdefault(n.key, len(self.pk_slots))
        if k == len(self.pack):
            self.pack.append(f"        pk[{k}] = {self.exact(n)};")
        return f"pk[{k}]"

    def temp(self, code):
        k = self.count
        self.count += 1
        self.phases.append(("temp", k, code))
        return f"t{k}[j]"

    def factors(self, n):
        """(a, b) when the SP-level node is a product a*b that a parent sum can fuse."""
        if n.level != _LSP:
            return None
        if n.op == "mul":
            return self.emit(n.args[0]), self.emit(n.args[1])
        if n.op == "sq":
            a = self.emit(n.args[0])
            if n.args[0].level == _LSP and not a.startswith(("t", "r")):
                a = self.memo[n.args[0].key] = self.temp(a)
            return a, a
        if n.op == "div":
            num, den = n.args
            if num.op == "num" and num.val == 1.0:
                self.reciprocal(den)
                return None
            if num.level != _LSP:                 # setting-independent numerator: into the tree's root
                return self.quotient_factors(num, den)
            return self.emit(num), self.reciprocal(den)
        return None

    def quotient_factors(self, num, den):
        """num / den for a setting-independent numerator, as the two factors of the final
        multiply: (pair inverse with the numerator folded into the tree's root, sibling)."""
        if self.safe:
            return self.emit(num), self.reciprocal(den)
        key = ("quot", num.key, den.key)
        if key not in self.memo:
            k = self.count
            self.count += 1
            den_code, num_code = self.emit(den), self.emit(num)
            self.phases.append(("quot", k, den_code, num_code))
            self.memo[key] = (f"ip{k}[j / 2]", f"sibling_of<SPT>(den{k}, j)")
        return self.memo[key]

    def reciprocal(self, den):
        key = ("rcp", den.key)
        if key not in self.memo:
            k = self.count
            self.count += 1
            self.phases.append(("rcp", k, self.emit(den)))
            self.memo[key] = f"r{k}[j]"
        return self.memo[key]

    # ---- rendering: one particle per call, or two that share every inversion tree ----
    _LOOP = "        _Pragma(\"unroll\") for (int j = 0; j < SPT; ++j) "

    def render(self, outputs):
        """sweep_eval body: the phases, then v[j][c] = outputs[c]."""
        lines = []
        for ph in self.phases:
            if ph[0] == "temp":
                lines += [f"        double t{ph[1]}[SPT];", f"{self._LOOP}t{ph[1]}[j] = {ph[2]};"]
            elif ph[0] == "quot":
                k = ph[1]
                lines += [f"        double den{k}[SPT], ip{k}[(SPT + 1) / 2];", f"{self._LOOP}den{k}[j] = {ph[2]};",
                          f"        batch_div_poisoned<SPT>(den{k}, {ph[3]}, ip{k});"]
            else:
                k = ph[1]
                lines += [f"        double den{k}[SPT], r{k}[SPT];", f"{self._LOOP}den{k}[j] = {ph[2]};",
                          (f"{self._LOOP}r{k}[j] = guarded_rcp(den{k}[j]);" if self.safe
                           else f"        batch_rcp_poisoned<SPT>(den{k}, r{k});")]
        lines += [f"{self._LOOP}v[j][{c}] = {code};" for c, code in enumerate(outputs)]
        return "\n".join(lines)

    @staticmethod
    def _for_particle(code, x):
        """The code of one of the two particles of a pair: its own temporaries and packed values."""
        code = re.sub(r"\b(t|r|ip)(\d+)\[", lambda m: f"{m.group(1)}{m.group(2)}{x}[", code)
        code = re.sub(r"\bden(\d+)\b", lambda m: f"den{m.group(1)}{x}", code)
        return re.sub(r"\bpk\[", f"p{x}[", code)

    def render_pair(self, outputs):
        """sweep_eval_pair body: particles a and b side by side; every batched division runs ONE
        inversion tree over both particles' denominators."""
        fp, lines = self._for_particle, []
        for ph in self.phases:
            k = ph[1]
            for x in "ab":
                if ph[0] == "temp":
                    lines += [f"        double t{k}{x}[SPT];", f"{self._LOOP}t{k}{x}[j] = {fp(ph[2], x)};"]
                else:
                    lines += [f"        double den{k}{x}[SPT];", f"{self._LOOP}den{k}{x}[j] = {fp(ph[2], x)};"]
            if ph[0] == "quot":
                lines += [f"        double ip{k}a[SPT / 2], ip{k}b[SPT / 2];",
                          f"        batch_div_poisoned2<SPT>(den{k}a, den{k}b, {fp(ph[3], 'a')}, {fp(ph[3], 'b')}, "
                          f"ip{k}a, ip{k}b);"]
            elif ph[0] == "rcp":
                lines += [f"        double r{k}a[SPT], r{k}b[SPT];",
                          f"        batch_rcp_poisoned2<SPT>(den{k}a, den{k}b, r{k}a, r{k}b);"]
        for c, code in enumerate(outputs):
            lines += [f"{self._LOOP}va[j][{c}] = {fp(code, 'a')};", f"{self._LOOP}vb[j][{c}] = {fp(code, 'b')};"]
        return "\n".join(lines)

    def emit(self, n):
        if n.op == "num":
            return repr(n.val)
        if n.level != _LSP:
            return self.slot(n)
        if n.key in self.memo:
            return self.memo[n.key]
        a, b = (n.args + (None, None))[:2]
        if n.op in ("add", "sub"):
            fa, fb = self.factors(a), self.factors(b)
            if fb is not None:
                code = f"fma({'-' if n.op == 'sub' else ''}{fb[0]}, {fb[1]}, {self.emit(a)})"
            elif fa is not None:
                code = f"fma({fa[0]}, {fa[1]}, {'-' if n.op == 'sub' else ''}{self.emit(b)})"
            else:
                code = f"({self.emit(a)} {'+' if n.op == 'add' else '-'} {self.emit(b)})"
        elif n.op in ("mul", "sq"):
            f = self.factors(n)
            code = f"({f[0]} * {f[1]})"
        elif n.op == "div":
            f = self.factors(n)
            code = self.reciprocal(b) if f is None else f"({f[0]} * {f[1]})"
        elif n.op == "neg":
            code = f"(-{self.emit(a)})"
        elif n.op == "call":
            # inner-level calls of the fast form: range-checked polynomial / rsq versions
            # (obe_models.h, "elementary functions for the inner level"); the safe twin keeps ocml
            name = n.val if self.safe else _FAST_CALLS.get(n.val, n.val)
            code = f"{name}({', '.join(self.emit(c) for c in n.args)})"
        else:
            raise AssertionError(n.op)
        self.memo[n.key] = code
        return code


def _sweep_code(trees, settings, parameters, constants):
    """(prep_setting body, NXS, pack body, NPK, sweep_eval body, sweep_eval_safe body,
    sweep_eval_pair body) of the generated model."""
    build = _SweepBuilder(settings, parameters, constants)
    sw = _Node("sw", level=_LP)
    roots = [_mul(build.visit(t), sw) for t in trees]         # the kernel wants sqrt(w) * y
    fast = _SweepEmitter()
    safe = _SweepEmitter(safe=True, share=fast)
    fast_out = [fast.emit(r) for r in roots]
    safe_out = [safe.emit(r) for r in roots]
    nl = "\n"
    return (nl.join(fast.prep), max(1, len(fast.xs_slots)), nl.join(fast.pack), max(1, len(fast.pk_slots)),
            fast.render(fast_out), safe.render(safe_out), fast.render_pair(fast_out))


def _issue_slots(body):
    """Rough FP64 issue slots per evaluation of a generated sweep body: one per arithmetic operation, ~20 per
    elementary function, ~5 per element of a batched division."""
    import re
    ops = len(re.findall(r"fma\(| \* | - | \+ | / ", body))
    funcs = len(re.findall(r"\b(?:exp|log|log1p|expm1|pow|sin|cos|tan|fast_\w+|sqrt|hypot|atan2?|sinh|cosh|tanh)\(", body))
    return ops + 20 * funcs + 5 * body.count("batch_div")


def _check_names(names):
    seen = set()
    for n in names:
        if not n.isidentifier() or n in _FUNCS or n in _CONSTS or n in seen:
            raise ValueError(f"bad or duplicate name {n!r}")
        seen.add(n)


def translate(expressions, settings, parameters, constants):
    """(header text, numpy callable, content hash) for one or more channel expressions."""
    if isinstance(expressions, str):
        expressions = (expressions,)
    expressions, settings = tuple(expressions), tuple(settings)
    parameters, constants = tuple(parameters), tuple(constants)
    _check_names(settings + parameters + constants)
    names = set(settings + parameters + constants)
    trees = [ast.parse(e.strip(), mode="eval") for e in expressions]
    c_exprs = [_ToC(names).visit(t) for t in trees]
    prep_body, nxs, pack_body, npk, sweep_body, safe_body, pair_body = _sweep_code(trees, settings, parameters,
                                                                                     constants)
    ns, nc, npar, ncon = len(settings), len(expressions), len(parameters), len(constants)
    # evaluation time relative to the one-peak Lorentzian (~7 FP64 issue slots per evaluation): the sweep plans
    # its grid by time (csrc/obe_sweep.hip: plan_sweep)
    sweep_cost = min(16, max(1, round(_issue_slots(sweep_body) / 7)))
    decl = [f"        const double u_{n} = x_[{i}];" for i, n in enumerate(settings)]
    decl += [f"        const double u_{n} = th_[{i}];" for i, n in enumerate(parameters)]
    decl += [f"        const double u_{n} = c_[{i}];" for i, n in enumerate(constants)]
    body = [f"        y_[{c}] = {e};" for c, e in enumerate(c_exprs)]
    nl = "\n"
    header = f"""// generated by optbayesexpt_amd.models.from_expression — do not edit
// settings {settings}, parameters {parameters}, constants {constants}
// {(nl + '// ').join(expressions)}
#pragma once
namespace obe {{
struct PluginModel {{
    static constexpr int NS = {ns}, NC = {nc}, NREAD = {npar}, NCONST = {ncon}, NXS = {nxs}, NPK = {npk};
    static constexpr int kSweepCost = {sweep_cost};
    __device__ __forceinline__ static double sq(double v) {{ return v * v; }}
    __device__ __forceinline__ static void formula(const double* x_, const double* th_, const double* c_,
                                                   double* y_) {{
{nl.join(decl)}
        (void)x_; (void)th_; (void)c_;
{nl.join(body)}
    }}
    __device__ static void eval(const double* x, const ParamRef& th, const obe_model& m, double* y) {{
        double t[NREAD];
#pragma unroll
        for (int i = 0; i < NREAD; ++i) t[i] = th(i);
        formula(x, t, m.consts, y);
    }}
    // sweep form (see _exprmodel.py): per-setting terms -> xs, per-particle terms (times
    // sqrt(w) where it folds in) -> pk, the rest batched over the SPT settings of a lane
    __device__ static void prep_setting(const double* x, const obe_model& m, double* xs) {{
        (void)x; (void)m; (void)xs;
{prep_body}
    }}
    __device__ static void pack(const ParamRef& th, const double*, const obe_model& m, double sw, double* pk) {{
        (void)th; (void)m; (void)sw; (void)pk;
{pack_body}
    }}
    template <int SPT>
    __device__ __forceinline__ static void sweep_eval(const double (&xs)[SPT][NXS], const double* pk, double,
                                                      const obe_model&, double (&v)[SPT][NC]) {{
        (void)xs; (void)pk;
{sweep_body}
    }}
    // two particles at once: their denominators share every inversion tree
    static constexpr bool kHasPairEval = true;
    template <int SPT>
    __device__ __forceinline__ static void sweep_eval_pair(const double (&xs)[SPT][NXS], const double* pa,
                                                           const double* pb, double (&va)[SPT][NC],
                                                           double (&vb)[SPT][NC]) {{
        static_assert(SPT >= 2, "pair evaluation needs an even number of settings per lane");
        (void)xs; (void)pa; (void)pb;
{pair_body}
    }}
    // the same with one guarded IEEE reciprocal per element: used for the repeat after a sweep
    // in which a batch above left its exact range (NaN-poisoned; include/obe_hip.h OBE_SWEEP_SAFE)
    static constexpr bool kHasSafeEval = true;
    template <int SPT>
    __device__ __forceinline__ static void sweep_eval_safe(const double (&xs)[SPT][NXS], const double* pk, double,
                                                           const obe_model&, double (&v)[SPT][NC]) {{
        (void)xs; (void)pk;
{safe_body}
    }}
}};
}}  // namespace obe
"""
    codes = [compile(t, "<model expression>", "eval") for t in trees]
    env = {k: v[1] for k, v in _FUNCS.items()}
    env.update(_CONSTS)
    env["__builtins__"] = {}

    def numpy_form(sets, pars, cons):
        scope = dict(zip(settings, sets))
        scope.update(zip(parameters, pars))
        scope.update(zip(constants, cons))
        out = [eval(c, env, scope) for c in codes]             # noqa: S307 (restricted AST, no builtins)
        return out[0] if nc == 1 else np.array(np.broadcast_arrays(*out))

    digest = hashlib.sha1(header.encode()).hexdigest()[:16]
    return header, numpy_form, digest
