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
    def __init__(self, names, fast_division=False):
        self.names = names
        self.fast_division = fast_division

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
            if self.fast_division:
                return f"({a} * guarded_rcp({b}))"      # sweep form: ~10 issue slots instead of ~28
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
    c_fast = [_ToC(names, fast_division=True).visit(t) for t in trees]
    ns, nc, npar, ncon = len(settings), len(expressions), len(parameters), len(constants)
    decl = [f"        const double u_{n} = x_[{i}];" for i, n in enumerate(settings)]
    decl += [f"        const double u_{n} = th_[{i}];" for i, n in enumerate(parameters)]
    decl += [f"        const double u_{n} = c_[{i}];" for i, n in enumerate(constants)]
    body = [f"        y_[{c}] = {e};" for c, e in enumerate(c_exprs)]
    body_fast = [f"        y_[{c}] = {e};" for c, e in enumerate(c_fast)]
    nl = "\n"
    header = f"""// generated by optbayesexpt_amd.models.from_expression — do not edit
// settings {settings}, parameters {parameters}, constants {constants}
// {(nl + '// ').join(expressions)}
#pragma once
namespace obe {{
struct PluginModel {{
    static constexpr int NS = {ns}, NC = {nc}, NREAD = {npar}, NCONST = {ncon}, NXS = {ns}, NPK = {npar};
    __device__ __forceinline__ static double sq(double v) {{ return v * v; }}
    __device__ __forceinline__ static void formula(const double* x_, const double* th_, const double* c_,
                                                   double* y_) {{
{nl.join(decl)}
        (void)x_; (void)th_; (void)c_;
{nl.join(body)}
    }}
    // 1/b for the flop-bound sweep: v_rcp_f64 + one cubic correction (~1 ulp); falls back to the
    // raw reciprocal for b = 0 / inf / NaN so that a/0 = inf and a/inf = 0 as in IEEE division
    __device__ __forceinline__ static double guarded_rcp(double b) {{
        const double r0 = __builtin_amdgcn_rcp(b);
        const double e = fma(-b, r0, 1.0);
        const double r = fma(r0, fma(e, e, e), r0);
        return r == r ? r : r0;
    }}
    // the same formula with divisions as multiplications by guarded_rcp (sweep only; the
    // Bayes update and eval_over_* use the exact IEEE form above)
    __device__ __forceinline__ static void formula_fast(const double* x_, const double* th_, const double* c_,
                                                        double* y_) {{
{nl.join(decl)}
        (void)x_; (void)th_; (void)c_;
{nl.join(body_fast)}
    }}
    __device__ static void eval(const double* x, const ParamRef& th, const obe_model& m, double* y) {{
        double t[NREAD];
#pragma unroll
        for (int i = 0; i < NREAD; ++i) t[i] = th(i);
        formula(x, t, m.consts, y);
    }}
    __device__ static void prep_setting(const double* x, const obe_model&, double* xs) {{
#pragma unroll
        for (int k = 0; k < NS; ++k) xs[k] = x[k];
    }}
    __device__ static void pack(const ParamRef& th, const double*, const obe_model&, double, double* pk) {{
#pragma unroll
        for (int i = 0; i < NREAD; ++i) pk[i] = th(i);
    }}
    template <int SPT>
    __device__ __forceinline__ static void sweep_eval(const double (&xs)[SPT][NXS], const double* pk, double sw,
                                                      const obe_model& m, double (&v)[SPT][NC]) {{
#pragma unroll
        for (int j = 0; j < SPT; ++j) {{
            double y[NC];
            formula_fast(xs[j], pk, m.consts, y);
#pragma unroll
            for (int c = 0; c < NC; ++c) v[j][c] = sw * y[c];
        }}
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
