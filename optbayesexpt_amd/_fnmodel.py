"""Device models from the *source* of a reference-style model function.

Every demo of the reference defines its model as a small Python function

    def my_model_function(sets, pars, cons):
        x, = sets
        x0, a, b = pars
        d, = cons
        return b + a / (((x - x0) / d) ** 2 + 1)

(demos/find_peak/sequentialLorentzian.py:53-75, demos/pipulse/pipulse.py:18-49, ...).  When
such a function is straight-line arithmetic — tuple unpacking of the three arguments, local
assignments, NumPy/math elementwise functions, one ``return`` — its source is translated,
statement by statement, into the expression language of ``models.from_expression`` (local
names are substituted symbolically; nothing of the function is executed during translation),
the NumPy form of the result is checked against the function itself on random inputs, bit for
bit, and the model then runs in the HIP kernels.  Anything else (branches, loops, complex
numbers, attribute access on other objects, global variables) is rejected with a ValueError
and the caller keeps the function as a host-callable model.
"""
import ast
import copy
import inspect
import textwrap
import types

import numpy as np

from ._exprmodel import _CONSTS, _FUNCS

_MODULES = {"np", "numpy", "math"}
_KINDS = ("s", "p", "c")          # generated names: s0.. settings, p0.. parameters, c0.. constants
_EXTRA_CALLS = {"square", "power", "absolute", "float64", "asarray", "array"}


def _function_def(fn):
    """The ast.FunctionDef of a plain Python function, from its source."""
    try:
        source = textwrap.dedent(inspect.getsource(fn))
    except (OSError, TypeError) as exc:
        raise ValueError(f"source of '{getattr(fn, '__name__', fn)}' is not available: {exc}")
    try:
        tree = ast.parse(source)
    except SyntaxError:
        raise ValueError("not a plain 'def' function (its source does not parse on its own)")
    fdef = next((n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef)), None)
    if fdef is None:
        raise ValueError("not a plain 'def' function")
    a = fdef.args
    if a.vararg or a.kwarg or a.kwonlyargs or a.posonlyargs or a.defaults:
        raise ValueError(f"'{fdef.name}': only plain positional arguments are supported")
    return fdef


class _Translator(ast.NodeTransformer):
    def __init__(self, arg_names, scope=None, owner=None, depth=0):
        self.group = dict(zip(arg_names, _KINDS))     # function argument -> kind
        self.count = {k: 0 for k in _KINDS} if owner is None else owner.count
        self.env = {}                                  # local name -> substituted expression
        self.scope = scope or {}                       # the function's globals: helper functions live there
        self.depth = depth

    def run_body(self, fdef):
        """Walk the statements of a function body; returns the list of returned expressions."""
        for stmt in fdef.body:
            if isinstance(stmt, ast.Expr) and isinstance(stmt.value, ast.Constant) and isinstance(stmt.value.value, str):
                continue                                             # docstring
            if isinstance(stmt, ast.Assign):
                if len(stmt.targets) != 1:
                    raise ValueError("chained assignment is not supported")
                self.assign(stmt.targets[0], stmt.value)
            elif isinstance(stmt, ast.Return):
                if stmt.value is None:
                    raise ValueError("the function returns nothing")
                return self.channels(stmt.value)
            else:
                raise ValueError(f"unsupported statement: {type(stmt).__name__}")
        raise ValueError("no return statement")

    def inline(self, helper, args):
        """A call of another plain Python function of the same module (the demos factor their
        formulas out that way): its body is translated with the parameters bound to the
        caller's argument expressions."""
        if self.depth >= 8:
            raise ValueError("helper functions nested too deeply (or recursive)")
        fdef = _function_def(helper)
        params = [a.arg for a in fdef.args.args]
        if len(params) != len(args):
            raise ValueError(f"'{fdef.name}' takes {len(params)} arguments, {len(args)} given")
        sub = _Translator((), scope=getattr(helper, "__globals__", {}), owner=self, depth=self.depth + 1)
        sub.env = dict(zip(params, args))
        out = sub.run_body(fdef)
        if len(out) != 1:
            raise ValueError(f"'{fdef.name}' returns several values")
        return out[0]

    def ref(self, kind, index):
        if index < 0:
            raise ValueError("negative indices into sets / pars / cons are not supported")
        self.count[kind] = max(self.count[kind], index + 1)
        return ast.Name(id=f"{kind}{index}", ctx=ast.Load())

    # --- expressions -------------------------------------------------------------------
    def visit_Name(self, node):
        if node.id in self.env:
            return copy.deepcopy(self.env[node.id])
        if node.id in self.group:
            raise ValueError(f"'{node.id}' is used as a whole; unpack it or index it with constants")
        if node.id in _FUNCS or node.id in _CONSTS:
            return node
        raise ValueError(f"name '{node.id}' is not a local of the function (globals are not supported)")

    def visit_Subscript(self, node):
        if isinstance(node.value, ast.Name) and node.value.id in self.group \
                and isinstance(node.slice, ast.Constant) and isinstance(node.slice.value, int):
            return self.ref(self.group[node.value.id], node.slice.value)
        raise ValueError("only constant indices into sets / pars / cons are supported")

    def visit_Attribute(self, node):
        if isinstance(node.value, ast.Name) and node.value.id in _MODULES and node.attr in _CONSTS:
            return ast.Name(id=node.attr, ctx=ast.Load())
        raise ValueError(f"unsupported attribute access '{ast.unparse(node)}'")

    def visit_Call(self, node):
        f = node.func
        if isinstance(f, ast.Attribute) and isinstance(f.value, ast.Name) and f.value.id in _MODULES:
            name = f.attr
        elif isinstance(f, ast.Name) and f.id not in self.env and f.id not in self.group:
            name = f.id
        else:
            raise ValueError(f"unsupported call '{ast.unparse(node)}'")
        if node.keywords:
            raise ValueError("keyword arguments are not supported")
        args = [self.visit(a) for a in node.args]
        if isinstance(f, ast.Name) and isinstance(self.scope.get(name), types.FunctionType):
            return self.inline(self.scope[name], args)
        if name == "square" and len(args) == 1:
            return ast.BinOp(left=args[0], op=ast.Pow(), right=ast.Constant(value=2))
        if name == "power" and len(args) == 2:
            return ast.BinOp(left=args[0], op=ast.Pow(), right=args[1])
        if name == "absolute":
            name = "abs"
        if name in ("float64",) and len(args) == 1:
            return args[0]
        if name not in _FUNCS:
            raise ValueError(f"function '{name}' is not available on the device")
        return ast.Call(func=ast.Name(id=name, ctx=ast.Load()), args=args, keywords=[])

    def visit_Constant(self, node):
        if isinstance(node.value, complex):
            raise ValueError("complex arithmetic is not supported")
        return node

    def generic_visit(self, node):
        if isinstance(node, (ast.BinOp, ast.UnaryOp, ast.operator, ast.unaryop, ast.expr_context, ast.Load)):
            return super().generic_visit(node)
        raise ValueError(f"unsupported syntax: {type(node).__name__}")

    # --- statements --------------------------------------------------------------------
    def assign(self, target, value):
        if isinstance(target, (ast.Tuple, ast.List)):
            if isinstance(value, ast.Name) and value.id in self.group:
                kind = self.group[value.id]
                for i, elt in enumerate(target.elts):
                    if not isinstance(elt, ast.Name):
                        raise ValueError("starred / nested unpacking is not supported")
                    self.env[elt.id] = self.ref(kind, i)
                return
            if isinstance(value, (ast.Tuple, ast.List)) and len(value.elts) == len(target.elts):
                values = [self.visit(v) for v in value.elts]
                for elt, v in zip(target.elts, values):
                    if not isinstance(elt, ast.Name):
                        raise ValueError("nested unpacking is not supported")
                    self.env[elt.id] = v
                return
            raise ValueError("unsupported unpacking")
        if not isinstance(target, ast.Name):
            raise ValueError("assignments to attributes or subscripts are not supported")
        self.env[target.id] = self.visit(value)

    def channels(self, value):
        """The returned expression(s): a tuple / list / np.array((...)) gives one per channel."""
        if isinstance(value, ast.Call) and value.args and not value.keywords:
            f = value.func
            name = f.attr if isinstance(f, ast.Attribute) else getattr(f, "id", None)
            if name in ("array", "asarray") and isinstance(value.args[0], (ast.Tuple, ast.List)):
                value = value.args[0]
        if isinstance(value, (ast.Tuple, ast.List)):
            return [self.visit(v) for v in value.elts]
        return [self.visit(value)]


def expressions_from_function(fn):
    """(expressions, settings, parameters, constants) — strings in the expression language and the
    generated argument names — from the source of ``fn(sets, pars, cons)``."""
    fdef = _function_def(fn)
    if len(fdef.args.args) != 3:
        raise ValueError("the model function must take exactly (sets, pars, cons)")
    tr = _Translator([x.arg for x in fdef.args.args], scope=getattr(fn, "__globals__", {}))
    result = tr.run_body(fdef)
    exprs = tuple(ast.unparse(ast.fix_missing_locations(ast.Expression(body=r))) for r in result)
    names = [tuple(f"{k}{i}" for i in range(tr.count[k])) for k in _KINDS]
    return exprs, names[0], names[1], names[2]


def check_against_function(fn, numpy_form, n_set, n_par, n_con, trials=3):
    """The translated formula must reproduce ``fn`` bit for bit in both broadcasting directions
    of the reference's calling convention (one setting x many parameters, many x one)."""
    g = np.random.default_rng(12345)
    for _ in range(trials):
        cons = tuple(g.uniform(0.5, 2.0, n_con))
        many_s = tuple(g.uniform(0.5, 3.0, 11) for _ in range(n_set))
        one_s = tuple(float(g.uniform(0.5, 3.0)) for _ in range(n_set))
        many_p = tuple(g.uniform(0.5, 3.0, 13) for _ in range(n_par))
        one_p = tuple(float(g.uniform(0.5, 3.0)) for _ in range(n_par))
        for sets, pars in ((many_s, one_p), (one_s, many_p)):
            with np.errstate(all="ignore"):
                want = np.asarray(fn(sets, pars, cons), dtype=np.float64)
                got = np.asarray(numpy_form(sets, pars, cons), dtype=np.float64)
            if want.shape != got.shape or not np.array_equal(want, got, equal_nan=True):
                raise ValueError("the translated formula does not reproduce the function bit for bit")
