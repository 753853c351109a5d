"""The model functions of the reference's own demos and tests, taken from its source tree where
that is mounted (the build container; skipped elsewhere): which of them models.from_function
can put on the device.  Only the function definitions are extracted (nothing of the demo
scripts runs) into a temporary module; the translation is then checked against the function
itself, bit for bit."""
import ast
import glob
import importlib.util
import os
import textwrap

import pytest

REFERENCE = "/root/reference"


def model_function_sources():
    out = []
    for path in sorted(glob.glob(os.path.join(REFERENCE, "demos", "**", "*.py"), recursive=True)
                       + glob.glob(os.path.join(REFERENCE, "tests", "*.py"))):
        text = open(path).read()
        try:
            tree = ast.parse(text)
        except SyntaxError:
            continue
        # module-level helper functions a model function may call (only definitions are taken)
        candidates = [n for n in tree.body if isinstance(n, ast.FunctionDef) and not n.decorator_list
                      and len(n.args.args) != 3 and not n.args.defaults]
        for node in ast.walk(tree):
            if isinstance(node, ast.FunctionDef) and len(node.args.args) == 3 and not node.decorator_list \
                    and [a.arg for a in node.args.args] in (["sets", "pars", "cons"],
                                                            ["settings", "parameters", "constants"]):
                body = textwrap.dedent(ast.get_source_segment(text, node))
                helpers = "\n\n\n".join(ast.get_source_segment(text, h) for h in candidates if h.name + "(" in body)
                out.append((os.path.relpath(path, REFERENCE), node.lineno, node.name, helpers + "\n\n\n" + body))
    return out


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference tree is only mounted in the build container")
def test_reference_demo_model_functions_translate(tmp_path):
    from optbayesexpt_amd import _exprmodel, _fnmodel
    verdicts = {}
    for k, (rel, line, name, src) in enumerate(model_function_sources()):
        mod_path = tmp_path / f"extracted_{k}.py"
        mod_path.write_text("import numpy as np\n\n\n" + src + "\n")
        spec = importlib.util.spec_from_file_location(f"extracted_{k}", mod_path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)                       # defines the one function, runs nothing else
        fn = getattr(mod, name)
        try:
            exprs, s, p, c = _fnmodel.expressions_from_function(fn)
            _, numpy_form, _ = _exprmodel.translate(exprs, s, p, c)
            _fnmodel.check_against_function(fn, numpy_form, len(s), len(p), len(c))
            verdicts[f"{rel}:{line}"] = "device"
        except ValueError as exc:
            verdicts[f"{rel}:{line}"] = f"host ({exc})"
    on_device = {k for k, v in verdicts.items() if v == "device"}
    print("\n".join(f"{k}: {v}" for k, v in sorted(verdicts.items())))
    # every real-arithmetic demo model is translatable ...
    for demo in ("demos/find_peak/sequentialLorentzian.py", "demos/sweeper/sweeper.py", "demos/pipulse/pipulse.py",
                 "demos/line_plus_noise/line_plus_noise.py", "demos/fit_vs_obe/fit_vs_obe_makedata.py",
                 "demos/find_peak/seqLor_pdfevolve.py", "demos/server/server_script.py"):
        assert any(k.startswith(demo) for k in on_device), (demo, verdicts)
    # ... the complex-impedance coil model is not, and says why
    coil = [v for k, v in verdicts.items() if k.startswith("demos/lockin/lockin_of_coil.py")]
    assert coil and all(v.startswith("host") for v in coil)
