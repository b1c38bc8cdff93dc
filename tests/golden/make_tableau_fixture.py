"""Writes tests/golden/tableaux_reference.json: the Butcher tableaux the reference holds as literals.

The five tableau classes of the reference (src/timesteppers/hdg_imex.py:668-1038) are the only numeric data it holds for the
hot path.  The module cannot be imported (it starts with `from firedrake import *`, hdg_imex.py:7; Firedrake is not installed),
so this script reads the file as TEXT, walks its syntax tree and evaluates the bodies of the six properties `nstages, _a_expl,
_a_impl, _b_expl, _b_impl, _c_expl` of every `IncompressibleEulerHDGIMEX*` class -- pure numpy arithmetic on literals -- with
`np` as the only name in scope.  Nothing of the reference is imported or copied; the output is data (numbers, the class name
and label, the line range they came from).  Runs in the build container only (the reference does not travel to the GPU box):

    python tests/golden/make_tableau_fixture.py [/root/reference]

Values are stored twice: as decimal repr (exact round trip for IEEE doubles) and as float.hex() for a bit-for-bit check."""
import ast
import json
import os
import sys

import numpy as np

PROPS = ("nstages", "_a_expl", "_a_impl", "_b_expl", "_b_impl", "_c_expl")


def _evaluate(func):
    """value returned by a property body (docstring dropped), evaluated with numpy only"""
    body = [st for st in func.body if not (isinstance(st, ast.Expr) and isinstance(getattr(st, "value", None), ast.Constant))]
    fn = ast.FunctionDef(name="_f", args=ast.arguments(posonlyargs=[], args=[], kwonlyargs=[], kw_defaults=[], defaults=[]),
                         body=body, decorator_list=[], lineno=func.lineno, col_offset=0)
    mod = ast.fix_missing_locations(ast.Module(body=[fn], type_ignores=[]))
    scope = {"np": np, "__builtins__": {}}
    exec(compile(mod, "<tableau>", "exec"), scope)
    return scope["_f"]()


def _label(cls):
    for node in ast.walk(cls):
        if isinstance(node, ast.keyword) and node.arg == "label" and isinstance(node.value, ast.Constant):
            return node.value.value
    return None


def extract(path):
    with open(path) as f:
        tree = ast.parse(f.read())
    out = {}
    for cls in tree.body:
        if not (isinstance(cls, ast.ClassDef) and cls.name.startswith("IncompressibleEulerHDGIMEX") and cls.name != "IncompressibleEulerHDGIMEX"):
            continue
        entry = {"label": _label(cls), "lines": [cls.lineno, cls.end_lineno]}
        for fn in cls.body:
            if isinstance(fn, ast.FunctionDef) and fn.name in PROPS:
                v = _evaluate(fn)
                if fn.name == "nstages":
                    entry["nstages"] = int(v)
                else:
                    a = np.asarray(v, dtype=float)
                    entry[fn.name.lstrip("_")] = {"shape": list(a.shape), "values": [repr(float(x)) for x in a.ravel()],
                                                  "hex": [float(x).hex() for x in a.ravel()]}
        missing = [p for p in PROPS if p.lstrip("_") not in entry]
        if missing:
            raise SystemExit(f"{cls.name}: properties not found: {missing}")
        out[cls.name] = entry
    return out


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    src = os.path.join(ref, "src", "timesteppers", "hdg_imex.py")
    data = {"source": "src/timesteppers/hdg_imex.py (eikehmueller/IncompressibleEulerHDG), property bodies evaluated from the syntax tree",
            "classes": extract(src)}
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tableaux_reference.json")
    with open(dst, "w") as f:
        json.dump(data, f, indent=1)
        f.write("\n")
    print(f"{dst}: {len(data['classes'])} classes")


if __name__ == "__main__":
    main()
