"""Static gate (CPU): no Python file of the repo may reference a name that nothing binds.

Round 3 lost its GPU evidence to `_case(7, T, batch)` in a test without a `batch` parameter: a NameError that only shows when
the test RUNS, i.e. on the GPU box.  This test finds that class of defect here, without running anything: every name a scope
resolves as a global (or reads at module level) must be bound at module level, be a builtin, or be declared `global` and assigned
somewhere in the file.  Built on `symtable` (the compiler's own scoping), so closures, comprehensions, class bodies and
`nonlocal` are resolved exactly as CPython resolves them.  Also: a parametrized test must take exactly the arguments its
`parametrize` names (the other half of the same mistake)."""
import ast
import builtins
import glob
import os
import symtable

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODULE_IMPLICIT = {"__name__", "__file__", "__doc__", "__package__", "__spec__", "__loader__", "__builtins__", "__path__",
                   "__debug__", "__annotations__", "__class__", "__qualname__", "__module__", "__dict__"}


def python_files():
    pats = ["*.py", "tests/*.py", "tests/golden/*.py", "ode-rl_amd/**/*.py", "oracle/**/*.py", "tools/**/*.py"]
    out = []
    for p in pats:
        out += glob.glob(os.path.join(ROOT, p), recursive=True)
    return sorted(set(f for f in out if "/build/" not in f and "/__pycache__/" not in f))


def _walk(table):
    yield table
    for child in table.get_children():
        yield from _walk(child)


def undefined_names(path, src=None):
    """[(name, line)] of names that are read somewhere in `path` and bound nowhere they could be found at run time."""
    if src is None:
        with open(path) as fh:
            src = fh.read()
    tree = ast.parse(src, path)
    if any(isinstance(n, ast.ImportFrom) and any(a.name == "*" for a in n.names) for n in ast.walk(tree)):
        return []                       # a star import binds names this scan cannot see
    top = symtable.symtable(src, path, "exec")
    bound = set(MODULE_IMPLICIT) | set(dir(builtins))
    for s in top.get_symbols():
        if s.is_assigned() or s.is_imported() or s.is_namespace():
            bound.add(s.get_name())
    for t in _walk(top):                # `global x` + an assignment in any function binds x at module level
        if t is top:
            continue
        for s in t.get_symbols():
            if s.is_declared_global() and s.is_assigned():
                bound.add(s.get_name())
    missing = set()
    for t in _walk(top):
        for s in t.get_symbols():
            if not s.is_referenced() or s.get_name() in bound:
                continue
            if t is top or s.is_global() or (t.get_type() == "class" and not s.is_local() and not s.is_free()):
                missing.add(s.get_name())
    if not missing:
        return []
    lines = []
    for n in ast.walk(tree):
        if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load) and n.id in missing:
            lines.append((n.id, n.lineno))
    return sorted(set(lines), key=lambda x: (x[1], x[0]))


def test_the_scanner_sees_the_round_3_defect():
    src = ("import pytest\n"
           "def _case(seed, T, batch):\n    return seed\n"
           "@pytest.mark.parametrize('method,T', [('rk4', 4)])\n"
           "def test_x(cuda, method, T):\n    return _case(7, T, batch)\n")
    assert undefined_names("<r3>", src) == [("batch", 6)]
    ok = src.replace("batch)\n", "3)\n")
    assert undefined_names("<r3-fixed>", ok) == []
    # closures, comprehensions, class bodies, global declarations and conditional imports are not false positives
    fine = ("import os\nclass A:\n    x = 1\n    y = [k for k in range(x)]\n    def m(self):\n        return os, __class__\n"
            "def f(a):\n    def g():\n        return a + h\n    return g\n"
            "def init():\n    global h\n    h = 2\n"
            "try:\n    import json\nexcept ImportError:\n    json = None\n"
            "z = [q for q in range(3)]\n")
    assert undefined_names("<fine>", fine) == []


@pytest.mark.parametrize("path", python_files(), ids=lambda p: os.path.relpath(p, ROOT))
def test_no_undefined_names(path):
    assert undefined_names(path) == []


def _parametrize_names(dec):
    if not (isinstance(dec, ast.Call) and isinstance(dec.func, ast.Attribute) and dec.func.attr == "parametrize" and dec.args):
        return None
    a = dec.args[0]
    if isinstance(a, ast.Constant) and isinstance(a.value, str):
        return [x.strip() for x in a.value.split(",") if x.strip()]
    if isinstance(a, (ast.Tuple, ast.List)) and all(isinstance(e, ast.Constant) for e in a.elts):
        return [e.value for e in a.elts]
    return None


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(ROOT, "tests", "test_*.py"))), ids=os.path.basename)
def test_parametrized_tests_take_and_use_their_arguments(path):
    """Every name a `parametrize` lists is a parameter of the test, every case has as many values as names, and the test body
    reads each of them (round 3's second defect: a `batch` column that the body ignored, so the batch-20 case ran at batch 3)."""
    with open(path) as fh:
        tree = ast.parse(fh.read(), path)
    for fn in ast.walk(tree):
        if not isinstance(fn, ast.FunctionDef):
            continue
        params = {a.arg for a in fn.args.args + fn.args.kwonlyargs}
        read = {n.id for n in ast.walk(fn) if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load)}
        for dec in fn.decorator_list:
            names = _parametrize_names(dec)
            if names is None:
                continue
            for n in names:
                assert n in params, f"{fn.name}: parametrize names {n!r}, which is not a parameter"
                assert n in read, f"{fn.name}: parametrized argument {n!r} is never read (line {fn.lineno})"
            cases = dec.args[1] if len(dec.args) > 1 else None
            if len(names) > 1 and isinstance(cases, (ast.List, ast.Tuple)):
                for c in cases.elts:
                    if isinstance(c, (ast.Tuple, ast.List)):
                        assert len(c.elts) == len(names), f"{fn.name}: case at line {c.lineno} has {len(c.elts)} values for {names}"
