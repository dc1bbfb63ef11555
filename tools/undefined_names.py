"""Poor man's pyflakes (none is installed here and the GPU paths cannot run in the build container): names a
module's functions load that are bound nowhere -- module scope, builtins, any enclosing function, comprehension
or class body.  Usage: python tools/undefined_names.py file.py [...]"""

import ast
import builtins
import sys


def bound_names(node):
    names = set()
    for n in ast.walk(node):
        if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            names.add(n.name)
        if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.Lambda)):
            a = n.args
            for arg in a.posonlyargs + a.args + a.kwonlyargs + ([a.vararg] if a.vararg else []) + ([a.kwarg] if a.kwarg else []):
                names.add(arg.arg)
        elif isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
            names.add(n.id)
        elif isinstance(n, (ast.Import, ast.ImportFrom)):
            for alias in n.names:
                names.add((alias.asname or alias.name).split(".")[0])
        elif isinstance(n, ast.ExceptHandler) and n.name:
            names.add(n.name)
        elif isinstance(n, (ast.Global, ast.Nonlocal)):
            names.update(n.names)
        elif isinstance(n, ast.MatchAs) and n.name:
            names.add(n.name)
        elif isinstance(n, ast.MatchStar) and n.name:
            names.add(n.name)
    return names


def check(path):
    tree = ast.parse(open(path).read(), path)
    known = bound_names(tree) | set(dir(builtins)) | {"__file__", "__name__", "__doc__"}
    bad = []
    for n in ast.walk(tree):
        if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load) and n.id not in known:
            bad.append((n.lineno, n.id))
    return bad


if __name__ == "__main__":
    failed = False
    for path in sys.argv[1:]:
        for lineno, name in check(path):
            print(f"{path}:{lineno}: undefined name {name!r}")
            failed = True
    sys.exit(1 if failed else 0)
