"""The oracle is test infrastructure: nothing in the product (lp_mp_amd/, include/, tools/) may import, link or load it;
bench.py may only use it inside cpu_baseline(), __graft_entry__.py only in build() (compiling the checker) and
smoke() (checking against it)."""
import ast
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _py_files(d):
    for base, _, files in os.walk(os.path.join(ROOT, d)):
        for f in files:
            if f.endswith(".py"):
                yield os.path.join(base, f)


def _imports_oracle(path):
    tree = ast.parse(open(path).read())
    hits = []
    for node in ast.walk(tree):
        if isinstance(node, ast.Import) and any(a.name.split(".")[0] == "oracle" for a in node.names):
            hits.append(node.lineno)
        if isinstance(node, ast.ImportFrom) and (node.module or "").split(".")[0] == "oracle":
            hits.append(node.lineno)
    return hits


def test_product_never_touches_the_oracle():
    for d in ("lp_mp_amd", "tools"):
        for f in _py_files(d):
            assert not _imports_oracle(f), f
    for base, _, files in os.walk(os.path.join(ROOT, "lp_mp_amd")):
        for f in files:
            if f.endswith((".cpp", ".hpp", ".hip", ".hxx", ".h")):
                src = open(os.path.join(base, f)).read()
                assert not re.search(r'#include\s*[<"][^>"]*oracle', src), f
                assert "liblpmp_oracle" not in src, f


def test_bench_uses_the_oracle_only_in_the_cpu_baseline_leg():
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef,)):
            uses = any(isinstance(n, ast.ImportFrom) and (n.module or "").startswith("oracle") for n in ast.walk(node))
            assert uses == (node.name == "cpu_baseline"), node.name
        else:
            assert not any(isinstance(n, (ast.Import, ast.ImportFrom)) and "oracle" in ast.dump(n) for n in ast.walk(node))
    tree = ast.parse(open(os.path.join(ROOT, "__graft_entry__.py")).read())
    for node in tree.body:
        if isinstance(node, ast.FunctionDef):
            uses = any(isinstance(n, ast.ImportFrom) and (n.module or "").startswith("oracle") for n in ast.walk(node))
            assert uses == (node.name in ("build", "smoke")), node.name


def test_product_library_is_not_an_experiment_build():
    """tools/build_variant.sh builds the same sources with tuning knobs and LPMP_ABLATE_* switches (the latter compute wrong
    duals).  kernels.hip refuses the switches without LPMP_EXPERIMENT_BUILD, lp_mp_amd/build.py never sets it, and the library
    says which it is"""
    from lp_mp_amd import build as B, engine as E
    assert not any("ABLATE" in f or "EXPERIMENT" in f for f in B.FLAGS)
    assert E.lib().lpmp_experiment_build() == 0
    src = open(os.path.join(ROOT, "lp_mp_amd", "csrc", "kernels.hip")).read()
    used = set(re.findall(r"LPMP_ABLATE_[A-Z_0-9]+", src))
    guard = src[: src.index("namespace lpmp {")]
    assert used and all(m in guard for m in used), used - set(re.findall(r"LPMP_ABLATE_[A-Z_0-9]+", guard))
    assert "#error" in guard and "LPMP_EXPERIMENT_BUILD" in guard
