"""Repository rules that keep the parity claim honest (checked on CPU every round)."""
import ast
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _py_files(sub):
    for dp, _, fs in os.walk(os.path.join(ROOT, sub)):
        for f in fs:
            if f.endswith(".py"):
                yield os.path.join(dp, f)


def test_product_package_never_imports_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    for path in list(_py_files("ldpc_amd")) + list(_py_files("tools")):
        tree = ast.parse(open(path).read())
        for node in ast.walk(tree):
            names = []
            if isinstance(node, ast.Import):
                names = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                names = [node.module or ""]
            assert not any(n == "oracle" or n.startswith("oracle.") for n in names), f"{path} imports oracle"
    for path in (os.path.join(ROOT, "ldpc_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "ldpc_amd", "csrc"))):
        if path.endswith((".hip", ".h")):
            assert "oracle/" not in open(path).read().replace("oracle/bp_oracle.c: bp_oracle_set_math", ""), path


def test_nothing_that_runs_on_the_gpu_box_reads_the_reference_tree():
    """/root/reference does not exist on the GPU box: gpu tests, smoke() and bench.py must not open it.

    Path literals (string tokens that START with /root/reference) are only allowed as the argument of an
    os.path.isdir() presence check (build() uses one to decide whether oracle/_ref can be rebuilt)."""
    import io
    import tokenize
    for rel in ("bench.py", "__graft_entry__.py", "tests/test_gpu_parity.py", "tests/golden_util.py", "oracle/cpu_bench.py"):
        toks = list(tokenize.generate_tokens(io.StringIO(open(os.path.join(ROOT, rel)).read()).readline))
        for k, t in enumerate(toks):
            if t.type != tokenize.STRING:
                continue
            try:
                value = ast.literal_eval(t.string)
            except Exception:
                continue  # f-strings and the like
            if isinstance(value, str) and value.startswith("/root/reference"):
                before = "".join(x.string for x in toks[max(0, k - 4):k])
                assert before.endswith("isdir("), f"{rel}:{t.start[0]}: reference path used outside a presence check"


def test_no_reference_sources_in_the_repository():
    """oracle/_ref holds build outputs only and is git-ignored; no reference header is copied anywhere."""
    ignore = open(os.path.join(ROOT, ".gitignore")).read()
    assert "oracle/_ref/" in ignore
    for dp, dn, fs in os.walk(ROOT):
        if ".git" in dp or "gpurun_out" in dp:
            continue
        for f in fs:
            assert f not in ("bp.hpp", "gf2sparse.hpp", "sparse_matrix_base.hpp", "osd.hpp"), os.path.join(dp, f)


def test_required_top_level_files_exist():
    for rel in ("bench.py", "__graft_entry__.py", "include/ldpc_hip.h", "oracle/bp_oracle.c", "oracle/Makefile",
                "oracle/ref_harness.cpp", "tests/golden/make_golden.py", "DESIGN.md", "INTEGRATION.md"):
        assert os.path.exists(os.path.join(ROOT, rel)), rel
