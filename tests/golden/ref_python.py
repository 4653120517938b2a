"""Build container only: the REAL reference's Python package, built in a scratch directory and put on sys.path.

Recipe = SURVEY.md Appendix A(3): copy the reference's src_python / src_cpp / include to a scratch directory OUTSIDE the
repository, cythonize just bp_decoder and bposd_decoder, and replace the two package ``__init__.py`` files that import what
this image lacks (the root needs stim / sinter / pymatching / package metadata; ckt_noise's imports the LSD and PyMatching
window decoders).  Every module that a generator then imports and runs is the reference's own file, byte for byte
(``assert_untouched``).  Nothing of the reference enters the repository: generators store inputs and observed outputs.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys

REF = "/root/reference"

_SETUP = '''
import numpy as np
from setuptools import setup, Extension
from Cython.Build import cythonize
exts = [Extension(f"ldpc.{m}._{m}", [f"src_python/ldpc/{m}/_{m}.pyx"], include_dirs=[np.get_include(), "src_cpp", "include/robin_map"],
                  extra_compile_args=["-std=c++2a", "-O3"], language="c++") for m in ("bp_decoder", "bposd_decoder")]
setup(name="ldpc_probe", ext_modules=cythonize(exts, include_path=["src_python"], language_level=3), package_dir={"": "src_python"}, packages=[])
'''


def build(scratch: str = "/tmp/ldpc_ref_py") -> str:
    """Returns the directory to put on sys.path (``<scratch>/src_python``)."""
    pkg = os.path.join(scratch, "src_python", "ldpc")
    built = os.path.isdir(os.path.join(pkg, "bp_decoder")) and any(f.endswith(".so") for f in os.listdir(os.path.join(pkg, "bp_decoder"))) \
        and any(f.endswith(".so") for f in os.listdir(os.path.join(pkg, "bposd_decoder")))
    if not built:
        os.makedirs(scratch, exist_ok=True)
        for d in ("src_python", "src_cpp", "include"):
            shutil.rmtree(os.path.join(scratch, d), ignore_errors=True)
            shutil.copytree(os.path.join(REF, d), os.path.join(scratch, d))
        subprocess.run(["chmod", "-R", "u+w", scratch], check=True)
        open(os.path.join(scratch, "setup_probe.py"), "w").write(_SETUP)
        subprocess.run([sys.executable, "setup_probe.py", "build_ext", "--inplace"], cwd=scratch, check=True, capture_output=True)
    open(os.path.join(pkg, "__init__.py"), "w").write(
        "from ldpc.bp_decoder import BpDecoder, SoftInfoBpDecoder\nfrom ldpc.bposd_decoder import BpOsdDecoder\n")
    open(os.path.join(pkg, "ckt_noise", "__init__.py"), "w").write("")  # (the original imports the LSD / PyMatching window decoders)
    open(os.path.join(pkg, "sinter_decoders", "__init__.py"), "w").write("")  # (the original imports the belief-find / LSD sinter decoders)
    return os.path.join(scratch, "src_python")


def use(scratch: str = "/tmp/ldpc_ref_py"):
    """Build if needed, put the scratch package first on sys.path, return the imported reference ``ldpc``."""
    path = build(scratch)
    if path not in sys.path:
        sys.path.insert(0, path)
    for name in [k for k in sys.modules if k == "ldpc" or k.startswith("ldpc.")]:
        del sys.modules[name]
    import ldpc
    assert os.path.realpath(ldpc.__file__).startswith(os.path.realpath(path)), ldpc.__file__
    return ldpc


def assert_untouched(module) -> None:
    """The imported module's file is byte-identical to the reference's own file."""
    here = os.path.realpath(module.__file__)
    rel = here.split(os.sep + "src_python" + os.sep, 1)[1]
    there = os.path.join(REF, "src_python", rel)
    a, b = (hashlib.sha256(open(p, "rb").read()).hexdigest() for p in (here, there))
    assert a == b, f"{here} differs from {there}"
