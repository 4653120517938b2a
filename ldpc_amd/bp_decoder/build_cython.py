"""Build ``ldpc_amd/bp_decoder/_bp_core*.so`` (Cython binding of include/ldpc_hip.hpp) in place.

    python ldpc_amd/bp_decoder/build_cython.py
Needs Cython + a C++ compiler and ldpc_amd/lib/libldpc_hip.so (make -C ldpc_amd/csrc); no GPU.
"""
import os
import sys


def build():
    from setuptools import Extension, setup
    from Cython.Build import cythonize
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(os.path.dirname(here))
    ext = Extension(
        "ldpc_amd.bp_decoder._bp_core", [os.path.join(here, "_bp_core.pyx")], language="c++",
        include_dirs=[os.path.join(root, "include")], library_dirs=[os.path.join(root, "ldpc_amd", "lib")],
        libraries=["ldpc_hip"], extra_compile_args=["-std=c++17", "-O2"],
        extra_link_args=["-Wl,-rpath,$ORIGIN/../lib", "-Wl,-rpath-link,/opt/rocm/lib"])
    cwd = os.getcwd()
    os.chdir(root)
    try:
        setup(name="ldpc_amd_bp_core", ext_modules=cythonize([ext], quiet=True, build_dir=os.path.join(root, "build", "cython")),
              script_args=["build_ext", "--inplace", "--build-temp", os.path.join(root, "build", "tmp")])
    finally:
        os.chdir(cwd)


if __name__ == "__main__":
    sys.exit(build())
