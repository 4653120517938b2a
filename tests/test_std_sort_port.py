"""The oracle's restatement of libstdc++'s std::sort (serial_relative re-sorts its bit order with it every iteration,
bp.hpp:470-483; the sort is unstable, so with tied keys the exact swap sequence is observable) against the host's real std::sort."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_port_equals_std_sort(oracle_built):
    src = os.path.join(HERE, "native", "std_sort_check.cpp")
    so = os.path.join(HERE, "native", "libstd_sort_check.so")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(oracle_built.ORACLE_SO)):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, src, "-L" + os.path.join(ROOT, "oracle"), "-lbp_oracle",
                        "-Wl,-rpath," + os.path.join(ROOT, "oracle")], check=True)
    lib = C.CDLL(so)
    lib.std_sort_port_mismatches.restype = C.c_long
    lib.std_sort_port_mismatches.argtypes = [C.c_long, C.c_uint]
    assert lib.std_sort_port_mismatches(6, 12345) == 0
