"""Where bp_relative_lds_kernel's wavefronts spend their cycles (LDPC_HIP_REL_PROF=1: the kernel adds up shader cycles per phase, the library
prints the shares to stderr).  GPU box:  python tools/serial_relative_phases.py [surface|bb|ldpc600 ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldpc_amd import codes  # noqa: E402
from ldpc_amd.engine import HipBpEngine  # noqa: E402

CASES = {
    "surface": lambda: (codes.rotated_surface_code_x(21), 0.05, 30, 1, 0.625),
    "bb": lambda: (codes.bivariate_bicycle_hx(), 0.05, 50, 0, 1.0),
    "bbms": lambda: (codes.bivariate_bicycle_hx(), 0.05, 50, 1, 0.625),
}
for which in (sys.argv[1:] or ["surface", "bb"]):
    h, p, it, meth, alpha = CASES[which]()
    for sw in ((), (("REL_LEVELS", 0),), (("REL_LDS", 16),)):
        eng = HipBpEngine(h.indptr, h.indices, h.shape[1], np.full(h.shape[1], p), it, meth, alpha)
        eng.set_schedule("serial_relative")
        eng.set_debug_switch("REL_PROF", 1)
        for k, v in sw:
            eng.set_debug_switch(k, v)
        s = eng.gen_bsc_syndromes(7, p, shot0=0, shots=65536, device="cuda:0")
        out = eng.decode_batch(s)
        print(which, dict(sw), "kernel ms %.2f" % eng.last_kernel_ms(), "mean iterations %.2f" % float(out[2].float().mean()), "converged %.3f" % float(out[3].float().mean()), flush=True)
