#!/usr/bin/env python3
"""Golden fixtures on the quantum-code matrices the reference's own tests ship (python_test/pcms/*.npz: hypergraph
product [[400,16,6]], toric d=20, planar surface d=20; used by python_test/test_qcodes.py:94-543, which only prints
logical error rates) with the decoder settings of that test file (:110-186): min-sum 0.625 / product-sum, max_iter 5,
parallel / serial schedule, OSD_0 / OSD_CS 3 / OSD_E 3 -- through the REAL reference (oracle/_ref/libref_bp.so).
Stored per fixture: hx and lx (CSR), the BSC errors (counter PRNG), decodings, BP flags and the logical-failure flags
lx (decoding + error) != 0 that test_qcodes counts.  Build container only:

    make -C oracle ref && python tests/golden/make_golden_qcodes.py
"""
from __future__ import annotations

import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from oracle import RefBpOsd, csr_arrays  # noqa: E402
from ldpc_amd.noise_models import generate_bsc_batch  # noqa: E402
from make_golden import h_crc  # noqa: E402

PCMS = "/root/reference/python_test/pcms"
OUT = os.path.dirname(os.path.abspath(__file__))
SETTINGS = [  # label, bp_method, alpha, schedule, osd_method id, osd_order   (test_qcodes.py:110-186)
    ("ms_par_osd0", "minimum_sum", 0.625, "parallel", 1, 0),
    ("ms_par_cs3", "minimum_sum", 0.625, "parallel", 3, 3),
    ("ms_par_e3", "minimum_sum", 0.625, "parallel", 2, 3),
    ("ms_ser_osd0", "minimum_sum", 0.625, "serial", 1, 0),
    ("ps_par_osd0", "product_sum", 1.0, "parallel", 1, 0),
]


def main():
    for code, p, shots in (("400_16_6", 0.03, 160), ("toric_20", 0.04, 128), ("surface_20", 0.04, 128)):
        hx = sp.csr_matrix(sp.load_npz(f"{PCMS}/hx_{code}.npz"), dtype=np.uint8)
        lx = sp.csr_matrix(sp.load_npz(f"{PCMS}/lx_{code}.npz"), dtype=np.uint8)
        hx.sort_indices()
        lx.sort_indices()
        m, n, rp, ci = csr_arrays(hx)
        err = generate_bsc_batch(n, p, 17, 0, shots)
        synd = np.ascontiguousarray((hx @ err.T % 2).T.astype(np.uint8))
        for label, method, alpha, schedule, osd_method, osd_order in SETTINGS:
            ref = RefBpOsd(hx, error_rate=p, max_iter=5, bp_method=method, ms_scaling_factor=alpha, osd_method=osd_method,
                           osd_order=osd_order, schedule=schedule)
            dec, llr, it, conv = ref.decode_batch(synd)
            resid = dec ^ err
            logical_fail = np.asarray((lx @ resid.T % 2).T, dtype=np.uint8).any(axis=1)
            name = f"qcodes_{code}_{label}"
            path = os.path.join(OUT, name + ".npz")
            np.savez_compressed(path, name=name, m=m, n=n, h_crc=np.uint32(h_crc(hx)), row_ptr=rp, col_idx=ci,
                                lx_row_ptr=lx.indptr.astype(np.int32), lx_col_idx=lx.indices.astype(np.int32), k=np.int32(lx.shape[0]),
                                error_rate=np.float64(p), max_iter=np.int32(5), bp_method=method, ms_scaling_factor=np.float64(alpha),
                                schedule=schedule, osd_method=np.int32(osd_method), osd_order=np.int32(osd_order),
                                errors=np.packbits(err, axis=1), decoding=np.packbits(dec, axis=1), converge=conv, iterations=it,
                                llr_rowsum=np.sum(np.where(np.abs(llr) < 1e100, llr, 0.0), axis=1), logical_fail=logical_fail)
            print(f"{name:34s} shots={shots} bp conv={conv.mean():.3f} logical failures={int(logical_fail.sum()):3d} "
                  f"{os.path.getsize(path) / 1024:6.1f} KiB")


if __name__ == "__main__":
    main()
