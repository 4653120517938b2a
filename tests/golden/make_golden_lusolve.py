#!/usr/bin/env python3
"""Repack the reference's own solver fixture -- cpp_test/test_inputs/gf2_lu_solve_test.csv, 400 consistent GF(2) systems
`m;n;rows as column lists;right-hand side`, the data behind TestGF2RowReduce.cpp:327-368 (lu_solve) and :456-497
(fast_solve) -- into one compressed npz (DATA only: the systems; the reference asserts A x == y for each).
Build container only:  python tests/golden/make_golden_lusolve.py"""
import ast
import os

import numpy as np

SRC = "/root/reference/cpp_test/test_inputs/gf2_lu_solve_test.csv"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lusolve_reference_systems.npz")


def main():
    ms, ns, row_ptr, col_idx, rhs = [], [], [0], [], []
    for line in open(SRC):
        line = line.strip()
        if not line:
            continue
        m, n, rows, y = line.split(";")
        m, n = int(m), int(n)
        rows = ast.literal_eval(rows)
        assert len(rows) == m and len(y) == m
        for r in rows:
            col_idx.extend(sorted(r))
            row_ptr.append(len(col_idx))
        ms.append(m)
        ns.append(n)
        rhs.extend(int(c) for c in y)
    np.savez_compressed(OUT, m=np.array(ms, np.int32), n=np.array(ns, np.int32), row_ptr=np.array(row_ptr, np.int64),
                        col_idx=np.array(col_idx, np.int32), rhs=np.packbits(np.array(rhs, np.uint8)), rhs_bits=np.int64(len(rhs)))
    print(len(ms), "systems, largest", max(ms), "x", max(ns), os.path.getsize(OUT) // 1024, "KiB")


if __name__ == "__main__":
    main()
