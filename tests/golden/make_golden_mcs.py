#!/usr/bin/env python3
"""Golden fail counts for MonteCarloBscSimulation, produced by the REFERENCE'S OWN class and decoders.

Build container only.  tests/golden/ref_python.py builds the reference's Python package in a scratch directory (SURVEY.md
Appendix A(3)); this script imports ``ldpc.monte_carlo_simulation.mcs.MonteCarloBscSimulation`` (mcs.py:10-171, the file
byte-identical to /root/reference's) and runs it around the reference's own ``BpDecoder`` / ``BpOsdDecoder``:

    python tests/golden/make_golden_mcs.py [--check]

Each fixture stores the recipe (code, p, seed, runs, decoder parameters) and the outcome (``fail_count`` as the reference's
``run()`` reports it, and per-run fail flags).  The class does not expose per-run flags: a recording proxy around the decoder
keeps every decoding it returned, and the errors are the same NumPy legacy-generator stream drawn again from the same seed
(mcs.py:96 seeds it; decode() draws nothing from it) -- their sum must equal the class's own count, which is asserted.
``--check`` regenerates in memory and compares with the committed files instead of writing.
``ldpc_amd.monte_carlo_simulation.MonteCarloBscSimulation`` must land on the same counts with the same seed.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import ref_python  # noqa: E402

ldpc = ref_python.use()
from ldpc.monte_carlo_simulation import mcs as ref_mcs  # noqa: E402  (the reference's module)
from ldpc.noise_models import bsc as ref_bsc  # noqa: E402

ref_python.assert_untouched(ref_mcs)
ref_python.assert_untouched(ref_bsc)

sys.path.insert(0, ROOT)
from ldpc_amd import codes  # noqa: E402  (our own code constructions: the recipe stored in the fixture)

OUT = HERE
CHECK = "--check" in sys.argv


class Recording:
    """The reference decoder, with every decode() result kept."""

    def __init__(self, inner):
        self.inner, self.out = inner, []

    def decode(self, syndrome):
        d = self.inner.decode(syndrome)
        self.out.append(np.array(d, dtype=np.uint8))
        return d


def run(name, h, recipe, *, error_rate, seed, runs, max_iter, bp_method, ms_scaling_factor=1.0, osd=False):
    h = sp.csr_matrix(h, dtype=np.uint8)
    kw = dict(error_rate=error_rate, max_iter=max_iter, bp_method=bp_method, ms_scaling_factor=ms_scaling_factor)
    dec = Recording(ldpc.BpOsdDecoder(h, osd_method="osd_0", **kw) if osd else ldpc.BpDecoder(h, **kw))
    sim = ref_mcs.MonteCarloBscSimulation(h, error_rate, dec, target_run_count=runs, tqdm_disable=True, seed=seed)
    result = sim.run()  # mcs.py:107-151
    assert result["run_count"] == runs and len(dec.out) == runs
    np.random.seed(seed)  # the same stream again (mcs.py:96, noise_models/bsc.py:23)
    fails = np.zeros(runs, np.uint8)
    for r in range(runs):
        error = ref_bsc.generate_bsc_error(h.shape[1], error_rate)
        fails[r] = not np.array_equal(dec.out[r], error)
    assert int(fails.sum()) == result["fail_count"], (int(fails.sum()), result["fail_count"])
    path = os.path.join(OUT, name + ".npz")
    if CHECK:
        g = np.load(path)
        same = int(g["fail_count"]) == result["fail_count"] and np.array_equal(np.unpackbits(g["fails"])[:runs], fails)
        print(f"{name:28s} runs={runs} fail_count={result['fail_count']} {'== committed fixture' if same else 'DIFFERS from the committed fixture'}")
        assert same
        return
    np.savez_compressed(path, name=name, recipe=recipe, error_rate=np.float64(error_rate), seed=np.int64(seed),
                        runs=np.int64(runs), max_iter=np.int32(max_iter), bp_method=bp_method,
                        ms_scaling_factor=np.float64(ms_scaling_factor), osd=np.bool_(osd),
                        fail_count=np.int64(result["fail_count"]), fails=np.packbits(fails),
                        logical_error_rate=np.float64(result["logical_error_rate"]),
                        logical_error_rate_eb=np.float64(result["logical_error_rate_eb"]),
                        generated_by="the reference's MonteCarloBscSimulation.run() (mcs.py:107-151) around the reference's own decoder")
    print(f"{name:28s} runs={runs} fail_count={result['fail_count']} {os.path.getsize(path) / 1024:.1f} KiB")


def main():
    run("mcs_ldpc96_ps", codes.regular_ldpc_code(96, 3, 6, seed=3), "regular_ldpc_code(96,3,6,seed=3)",
        error_rate=0.04, seed=42, runs=1500, max_iter=20, bp_method="product_sum")
    run("mcs_hamming5_ms", codes.hamming_code(5), "hamming_code(5)",
        error_rate=0.03, seed=7, runs=1000, max_iter=10, bp_method="minimum_sum", ms_scaling_factor=0.75)
    run("mcs_bb144_ps_osd0", codes.bivariate_bicycle_hx(), "bivariate_bicycle_hx()",
        error_rate=0.03, seed=11, runs=600, max_iter=30, bp_method="product_sum", osd=True)


if __name__ == "__main__":
    main()
