"""MonteCarloBscSimulation (reference: monte_carlo_simulation/mcs.py): validation, batching logic against goldens captured
from the reference's per-run loop around the real reference decoder (tests/golden/make_golden_mcs.py)."""
import glob
import os

import numpy as np
import pytest
import scipy.sparse as sp

from ldpc_amd import codes
from ldpc_amd.monte_carlo_simulation import MonteCarloBscSimulation
from tests.golden_util import GOLDEN_DIR

MCS_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "mcs_*.npz")))


def _code(recipe):
    return eval("codes." + recipe, {"codes": codes})  # recipes are our own generator calls, written by make_golden_mcs.py


class _OracleDecoder:
    """decode_batch through the CPU oracle: exercises the simulation's host logic without a GPU."""

    def __init__(self, h, g):
        from oracle import BpOracle
        self.o = BpOracle(h, error_rate=float(g["error_rate"]), max_iter=int(g["max_iter"]), bp_method=str(g["bp_method"]),
                          ms_scaling_factor=float(g["ms_scaling_factor"]))
        self.osd = bool(g["osd"])

    def decode_batch(self, syndromes):
        dec = (self.o.bposd0_decode_batch(syndromes, want_llr=False) if self.osd else self.o.decode_batch(syndromes, want_llr=False))[0]
        dec[~syndromes.any(axis=1)] = 0
        return dec


def test_cases_present():
    assert len(MCS_CASES) >= 3


@pytest.mark.parametrize("case", MCS_CASES)
@pytest.mark.parametrize("batch_size", [64, 100000])
def test_fail_count_matches_reference_loop_cpu(case, batch_size):
    g = np.load(os.path.join(GOLDEN_DIR, case + ".npz"))
    h = sp.csr_matrix(_code(str(g["recipe"])))
    sim = MonteCarloBscSimulation(h, float(g["error_rate"]), _OracleDecoder(h, g), target_run_count=int(g["runs"]),
                                  tqdm_disable=True, seed=int(g["seed"]), batch_size=batch_size)
    out = sim.run()
    assert out["fail_count"] == int(g["fail_count"])
    assert out["run_count"] == int(g["runs"])
    assert out["logical_error_rate"] == int(g["fail_count"]) / int(g["runs"])
    assert set(out) == {"logical_error_rate", "logical_error_rate_eb", "error_rate", "run_count", "fail_count"}


def test_validation_messages():
    h = codes.hamming_code(3)
    dec = object()
    with pytest.raises(ValueError, match="parity_check_matrix should be of type"):
        MonteCarloBscSimulation([[1, 0]], 0.1, dec)
    with pytest.raises(ValueError, match="Invalid error rate"):
        MonteCarloBscSimulation(h, 1, dec)  # int, not float (mcs.py:62-69)
    with pytest.raises(ValueError, match="Invalid error rate"):
        MonteCarloBscSimulation(h, 1.5, dec)
    with pytest.raises(ValueError, match="Invalid Decoder"):
        MonteCarloBscSimulation(h, 0.1, None)
    with pytest.raises(ValueError, match="Invalid target run count"):
        MonteCarloBscSimulation(h, 0.1, dec, target_run_count=0)
    with pytest.raises(ValueError, match="tqdm_disable"):
        MonteCarloBscSimulation(h, 0.1, dec, tqdm_disable=1)
    with pytest.raises(ValueError, match="Invalid save interval"):
        MonteCarloBscSimulation(h, 0.1, dec, save_interval=0)
    with pytest.raises(ValueError, match="Invalid seed"):
        MonteCarloBscSimulation(h, 0.1, dec, seed=1.5)
    with pytest.raises(TypeError, match="decode_batch"):
        MonteCarloBscSimulation(h, 0.1, dec, tqdm_disable=True).run()


@pytest.mark.gpu
@pytest.mark.parametrize("case", MCS_CASES)
def test_fail_count_matches_reference_loop_gpu(case):
    from ldpc_amd.bp_decoder import BpDecoder
    from ldpc_amd.bposd_decoder import BpOsdDecoder
    g = np.load(os.path.join(GOLDEN_DIR, case + ".npz"))
    h = sp.csr_matrix(_code(str(g["recipe"])))
    kw = dict(error_rate=float(g["error_rate"]), max_iter=int(g["max_iter"]), bp_method=str(g["bp_method"]),
              ms_scaling_factor=float(g["ms_scaling_factor"]))
    dec = BpOsdDecoder(h, osd_method="osd_0", **kw) if bool(g["osd"]) else BpDecoder(h, **kw)
    sim = MonteCarloBscSimulation(h, float(g["error_rate"]), dec, target_run_count=int(g["runs"]), tqdm_disable=True,
                                  seed=int(g["seed"]), batch_size=512)
    assert sim.run()["fail_count"] == int(g["fail_count"])


@pytest.mark.gpu
def test_device_noise_runs_entirely_on_gpu():
    from ldpc_amd.bp_decoder import BpDecoder
    h = sp.csr_matrix(codes.regular_ldpc_code(96, 3, 6, seed=3))
    dec = BpDecoder(h, error_rate=0.04, max_iter=20, bp_method="product_sum")
    sim = MonteCarloBscSimulation(h, 0.04, dec, target_run_count=20000, tqdm_disable=True, seed=5, batch_size=8192,
                                  device_noise=True)
    out = sim.run()
    assert out["run_count"] == 20000
    assert 0.04 < out["logical_error_rate"] < 0.11  # reference loop at this point: 107 / 1500 = 0.071
    # the same shots decoded from host copies give the same count (device noise is the counter-based generator)
    from ldpc_amd.noise_models import generate_bsc_batch
    e = generate_bsc_batch(96, 0.04, 5, 0, 20000)
    s = np.ascontiguousarray((h @ e.T % 2).T.astype(np.uint8))
    d = dec.decode_batch(s)
    assert int((d != e).any(axis=1).sum()) == out["fail_count"]
