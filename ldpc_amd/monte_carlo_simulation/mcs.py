"""Batch-aware mirror of the reference's ``MonteCarloBscSimulation`` (monte_carlo_simulation/mcs.py:10-171).

The reference draws one error, forms one syndrome and calls ``Decoder.decode`` per run (mcs.py:124-149).  Here the
runs of a chunk are drawn together, their syndromes are formed together and the whole chunk goes through
``Decoder.decode_batch`` -- one launch of the HIP belief-propagation path instead of ``chunk`` Python round trips.

Same constructor, validation, attributes and ``save()`` dictionary.  With a ``seed`` the errors are the ones the
reference's loop draws: NumPy's legacy ``binomial(1, p, n)`` called R times consumes the global stream exactly as one
``binomial(1, p, (R, n))`` call does, so ``fail_count`` is reproducible against the reference run for run
(``tests/golden/mcs_*.npz``).  ``device_noise=True`` instead draws the errors on the GPU from the counter-based
generator (``ldpc_hip_gen_bsc_syndromes``): nothing but the fail count crosses PCIe, at the price of a different
(equally distributed) sample.
"""
from __future__ import annotations

import datetime
import time
from typing import Dict, Union

import numpy as np
import scipy.sparse as sp


class MonteCarloBscSimulation:
    def __init__(self, parity_check_matrix: Union[np.ndarray, sp.csr_matrix] = None, error_rate: float = None,
                 Decoder=None, target_run_count=1000, tqdm_disable=False, save_interval=60, seed=None, run=False,
                 batch_size: int = 65536, device_noise: bool = False) -> None:
        # validation in the reference's order and wording (mcs.py:53-97)
        if parity_check_matrix is None or not isinstance(parity_check_matrix, (np.ndarray, sp.csr_matrix)):
            raise ValueError(
                f"parity_check_matrix should be of type np.ndarray or scipy.sparse.csr_matrix. Not {type(parity_check_matrix)}")
        self.parity_check_matrix = parity_check_matrix
        if error_rate is None or not isinstance(error_rate, float) or error_rate < 0 or error_rate > 1:
            raise ValueError("Invalid error rate provided. The error rate should be a float with value between 0 and 1.")
        self.error_rate = error_rate
        if Decoder is None:
            raise ValueError("Invalid Decoder object provided.")
        self.Decoder = Decoder
        if not isinstance(target_run_count, int) or target_run_count <= 0:
            raise ValueError("Invalid target run count provided.")
        self.target_run_count = target_run_count
        if not isinstance(tqdm_disable, bool):
            raise ValueError("Invalid value for tqdm_disable flag.")
        self.tqdm_disable = tqdm_disable
        if not isinstance(save_interval, int) or save_interval <= 0:
            raise ValueError("Invalid save interval provided.")
        self.save_interval = save_interval
        if seed is None:
            self.seed = None
        else:
            if not isinstance(seed, int):
                raise ValueError("Invalid seed provided. Please provide a postive integer")
            self.seed = seed
            np.random.seed(self.seed)
        if not isinstance(batch_size, int) or batch_size <= 0:
            raise ValueError("Invalid batch size provided.")
        self.batch_size = batch_size
        self.device_noise = bool(device_noise)

        self.run_count = 0
        self.fail_count = 0
        self.logical_error_rate = 0.0
        self.logical_error_rate_eb = 0.0
        if run:
            self.run()

    # one chunk of runs: errors -> syndromes -> decode_batch -> number of decoding failures
    def _chunk_failures(self, shot0: int, shots: int) -> int:
        n = self.parity_check_matrix.shape[1]
        if self.device_noise:
            return self._chunk_failures_device(shot0, shots)
        errors = np.random.binomial(1, self.error_rate, (shots, n)).astype(np.uint8)  # == `shots` calls of generate_bsc_error
        H = self.parity_check_matrix
        if sp.issparse(H):
            syndromes = np.asarray((H @ errors.T).T % 2, dtype=np.uint8)
        else:
            syndromes = (errors.astype(np.int64) @ np.asarray(H, dtype=np.int64).T % 2).astype(np.uint8)
        decodings = self.Decoder.decode_batch(np.ascontiguousarray(syndromes))
        return int(np.count_nonzero((np.asarray(decodings) != errors).any(axis=1)))

    def _chunk_failures_device(self, shot0: int, shots: int) -> int:
        import torch

        eng = self.Decoder._get_engine()  # the decoder's HipBpEngine: errors, syndromes and decodings stay in HBM
        seed = 0 if self.seed is None else self.seed
        device = torch.device("cuda", eng.device if eng.device >= 0 else torch.cuda.current_device())
        syndromes, errors = eng.gen_bsc_syndromes(seed, self.error_rate, shot0, shots, device=device, want_errors=True)
        decodings = self.Decoder.decode_batch(syndromes)
        return int(torch.count_nonzero((decodings != errors).any(dim=1).bool()).item())

    def run(self) -> Dict:
        """Runs ``run_count + 1 .. target_run_count`` (mcs.py:104-150), ``batch_size`` runs per launch."""
        self.start_date = datetime.datetime.fromtimestamp(time.time()).strftime("%A, %B %d, %Y %H:%M:%S")
        if not hasattr(self.Decoder, "decode_batch"):
            raise TypeError("The Decoder must provide decode_batch (ldpc_amd.bp_decoder.BpDecoder / ldpc_amd.bposd_decoder.BpOsdDecoder).")
        try:
            from tqdm import tqdm
        except ImportError:  # pragma: no cover
            tqdm = None
        first = self.run_count + 1
        total = self.target_run_count - self.run_count
        pbar = None if (tqdm is None or self.tqdm_disable) else tqdm(total=max(total, 0), ncols=0)
        self.fail_count = 0
        done = 0
        while done < total:
            shots = min(self.batch_size, total - done)
            self.fail_count += self._chunk_failures(first - 1 + done, shots)
            done += shots
            self.run_count = first - 1 + done
            self.logical_error_rate = self.fail_count / self.run_count
            self.logical_error_rate_eb = np.sqrt(self.logical_error_rate * (1 - self.logical_error_rate) / self.run_count)
            if pbar is not None:
                pbar.update(shots)
                pbar.set_description(
                    f"Physical error rate: {100*self.error_rate:.2f}%; Logical error rate: {100*self.logical_error_rate:.2f}+-{100*self.logical_error_rate_eb:.2f}%")
        if pbar is not None:
            pbar.close()
        return self.save()

    def save(self):
        return {"logical_error_rate": self.logical_error_rate, "logical_error_rate_eb": self.logical_error_rate_eb,
                "error_rate": self.error_rate, "run_count": self.run_count, "fail_count": self.fail_count}
