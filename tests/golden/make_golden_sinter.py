#!/usr/bin/env python3
"""Golden fixtures for the sinter file decoder and the DEM -> matrices conversion, produced by the REFERENCE'S OWN modules.

Build container only:

    python tests/golden/make_golden_sinter.py [--check]

tests/golden/ref_python.py builds the reference's Python package in a scratch directory.  Two of
its modules are imported byte-identical to /root/reference's files (``assert_untouched``) and RUN, around the reference's
own ``BpOsdDecoder``:

* ``ldpc/sinter_decoders/sinter_bposd_decoder.py`` -- ``SinterBpOsdDecoder.decode_via_files`` (:57-126: model -> matrices ->
  ``BpOsdDecoder``; shot file in; the per-shot loop ``(observables_matrix @ decode(shot)) % 2`` :128-130; predictions out);
* ``ldpc/ckt_noise/dem_matrices.py`` -- ``detector_error_model_to_check_matrices`` (:61-171: hyperedge columns in order of
  first appearance, priors of repeated detector sets combined, last-occurrence observables, the ``^``-components as edges).

Both ``import stim`` (and ``sinter``), which this image lacks.  What they use of it, and what stands in for it HERE (in the
generator only -- nothing of this is in the product or travels):

* ``stim.DetectorErrorModel`` as a DATA object: ``from_file`` / ``flattened()`` / ``num_detectors`` / ``num_observables`` and, per
  instruction, ``type`` / ``args_copy()`` / ``targets_copy()`` with ``is_relative_detector_id()`` / ``is_logical_observable_id()``
  / ``is_separator()`` / ``val``.  The stand-in is a record of instructions this script BUILT programmatically (``Model`` below):
  no text is parsed on the reference side.  The same records are serialised to DEM text (the published file format:
  ``error(p) D0 D3 L1 ^ D4``), which is what the fixture stores and what ``ldpc_amd.ckt_noise.dem_text`` must read back.
  So the reference's conversion logic is pinned, and so is this repository's reading of FLAT model text; what stim itself does
  to a text with ``repeat`` blocks and ``shift_detectors`` (unrolling to absolute ids) is stim's, not the reference's, and
  stays checked only by this repository's own tests (tests/test_ckt_noise_host.py).
* ``stim.read_shot_data_file`` / ``write_shot_data_file`` with ``format="b8"``: the published shot format (bit i of a shot = bit
  i % 8 of byte i // 8, shots padded to whole bytes) restated with ``numpy.unpackbits`` / ``packbits``.  The fixture keeps the
  FILE BYTES on both sides.
* ``sinter.Decoder``: an empty base class.

Any other attribute of the stand-in modules raises, so a code path that needed more of stim than this would fail instead of
producing a fixture.  ``--check`` regenerates in memory and compares with the committed files instead of writing.
"""
from __future__ import annotations

import json
import os
import pathlib
import sys
import tempfile
import types

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import ref_python  # noqa: E402


# ----------------------------------------------------------------------------------------------------------------------
# the model as data, its text, and the object the reference walks
class Model:
    """A flat detector error model: ``errors`` = [(probability, [(detector ids, observable ids) per '^'-component])]."""

    def __init__(self, errors, num_detectors, num_observables):
        self.errors, self.num_detectors, self.num_observables = errors, num_detectors, num_observables

    def text(self) -> str:
        lines = []
        for p, parts in self.errors:
            toks = " ^ ".join(" ".join([f"D{d}" for d in dets] + [f"L{o}" for o in obs]) for dets, obs in parts)
            lines.append(f"error({p!r}) {toks}".rstrip())
        lines.append(f"detector D{self.num_detectors - 1}")            # pins num_detectors / num_observables even when the
        lines.append(f"logical_observable L{self.num_observables - 1}")  # last ids take part in no error
        return "\n".join(lines) + "\n"


class _Target:
    def __init__(self, kind, val=None):
        self._kind, self.val = kind, val

    def is_relative_detector_id(self):
        return self._kind == "D"

    def is_logical_observable_id(self):
        return self._kind == "L"

    def is_separator(self):
        return self._kind == "^"


class _Instruction:
    def __init__(self, type_, args=(), targets=()):
        self.type, self._args, self._targets = type_, list(args), list(targets)

    def args_copy(self):
        return list(self._args)

    def targets_copy(self):
        return list(self._targets)


class _DemObject:
    """What ``stim.DetectorErrorModel.from_file`` hands the reference: the instructions of a ``Model``, already flat."""

    def __init__(self, model: Model):
        self.num_detectors, self.num_observables = model.num_detectors, model.num_observables
        self._instructions = []
        for p, parts in model.errors:
            targets = []
            for k, (dets, obs) in enumerate(parts):
                if k:
                    targets.append(_Target("^"))
                targets += [_Target("D", d) for d in dets] + [_Target("L", o) for o in obs]
            self._instructions.append(_Instruction("error", [p], targets))
        self._instructions.append(_Instruction("detector", [], [_Target("D", model.num_detectors - 1)]))
        self._instructions.append(_Instruction("logical_observable", [], [_Target("L", model.num_observables - 1)]))

    def flattened(self):
        return list(self._instructions)


_MODELS_BY_PATH: dict = {}


class _DetectorErrorModelType:
    @staticmethod
    def from_file(path):
        return _DemObject(_MODELS_BY_PATH[str(path)])


def _read_shot_data_file(*, path, format, num_detectors=None, num_observables=None, **kw):
    assert format == "b8" and not kw
    bits = num_detectors if num_detectors is not None else num_observables
    raw = np.fromfile(str(path), dtype=np.uint8).reshape(-1, (bits + 7) // 8)
    return np.unpackbits(raw, axis=1, bitorder="little", count=bits).astype(bool)


def _write_shot_data_file(*, data, path, format, num_observables=None, num_detectors=None, **kw):
    assert format == "b8" and not kw
    bits = num_observables if num_observables is not None else num_detectors
    assert data.shape[1] == bits
    np.packbits(np.asarray(data, dtype=bool), axis=1, bitorder="little").tofile(str(path))


class _Refusing(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        raise RuntimeError(f"the code under test USED {self.__name__}.{name}: not in this image, NOT pinned by this run")


assert "stim" not in sys.modules and "sinter" not in sys.modules
_stim = _Refusing("stim")
_stim.DetectorErrorModel = _DetectorErrorModelType
_stim.DemTarget = _Target  # (an annotation at dem_matrices.py:127)
_stim.read_shot_data_file = _read_shot_data_file
_stim.write_shot_data_file = _write_shot_data_file
_sinter = _Refusing("sinter")
_sinter.Decoder = type("Decoder", (), {})
sys.modules["stim"], sys.modules["sinter"] = _stim, _sinter

ldpc = ref_python.use()
from ldpc.ckt_noise import dem_matrices as ref_dem  # noqa: E402  (the reference's modules)
from ldpc.sinter_decoders import sinter_bposd_decoder as ref_sinter  # noqa: E402

for mod in (ref_dem, ref_sinter):
    ref_python.assert_untouched(mod)

sys.path.insert(0, ROOT)
from ldpc_amd import codes  # noqa: E402  (our own code constructions)

OUT = HERE
CHECK = "--check" in sys.argv


# ----------------------------------------------------------------------------------------------------------------------
# models
def phenomenological(h, rounds, p_data, p_meas, logicals, *, hooks=0, duplicates=0, seed=0) -> Model:
    """``rounds`` rounds of a classical code: a data error per bit and round, a measurement error per check between rounds;
    ``logicals[k]`` = the bits on observable k.  ``hooks`` extra mechanisms are pairs of a round's data errors written as a
    decomposed hyperedge (``A ^ B``, a shared detector cancelling); ``duplicates`` repeat earlier detector sets with another
    probability and other observables (the reference combines the priors and keeps the LAST observables)."""
    rng = np.random.default_rng(seed)
    h = sp.csc_matrix(h)
    m, n = h.shape
    col = [list(map(int, h.indices[h.indptr[j]:h.indptr[j + 1]])) for j in range(n)]
    obs_of = [[k for k, sup in enumerate(logicals) if j in sup] for j in range(n)]
    pd = np.broadcast_to(np.asarray(p_data, np.float64), (n,))
    errors = []
    for t in range(rounds):
        for j in range(n):
            errors.append((float(pd[j]), [([t * m + i for i in col[j]], obs_of[j])]))
        if t < rounds - 1:
            for i in range(m):
                errors.append((float(p_meas), [([t * m + i, (t + 1) * m + i], [])]))
    for _ in range(hooks):
        t = int(rng.integers(rounds))
        a, b = (int(x) for x in rng.choice(n, 2, replace=False))
        errors.append((float(np.round(0.25 * pd[a], 6)), [([t * m + i for i in col[a]], obs_of[a]), ([t * m + i for i in col[b]], obs_of[b])]))
    for _ in range(duplicates):
        p, parts = errors[int(rng.integers(len(errors)))]
        dets = [d for part in parts for d in part[0]]
        other = sorted(set(int(x) for x in rng.choice(len(logicals), int(rng.integers(0, 2)) if len(logicals) > 1 else 0, replace=False)))
        errors.append((float(np.round(0.5 * p + 0.001, 6)), [(dets[::-1], other)]))
    return Model(errors, rounds * m, len(logicals))


def handmade() -> Model:
    """Every corner of dem_matrices.py:80-141 on purpose: a detector named twice in one mechanism (cancels), the same set reached
    by different components, an undecomposed hyperedge of five detectors, an error without detectors, observables named
    twice, a detector set seen three times, trailing ids that occur in no error."""
    e = [
        (0.01, [([0, 1], [0])]),
        (0.02, [([1, 2], [])]),
        (0.03, [([0, 1], []), ([1, 2], [1])]),            # = {0, 2}, observable 1; edges {0,1} and {1,2}
        (0.015, [([2, 0], [1, 1])]),                      # {0, 2} again: priors combine; L1 L1 cancels -> no observable (last wins)
        (0.04, [([3], [2])]),
        (0.05, [([3, 4, 5, 6, 7], [0, 2])]),              # undecomposed hyperedge (allowed by the sinter decoder)
        (0.011, [([4, 5], []), ([6, 7], [0]), ([3], [2])]),  # the same five detectors, decomposed: same column, edges recorded at first sight only
        (0.02, [([5, 6], [])]),
        (0.007, [([8], [])]),
        (0.009, [([8, 9], [3])]),
        (0.013, [([9, 9, 8], [])]),                       # {8} within ONE component: the set() of the component drops the repeat
        (0.02, [([], [3])]),                              # no detector at all: a column of weight 0
        (0.006, [([0, 1], [0, 1])]),                      # third sight of {0, 1}
        (0.03, [([6, 7], [])]),
        (0.025, [([4], [])]),
        (0.01, [([7, 10], [])]),
        (0.02, [([10, 2], [1])]),
    ]
    return Model(e, 12, 5)


def logical_supports(n, k, weight, seed):
    rng = np.random.default_rng(seed)
    return [set(int(x) for x in rng.choice(n, weight, replace=False)) for _ in range(k)]


# ----------------------------------------------------------------------------------------------------------------------
def csc_fields(prefix, a):
    a = sp.csc_matrix(a)
    a.sort_indices()
    return {prefix + "_indptr": a.indptr.astype(np.int64), prefix + "_indices": a.indices.astype(np.int64),
            prefix + "_shape": np.array(a.shape, np.int64)}


def run(name, model: Model, configs, *, shots, seed, scale=1.0):
    text = model.text()
    with tempfile.TemporaryDirectory() as tmp:
        tmp = pathlib.Path(tmp)
        dem_path = tmp / "model.dem"
        dem_path.write_text(text)
        _MODELS_BY_PATH[str(dem_path)] = model
        mats = ref_dem.detector_error_model_to_check_matrices(_DemObject(model), allow_undecomposed_hyperedges=True)
        rng = np.random.default_rng(seed)
        e = (rng.random((shots, mats.priors.size)) < np.minimum(mats.priors * scale, 0.5)[None, :]).astype(np.uint8)
        dets = np.ascontiguousarray((sp.csr_matrix(mats.check_matrix) @ e.T % 2).T.astype(np.uint8))
        dets[0] = 0  # the all-zero shot takes BpOsdDecoder.decode's shortcut
        dets_path = tmp / "dets.b8"
        np.packbits(dets, axis=1, bitorder="little").tofile(str(dets_path))
        out = dict(name=name, dem_text=text, num_shots=shots, num_dets=model.num_detectors, num_obs=model.num_observables,
                   dets_b8=np.fromfile(str(dets_path), dtype=np.uint8), priors=mats.priors, n_configs=len(configs))
        for key in ("check_matrix", "observables_matrix", "edge_check_matrix", "edge_observables_matrix", "hyperedge_to_edge_matrix"):
            out.update(csc_fields(key, getattr(mats, key)))
        for k, cfg in enumerate(configs):
            dec = ref_sinter.SinterBpOsdDecoder(**cfg)
            obs_path = tmp / f"obs_{k}.b8"
            dec.decode_via_files(num_shots=shots, num_dets=model.num_detectors, num_obs=model.num_observables, dem_path=dem_path,
                                 dets_b8_in_path=dets_path, obs_predictions_b8_out_path=obs_path, tmp_dir=tmp)
            assert np.array_equal(dec.matrices.priors, mats.priors)
            out[f"config_{k}"] = json.dumps(cfg, sort_keys=True)
            out[f"obs_b8_{k}"] = np.fromfile(str(obs_path), dtype=np.uint8)
            # the per-shot entry (:128-130) agrees with the file loop
            probe = shots // 3
            one = np.asarray(dec.decode(dets[probe])).astype(np.uint8).ravel()
            row = np.unpackbits(out[f"obs_b8_{k}"].reshape(shots, -1)[probe], bitorder="little", count=model.num_observables)
            assert np.array_equal(one, row)
    path = os.path.join(OUT, name + ".npz")
    if CHECK:
        g = np.load(path)
        bad = [k for k in out if k != "name" and not np.array_equal(g[k], out[k])]
        print(f"{name}: {'== committed fixture' if not bad else 'DIFFERS from the committed fixture in ' + str(bad)}")
        assert not bad
        return
    np.savez_compressed(path, generated_by="the reference's SinterBpOsdDecoder.decode_via_files (sinter_bposd_decoder.py:57-130) and "
                                           "detector_error_model_to_check_matrices (dem_matrices.py:61-171) around its own BpOsdDecoder", **out)
    flips = [int(np.unpackbits(out[f'obs_b8_{k}']).sum()) for k in range(len(configs))]
    print(f"{name}: {shots} shots, {model.num_detectors} detectors x {mats.priors.size} columns ({len(model.errors)} mechanisms), "
          f"{model.num_observables} observables; predicted observable flips per config {flips}")


def main():
    sweep = [dict(max_iter=8, osd_method="osd0"),
             dict(max_iter=8, osd_method="osd_cs", osd_order=6),
             dict(max_iter=8, osd_method="osd_e", osd_order=5)]
    ring = codes.ring_code(8)
    run("sinter_ring8_r6", phenomenological(ring, 6, 0.04, 0.03, [{0}], hooks=6, duplicates=5, seed=1), sweep + [
        dict(max_iter=0, bp_method="ps", osd_method="osd_cs", osd_order=3),                     # max_iter 0 -> the number of columns
        dict(max_iter=6, bp_method="ms", ms_scaling_factor=0.8, schedule="serial", osd_method="osd_e", osd_order=4)], shots=203, seed=11, scale=1.5)
    ham = codes.hamming_code(3)
    run("sinter_hamming3_r7", phenomenological(ham, 7, np.linspace(0.01, 0.05, 7), 0.02, [{0, 1, 2}, {2, 4, 6}, {3}], hooks=4, duplicates=6, seed=2),
        sweep + [dict(max_iter=5, bp_method="product_sum", osd_method="osd_e", osd_order=7)], shots=257, seed=12, scale=2.0)
    surf = codes.rotated_surface_code_x(5)
    run("sinter_surface5_r4", phenomenological(surf, 4, 0.02, 0.02, [set(range(0, 25, 5)), set(range(5))], hooks=10, duplicates=8, seed=3),
        [dict(max_iter=10, osd_method="osd0"), dict(max_iter=10, osd_method="osd_cs", osd_order=8), dict(max_iter=10, osd_method="osd_e", osd_order=6),
         dict(max_iter=4, bp_method="ps", schedule="serial", osd_method="osd_cs", osd_order=5)], shots=300, seed=13, scale=1.5)
    bb = codes.bivariate_bicycle_hx()
    run("sinter_bb144_r3", phenomenological(bb, 3, 0.006, 0.006, logical_supports(144, 12, 12, 5), hooks=12, duplicates=10, seed=4),
        [dict(max_iter=12, osd_method="osd0"), dict(max_iter=12, osd_method="osd_cs", osd_order=10), dict(max_iter=12, osd_method="osd_e", osd_order=8)],
        shots=192, seed=14, scale=2.0)
    run("sinter_handmade", handmade(), sweep + [dict(max_iter=3, bp_method="ps", osd_method="osd_e", osd_order=3)], shots=130, seed=15, scale=3.0)


if __name__ == "__main__":
    main()
