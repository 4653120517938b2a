"""The Python surface against what the real reference does with the same calls.

tests/golden/api_reference.json was produced by tests/golden/make_golden_api.py from the reference's own Cython classes
(430 probes: every ingest container / dtype, every constructor keyword and alias, every setter's good and bad values,
decode()'s length / dtype / zero-shortcut rules, the OSD and soft-information parameters).  Here the same probes run
against ldpc_amd: values, exception TYPES AND MESSAGES and warnings must be identical.  Probes that run the decoder on a
non-zero input need the device and are the `gpu` half; two inputs crash the reference process itself and are skipped.
"""
import json
import os

import pytest

import api_probes

HERE = os.path.dirname(os.path.abspath(__file__))
REF = {r["id"]: r for r in json.load(open(os.path.join(HERE, "golden", "api_reference.json")))["results"]}


def _namespace():
    from ldpc_amd.bp_decoder import BpDecoder, SoftInfoBpDecoder, io_test
    from ldpc_amd.bposd_decoder import BpOsdDecoder
    from ldpc_amd.helpers.scipy_helpers import convert_to_binary_sparse
    return {"BpDecoder": BpDecoder, "BpOsdDecoder": BpOsdDecoder, "SoftInfoBpDecoder": SoftInfoBpDecoder,
            "convert_to_binary_sparse": convert_to_binary_sparse, "io_test": io_test}


def _check(probe):
    want = REF[probe["id"]]
    if "crash" in want:
        pytest.skip("the reference process crashes on this input")
    got = json.loads(json.dumps(api_probes.run_probe(probe, _namespace())))
    assert got == want, f"\n mirror   : {json.dumps(got)[:1500]}\n reference: {json.dumps(want)[:1500]}"


def test_fixture_covers_every_probe():
    assert len(api_probes.PROBES) >= 400 and {p["id"] for p in api_probes.PROBES} == set(REF)
    assert len({p["id"] for p in api_probes.PROBES}) == len(api_probes.PROBES)


@pytest.mark.parametrize("probe", [p for p in api_probes.PROBES if not p["gpu"]], ids=lambda p: p["id"])
def test_surface_matches_reference(probe):
    _check(probe)


@pytest.mark.gpu
@pytest.mark.parametrize("probe", [p for p in api_probes.PROBES if p["gpu"]], ids=lambda p: p["id"])
def test_decoding_surface_matches_reference(probe):
    _check(probe)
