"""The caller-loop fixtures come from the reference's OWN Python code: re-run the generators in
--check mode wherever the reference is present (the build container) and require the committed bytes.

tests/golden/make_golden_mcs.py imports the reference's ``MonteCarloBscSimulation`` (monte_carlo_simulation/mcs.py:10-171);
tests/golden/make_golden_window.py imports the reference's ``BaseOverlappingWindowDecoder`` / ``BpOsdOverlappingWindowDecoder``
(ckt_noise/base_overlapping_window_decoder.py:139-226, bposd_overlapping_window.py) behind a ``stim`` placeholder that raises
on any use; tests/golden/make_golden_sinter.py imports the reference's ``SinterBpOsdDecoder`` (sinter_decoders/sinter_bposd_decoder.py:57-130)
and ``detector_error_model_to_check_matrices`` (ckt_noise/dem_matrices.py:61-171) and hands them the model as data and b8 files.  On a machine without /root/reference (the GPU box) these tests skip: the fixtures are what travels."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/src_python/ldpc"), reason="the reference is not on this machine")


@pytest.mark.parametrize("script", ["make_golden_mcs.py", "make_golden_window.py", "make_golden_sinter.py"])
def test_generator_reproduces_the_committed_fixtures_from_the_reference_itself(script):
    pytest.importorskip("Cython")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", script), "--check"], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert "== committed fixture" in r.stdout and "DIFFERS" not in r.stdout


def test_stim_placeholder_refuses_every_use():
    """The guard itself: a token may sit in an annotation, nothing else."""
    code = ("import sys, types\n"
            f"src = open({os.path.join(ROOT, 'tests', 'golden', 'make_golden_window.py')!r}).read()\n"
            "head = src[src.index('class _StimToken'):src.index('assert \"stim\" not in sys.modules')]\n"
            "ns = {'types': types}; exec(head, ns)\n"
            "stim = ns['_StimPlaceholder']('stim')\n"
            "tok = stim.DetectorErrorModel\n"
            "def f(model: stim.DetectorErrorModel): return 1\n"
            "assert f(None) == 1\n"
            "for use in (lambda: tok(), lambda: tok.num_detectors, lambda: stim.Circuit('X 0'), lambda: list(tok), lambda: bool(tok), lambda: tok[0]):\n"
            "    try:\n        use()\n        raise SystemExit('not refused')\n    except RuntimeError as e:\n        assert 'NOT pinned' in str(e)\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-1500:]
