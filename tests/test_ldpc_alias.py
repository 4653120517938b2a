"""`ldpc_amd.install_as_ldpc()`: the name-level drop-in north_star asks for ("keeps the ldpc.BpDecoder ... API").

Runs in a subprocess (the alias changes sys.modules / sys.meta_path of the process that asks for it).  The reference's export
lists: src_python/ldpc/__init__.py:1-15, bp_decoder/__init__.py:1-7, bposd_decoder/__init__.py."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SCRIPT = r"""
import json, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
import ldpc_amd
root = ldpc_amd.install_as_ldpc()
assert ldpc_amd.install_as_ldpc() is root                     # idempotent
import ldpc
assert ldpc is root and ldpc.__version__.startswith("2.4.1")
from ldpc import BpDecoder, BpOsdDecoder, SoftInfoBpDecoder, SinterBpOsdDecoder
from ldpc.bp_decoder import BpDecoder as B2, SoftInfoBpDecoder as S2, io_test, BpDecoderBase, bp_decoder   # bp_decoder/__init__.py:1-7
from ldpc.bposd_decoder import BpOsdDecoder as O2, bposd_decoder
import ldpc.bp_decoder, ldpc.bposd_decoder, ldpc.codes, ldpc.noise_models, ldpc.monte_carlo_simulation, ldpc.sinter_decoders, ldpc.ckt_noise
from ldpc.codes import rep_code, hamming_code, ring_code
from ldpc.noise_models import generate_bsc_error
from ldpc.monte_carlo_simulation import MonteCarloBscSimulation
from ldpc.helpers.scipy_helpers import convert_to_binary_sparse
import ldpc_amd.bp_decoder, ldpc_amd.codes
assert B2 is BpDecoder is ldpc_amd.bp_decoder.BpDecoder and O2 is BpOsdDecoder and S2 is SoftInfoBpDecoder
assert sys.modules["ldpc.codes"] is ldpc_amd.codes and ldpc_amd.codes.__spec__.name == "ldpc_amd.codes"
assert ldpc.bp_decoder is bp_decoder and ldpc.bposd_decoder is bposd_decoder   # "Legacy syntax" rebinding, __init__.py:13-15
for missing in ("BpLsdDecoder", "BeliefFindDecoder", "UnionFindDecoder"):
    try:
        getattr(__import__("ldpc", fromlist=[missing]), missing)
        raise SystemExit(f"{{missing}} should not exist")
    except AttributeError:
        pass
try:
    import ldpc.bplsd_decoder
    raise SystemExit("ldpc.bplsd_decoder should not exist")
except ImportError:
    pass
# the API probes, through the alias
import api_probes
ns = {{"BpDecoder": BpDecoder, "BpOsdDecoder": BpOsdDecoder, "SoftInfoBpDecoder": SoftInfoBpDecoder,
      "convert_to_binary_sparse": convert_to_binary_sparse, "io_test": io_test}}
ref = {{r["id"]: r for r in json.load(open({fixture!r}))["results"]}}
bad, ran = [], 0
for probe in api_probes.PROBES:
    if probe["gpu"] or "crash" in ref[probe["id"]]:
        continue
    got = json.loads(json.dumps(api_probes.run_probe(probe, ns)))
    ran += 1
    if got != ref[probe["id"]]:
        bad.append(probe["id"])
print(json.dumps({{"ran": ran, "bad": bad}}))
"""


def test_import_ldpc_is_served_by_ldpc_amd_and_passes_the_api_probes():
    script = _SCRIPT.format(root=ROOT, tests=os.path.join(ROOT, "tests"), fixture=os.path.join(ROOT, "tests", "golden", "api_reference.json"))
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    got = json.loads(r.stdout.strip().splitlines()[-1])
    assert got["ran"] >= 400 and got["bad"] == []


def test_install_refuses_to_shadow_a_real_ldpc(tmp_path):
    (tmp_path / "ldpc").mkdir()
    (tmp_path / "ldpc" / "__init__.py").write_text("REAL = True\n")
    code = (f"import sys; sys.path.insert(0, {str(tmp_path)!r}); sys.path.insert(0, {ROOT!r})\n"
            "import ldpc_amd\n"
            "try:\n    ldpc_amd.install_as_ldpc()\n    raise SystemExit('not refused')\nexcept RuntimeError as e:\n    assert 'real' in str(e)\n"
            "import ldpc; assert ldpc.REAL\n"
            "root = ldpc_amd.install_as_ldpc(force=True)\n"
            "import ldpc as again; assert again is root and not hasattr(again, 'REAL')\n"
            "from ldpc import BpDecoder\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]


@pytest.mark.gpu
def test_reference_style_script_decodes_through_the_alias():
    """The reference's README usage, verbatim but for the install line, on the device."""
    code = (f"import sys; sys.path.insert(0, {ROOT!r})\n"
            "import ldpc_amd; ldpc_amd.install_as_ldpc()\n"
            "import numpy as np\n"
            "from ldpc.codes import rep_code\n"
            "from ldpc import BpDecoder, BpOsdDecoder\n"
            "H = rep_code(5)\n"
            "bpd = BpDecoder(H, error_rate=0.1, max_iter=5, bp_method='product_sum')\n"
            "s = np.array([1, 0, 0, 0], dtype=np.uint8)\n"
            "d = bpd.decode(s)\n"
            "assert bpd.converge and np.array_equal((H @ d) % 2, s), d\n"
            "osd = BpOsdDecoder(H, error_rate=0.1, max_iter=2, bp_method='ms', osd_method='osd_0')\n"
            "d2 = osd.decode(s)\n"
            "assert np.array_equal((H @ d2) % 2, s)\n"
            "print('ok')\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]
