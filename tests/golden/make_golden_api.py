#!/usr/bin/env python3
"""Generate tests/golden/api_reference.json: what the REAL reference's Python classes do with every probe of tests/api_probes.py.

Runs only in the build container (needs /root/reference and Cython).  Recipe = SURVEY.md Appendix A(3): copy the reference's
src_python / src_cpp / include into a scratch directory, cythonize just bp_decoder and bposd_decoder, replace ldpc/__init__.py
by one that imports only those (the original needs stim / sinter / pymatching), import, probe.  Nothing of the reference is
copied into the repository: the JSON holds inputs' names and the observed outputs (values, exception types and messages).

    python tests/golden/make_golden_api.py [scratch_dir]
"""
import json
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
REF = "/root/reference"
scratch = sys.argv[1] if len(sys.argv) > 1 else "/tmp/ldpc_ref_py"

sys.path.insert(0, HERE)
import ref_python  # noqa: E402  (the scratch build of the reference: tests/golden/ref_python.py)
ref_python.use(scratch)
sys.path.insert(0, TESTS)
import ldpc  # noqa: E402  (the reference)
from ldpc.bp_decoder import io_test  # noqa: E402
from ldpc.helpers.scipy_helpers import convert_to_binary_sparse  # noqa: E402
import api_probes  # noqa: E402

ns = {"BpDecoder": ldpc.BpDecoder, "BpOsdDecoder": ldpc.BpOsdDecoder, "SoftInfoBpDecoder": ldpc.SoftInfoBpDecoder,
      "convert_to_binary_sparse": convert_to_binary_sparse, "io_test": io_test}


def isolated(probe):
    """Run one probe in a forked child: some inputs crash the reference outright (segmentation fault in its C++), which is
    recorded as such -- the mirror is not asked to reproduce a crash."""
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:
        os.close(r)
        try:
            payload = json.dumps(api_probes.run_probe(probe, ns)).encode()
        except BaseException as exc:  # noqa: BLE001
            payload = json.dumps({"id": probe["id"], "harness_error": repr(exc)}).encode()
        with os.fdopen(w, "wb") as f:
            f.write(payload)
        os._exit(0)
    os.close(w)
    with os.fdopen(r, "rb") as f:
        data = f.read()
    _, status = os.waitpid(pid, 0)
    if os.WIFSIGNALED(status) or not data:
        return {"id": probe["id"], "crash": f"the reference process died with signal {os.WTERMSIG(status) if os.WIFSIGNALED(status) else '?'}"}
    return json.loads(data)


results = [isolated(p) for p in api_probes.PROBES]
out = {"reference": "quantumgizmos/ldpc (src_python/ldpc/bp_decoder/_bp_decoder.pyx, bposd_decoder/_bposd_decoder.pyx, helpers/scipy_helpers.py), "
                    "built here by this script", "probes": len(results), "results": results}
path = os.path.join(HERE, "api_reference.json")
json.dump(out, open(path, "w"), indent=0, sort_keys=True)
print(f"{len(results)} probes -> {path} ({os.path.getsize(path)} bytes); crashed: {[r['id'] for r in results if 'crash' in r]}")
