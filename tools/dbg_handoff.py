import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from golden_util import load_case
from ldpc_amd.engine import HipBpEngine
which = sys.argv[1]
c = load_case("ldpc36_n600_ps50_p070")
h = c["h"]
eng = HipBpEngine(h.indptr, h.indices, c["n"], c["channel_probs"], c["max_iter"], 0, 1.0)
eng.set_small_code_kernel(0)
if which == "off":
    eng.set_handoff(0)
elif which == "all":
    eng.set_handoff(100000)
print("decoding", which, flush=True)
dec, llr, it, cv = eng.decode_batch(c["syndromes"])
print(which, "ok", np.array_equal(dec, c["decoding"]), np.array_equal(it, c["iterations"]), flush=True)
