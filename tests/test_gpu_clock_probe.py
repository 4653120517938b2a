"""ldpc_hip_bp_clock_probe: the shader clock the BP kernels ran at, from device-side cycle / tick counters (bench.py's issue fractions
use it instead of a clock copied from a profile taken on another box)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _engine(kind):
    from ldpc_amd import codes
    from ldpc_amd.engine import HipBpEngine
    if kind == "streamed":     # bp_decode_kernel (+ per-pass kernels)
        h, p, it, method, alpha, B = codes.regular_ldpc_code(2400, 3, 6, seed=5), 0.09, 20, 0, 1.0, 40000
    elif kind == "edge":       # bp_edge_kernel
        h, p, it, method, alpha, B = codes.rotated_surface_code_x(21), 0.05, 30, 1, 0.625, 65536
    elif kind == "edge8":      # bp_edge8_kernel
        h, p, it, method, alpha, B = codes.bivariate_bicycle_hx(), 0.05, 30, 1, 0.625, 65536
    elif kind == "wave_ps":    # bp_wave_ps_kernel
        h, p, it, method, alpha, B = codes.bivariate_bicycle_hx(), 0.05, 50, 0, 1.0, 8192
    else:                      # bp_wave_kernel (lane = node)
        h, p, it, method, alpha, B = codes.rotated_surface_code_x(21), 0.05, 30, 1, 0.625, 65536
    eng = HipBpEngine(h.indptr, h.indices, h.shape[1], np.full(h.shape[1], p), it, method, alpha)
    if kind == "streamed":
        eng.set_small_code_kernel(0)
    if kind == "wave":
        eng.set_small_code_kernel(4)
    return eng, p, B


@pytest.mark.parametrize("kind", ["streamed", "edge", "edge8", "wave_ps", "wave"])
def test_clock_probe_reads_a_plausible_shader_clock(kind):
    eng, p, B = _engine(kind)
    s = eng.gen_bsc_syndromes(7, p, shot0=0, shots=B, device="cuda:0")
    eng.decode_batch(s, want_llr=False)
    before = eng.clock_probe()
    for _ in range(3):
        eng.decode_batch(s, want_llr=False)
    after = eng.clock_probe()
    assert after[0] > before[0] and after[1] > before[1], "the BP kernel's workgroups must have added cycles and ticks"
    assert 50e6 <= after[2] <= 200e6, f"tick rate {after[2]} Hz (the constant-rate counter runs at 100 MHz on gfx9)"
    ghz = eng.clock_ghz(before, after)
    assert ghz is not None and 0.5 < ghz < 2.6, f"{kind}: {ghz} GHz is not a shader clock of an MI355X (max 2.4 GHz)"
    idle = eng.clock_probe()
    assert idle[:2] == after[:2], "nothing ran: the counters must not move"
    eng.close()


def test_copy_probe_reports_a_plausible_rate_and_checks_its_arguments():
    """ldpc_hip_bp_copy_probe (bench.py's box normaliser): a bare copy of message segments between the handle's two message arrays."""
    from ldpc_amd import codes
    from ldpc_amd._lib import LdpcHipError
    from ldpc_amd.engine import HipBpEngine
    h = codes.regular_ldpc_code(1200, 3, 6, seed=2)
    eng = HipBpEngine(h.indptr, h.indices, 1200, np.full(1200, 0.05), 10, 0, 1.0)
    ms, rate = eng.copy_probe(512, 30000, passes=2)      # 2 x 7.9 GB, as half of the headline's tiles
    assert ms > 0 and 1500.0 < rate < 8000.0, (ms, rate)  # (GB/s read + written; the spec's 8 TB/s is the ceiling)
    ms2, rate2 = eng.copy_probe(8)                         # segments per tile default to the handle's nnz; tiny: latency, any positive rate
    assert ms2 > 0 and rate2 > 0
    for bad in ((0, 100, 1), (4, -3, 1), (4, 100, 0), (4, 100, 65)):
        with pytest.raises((LdpcHipError, ValueError)):
            eng.copy_probe(*bad)
    # the decode after a probe is unaffected (the message arrays are scratch between decodes)
    s = eng.gen_bsc_syndromes(3, 0.05, shot0=0, shots=500, device="cuda:0")
    a = [x.cpu().numpy() for x in eng.decode_batch(s)]
    eng.copy_probe(16, 3600)
    b = [x.cpu().numpy() for x in eng.decode_batch(s)]
    assert all(np.array_equal(x.view(np.int64) if x.dtype == np.float64 else x, y.view(np.int64) if y.dtype == np.float64 else y) for x, y in zip(a, b))
    eng.close()
