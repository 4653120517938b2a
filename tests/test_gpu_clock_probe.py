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
