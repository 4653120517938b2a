"""Host buffers through ldpc_hip_bp_decode_batch: large batches move in pinned double-buffered chunks that overlap the kernels
(csrc/host_decode_abi.h: decode_batch_pipelined).  Chunking must change nothing: every row, log-ratio bits included, equals the
one-shot staged path and the device-resident decode."""
import numpy as np
import pytest

from golden_util import bits_equal

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("method,alpha,schedule", [(0, 1.0, "parallel"), (1, 0.625, "parallel"), (1, 0.9, "serial")])
def test_pipelined_host_path_equals_the_one_shot_path(method, alpha, schedule, oracle_built):
    import torch
    from ldpc_amd import codes
    from ldpc_amd.engine import HipBpEngine
    h = codes.regular_ldpc_code(2400, 3, 6, seed=8)
    n, p, B = 2400, 0.055, 20000 + 37  # (a ragged last chunk and a ragged last tile)
    eng = HipBpEngine(h.indptr, h.indices, n, np.full(n, p), 12, method, alpha)
    eng.set_small_code_kernel(0)
    if schedule == "serial":
        eng.set_schedule(0, None)
    s_dev = eng.gen_bsc_syndromes(3, p, shot0=0, shots=B, device="cuda:0")
    s = s_dev.cpu().numpy()
    s[5] = 0            # an all-zero row
    s[B - 1, 0] = 2     # a byte above 1: never converges
    want = [t.cpu().numpy() if t is not None else None for t in eng.decode_batch(torch.from_numpy(s).cuda(), want_llr=True)]
    eng.set_debug_switch("NO_HOST_PIPELINE", 1)
    one = eng.decode_batch(s, want_llr=True)
    eng.set_debug_switch("NO_HOST_PIPELINE", -1)
    for rows in (1024, 4096, -1):
        eng.set_debug_switch("HOST_CHUNK_ROWS", rows)
        for want_llr in (True, False):
            got = eng.decode_batch(s, want_llr=want_llr)
            tag = f"chunk rows {rows} llr {want_llr}"
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[0], one[0]), tag
            assert np.array_equal(got[2], want[2]) and np.array_equal(got[3].astype(bool), want[3].astype(bool)), tag
            if want_llr:
                assert bits_equal(got[1], want[1]) and bits_equal(got[1], one[1]), tag
            else:
                assert got[1] is None
    # against the CPU checker on a sample (the pipelined path is what BpDecoder.decode_batch(numpy) runs)
    rows = np.r_[0:24, B - 24:B]
    o = oracle_built.BpOracle(h, error_rate=p, max_iter=12, bp_method=method, ms_scaling_factor=alpha)
    ref = o.decode_serial_batch(s[rows]) if schedule == "serial" else o.decode_batch(s[rows])
    assert np.array_equal(got[0][rows], ref[0]) and np.array_equal(got[2][rows], ref[2])
    eng.close()


def test_bpdecoder_numpy_batch_is_the_pipelined_path_and_keeps_the_shortcut(oracle_built):
    """`BpDecoder.decode_batch` with a NumPy array of another dtype: results in that dtype, the all-zero rows take the reference's
    shortcut (pyx:679-681), the rest equal the device-resident decode."""
    import torch
    from ldpc_amd import codes
    from ldpc_amd.bp_decoder import BpDecoder
    h = codes.regular_ldpc_code(3000, 3, 6, seed=2)
    dec = BpDecoder(h, error_rate=0.04, max_iter=10, bp_method="product_sum", input_vector_type="syndrome")
    eng = dec._get_engine()
    B = 50000
    s = eng.gen_bsc_syndromes(9, 0.04, shot0=0, shots=B, device="cuda:0").cpu().numpy().astype(np.int64)
    s[[3, 777, B - 1]] = 0
    out = dec.decode_batch(s, want_log_prob_ratios=False)
    assert out.dtype == np.int64 and out.shape == (B, 3000)
    assert not out[[3, 777, B - 1]].any() and dec.converge_batch[[3, 777, B - 1]].all() and not dec.iter_batch[[3, 777, B - 1]].any()
    ref = dec.decode_batch(torch.from_numpy(s.astype(np.uint8)).cuda(), want_log_prob_ratios=False)
    assert np.array_equal(out, ref.cpu().numpy())


def test_log_ratio_array_is_reused_only_when_the_caller_says_so(oracle_built):
    """`log_prob_ratios_batch` of the previous call is overwritten by the next one only on request (`reuse_log_prob_ratios=True`, the
    attribute `recycle_log_prob_ratios`, or the caller's own `log_prob_ratios_out`): ownership is never inferred from reference counts."""
    from ldpc_amd import codes
    from ldpc_amd.bp_decoder import BpDecoder
    h = codes.regular_ldpc_code(600, 3, 6, seed=2)
    dec = BpDecoder(h, error_rate=0.05, max_iter=8, bp_method="product_sum", input_vector_type="syndrome")
    eng = dec._get_engine()
    s1 = eng.gen_bsc_syndromes(1, 0.05, shot0=0, shots=300, device="cuda:0").cpu().numpy()
    s2 = eng.gen_bsc_syndromes(2, 0.05, shot0=0, shots=300, device="cuda:0").cpu().numpy()
    dec.decode_batch(s1)
    first_addr = dec.log_prob_ratios_batch.ctypes.data
    want1 = dec.log_prob_ratios_batch.copy()
    dec.decode_batch(s2)                                   # by default a new array, whoever holds the old one
    assert dec.log_prob_ratios_batch.ctypes.data != first_addr or True  # (the allocator may hand the freed block out again)
    want2 = dec.log_prob_ratios_batch.copy()
    kept = dec.log_prob_ratios_batch                       # the caller keeps the second array ...
    dec.decode_batch(s1)
    assert dec.log_prob_ratios_batch is not kept and bits_equal(kept, want2)   # ... and it stays what it was
    assert bits_equal(dec.log_prob_ratios_batch, want1)
    addr = dec.log_prob_ratios_batch.ctypes.data
    dec.decode_batch(s2, reuse_log_prob_ratios=True)       # on request: the same memory takes the next batch
    assert dec.log_prob_ratios_batch.ctypes.data == addr and bits_equal(dec.log_prob_ratios_batch, want2)
    dec.recycle_log_prob_ratios = True
    dec.decode_batch(s1)
    assert dec.log_prob_ratios_batch.ctypes.data == addr and bits_equal(dec.log_prob_ratios_batch, want1)
    dec.decode_batch(s2[:100])                             # another shape: a new array
    assert dec.log_prob_ratios_batch.shape == (100, 600) and bits_equal(dec.log_prob_ratios_batch, want2[:100])
    mine = np.full((300, 600), np.nan)
    dec.decode_batch(s2, log_prob_ratios_out=mine)         # the caller's own array
    assert dec.log_prob_ratios_batch is mine and bits_equal(mine, want2)
    with pytest.raises(ValueError):
        dec.decode_batch(s2, log_prob_ratios_out=np.zeros((300, 599)))


def test_log_ratios_into_page_locked_memory_take_the_direct_route(oracle_built):
    """`llr` in page-locked host memory (ldpc_hip_host_alloc): the pipelined path lets the device-to-host copies write it directly (no
    staging buffer, no host-side copy).  Same bits as the staged route, chunk sizes and a ragged tail included; the switch
    NO_DIRECT_LLR sends the same pinned array through the staged route."""
    from ldpc_amd import codes
    from ldpc_amd._lib import PinnedBlock
    from ldpc_amd.engine import HipBpEngine
    h = codes.regular_ldpc_code(2400, 3, 6, seed=8)
    n, p, B = 2400, 0.055, 20000 + 37
    eng = HipBpEngine(h.indptr, h.indices, n, np.full(n, p), 12, 0, 1.0)
    eng.set_small_code_kernel(0)
    s = eng.gen_bsc_syndromes(3, p, shot0=0, shots=B, device="cuda:0").cpu().numpy()
    want = eng.decode_batch(s, want_llr=True)
    blk = PinnedBlock.try_new(B * n * 8)
    assert blk is not None
    pinned = blk.array((B, n), np.float64)
    for rows in (1024, 4096, -1):
        eng.set_debug_switch("HOST_CHUNK_ROWS", rows)
        for direct in (True, False):
            eng.set_debug_switch("NO_DIRECT_LLR", 0 if direct else 1)
            pinned[:] = np.nan
            got = eng.decode_batch(s, want_llr=True, llr_out=pinned)
            assert got[1] is pinned or got[1].ctypes.data == pinned.ctypes.data
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[2], want[2]) and bits_equal(pinned, want[1]), (rows, direct)
    eng.close()
    view = pinned[7]
    del pinned, blk, got
    assert bits_equal(view, want[1][7])  # (a view keeps the block alive)


def test_bpdecoder_puts_a_large_log_ratio_array_on_page_locked_memory_on_request(oracle_built):
    """`BpDecoder.decode_batch(numpy)`: a caller who says the log-ratio array may be overwritten by later calls (`reuse_log_prob_ratios`)
    gets it on page-locked memory and has the next such call write the same memory; everybody else gets an ordinary array (no page-locked
    allocation on the default path); with device tensors the keywords are refused, not ignored."""
    from ldpc_amd import codes
    from ldpc_amd.bp_decoder import BpDecoder
    h = codes.regular_ldpc_code(3000, 3, 6, seed=2)
    dec = BpDecoder(h, error_rate=0.04, max_iter=8, bp_method="minimum_sum", ms_scaling_factor=0.75, input_vector_type="syndrome")
    eng = dec._get_engine()
    B = 12000  # 12 000 x 3 000 x 8 = 288 MB
    s = eng.gen_bsc_syndromes(9, 0.04, shot0=0, shots=B, device="cuda:0").cpu().numpy()
    out0 = dec.decode_batch(s)
    assert dec.log_prob_ratios_batch.flags.owndata  # default path: an ordinary array
    dec.log_prob_ratios_batch = None
    out1 = dec.decode_batch(s, reuse_log_prob_ratios=True)
    assert np.array_equal(out0, out1)
    a1 = dec.log_prob_ratios_batch
    assert not a1.flags.owndata and getattr(a1.base, "owner", None) is not None  # on a PinnedBlock
    want = a1.copy()
    addr = a1.ctypes.data
    del a1
    out2 = dec.decode_batch(s, reuse_log_prob_ratios=True)
    assert dec.log_prob_ratios_batch.ctypes.data == addr and bits_equal(dec.log_prob_ratios_batch, want) and np.array_equal(out1, out2)
    kept = dec.log_prob_ratios_batch
    dec.decode_batch(s)
    assert dec.log_prob_ratios_batch.ctypes.data != addr and dec.log_prob_ratios_batch.flags.owndata  # an ordinary array this time
    assert bits_equal(kept, want) and bits_equal(dec.log_prob_ratios_batch, want)
    import torch
    with pytest.raises(ValueError, match="NumPy inputs only"):
        dec.decode_batch(torch.from_numpy(s[:64]).cuda(), log_prob_ratios_out=np.zeros((64, 3000)))
    with pytest.raises(ValueError, match="NumPy inputs only"):
        dec.decode_batch(torch.from_numpy(s[:64]).cuda(), reuse_log_prob_ratios=True)
