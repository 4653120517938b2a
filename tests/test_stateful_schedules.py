"""schedule = 'serial_relative' and random_serial_schedule (bp.hpp:467-483): the two schedules that keep state -- the bit order,
the shuffle generator -- in the reference's decoder OBJECT.  tests/golden/stateful_*.npz hold the real reference's outputs for
  fresh    a new decoder object per syndrome (= the rows of one decode_batch on a new handle),
  carried  one decoder object for all syndromes (= a loop of decode calls on one object).
CPU: the oracle reproduces both.  GPU: the C ABI and the BpDecoder mirror reproduce both -- decisions, iteration counts,
converge flags, log-ratios bit for bit, and the rearranged serial_schedule_order."""
import glob
import os

import numpy as np
import pytest
import scipy.sparse as sp

from golden_util import GOLDEN_DIR, bits_equal

CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "stateful_*.npz")) if "_softrnd_" not in p)
SOFT_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "stateful_softrnd_*.npz")))


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    m, n = int(z["m"]), int(z["n"])
    h = sp.csr_matrix((np.ones(len(z["col_idx"]), np.uint8), z["col_idx"], z["row_ptr"]), shape=(m, n))
    c = dict(h=h, m=m, n=n, probs=z["channel_probs"], max_iter=int(z["max_iter"]), bp_method=int(z["bp_method"]), alpha=float(z["ms_scaling_factor"]),
             schedule=int(z["schedule"]), random=bool(z["random_serial"]), seed=int(z["seed"]), synd=np.unpackbits(z["syndromes"], axis=1, count=m))
    for kind in ("fresh", "carried"):
        c[kind] = (np.unpackbits(z[kind + "_decoding"], axis=1, count=n), z[kind + "_llr"], z[kind + "_iterations"].astype(np.int32),
                   z[kind + "_converge"].astype(bool))
    c["fresh_order_last"] = z["fresh_order_last"].astype(np.int32)
    c["carried_orders"] = z["carried_orders"].astype(np.int32)
    c["order0"] = z["order0"].astype(np.int32) if "order0" in z.files else None  # a serial_schedule_order given to the constructor (*_start*.npz)
    return c


def same(got, want):
    return (np.array_equal(got[0], want[0]) and bits_equal(got[1], want[1]) and np.array_equal(got[2], want[2])
            and np.array_equal(np.asarray(got[3], bool), want[3]))


def test_cases_present():
    assert len(CASES) >= 12 and any("_rel_" in c for c in CASES) and any("_rnd_" in c for c in CASES) and sum("_start" in c for c in CASES) >= 3


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_the_reference(name, oracle_built):
    c = load(name)
    o = oracle_built.BpOracle(c["h"], error_channel=c["probs"], max_iter=c["max_iter"], bp_method=c["bp_method"], ms_scaling_factor=c["alpha"])
    if c["random"]:
        assert same(o.decode_random_serial_batch(c["synd"], c["seed"]), c["fresh"])
    else:
        g = o.decode_serial_relative_batch(c["synd"], order_state=c["order0"], fresh=True)
        assert same(g, c["fresh"]) and np.array_equal(g[4], c["fresh_order_last"])
        g = o.decode_serial_relative_batch(c["synd"], order_state=c["order0"], fresh=False)
        assert same(g, c["carried"]) and np.array_equal(g[4], c["carried_orders"][-1])


def _engine(c, rel_lds=None):
    from ldpc_amd.engine import HipBpEngine
    eng = HipBpEngine(c["h"].indptr, c["h"].indices, c["n"], c["probs"], c["max_iter"], c["bp_method"], c["alpha"])
    eng.set_schedule(c["schedule"], c["order0"])
    if c["random"]:
        eng.set_random_serial(True, c["seed"])
    if rel_lds is not None:  # serial_relative: 0 = the per-lane kernel (state in HBM); 16 / 64 = on chip, that many lanes per syndrome; "walk" = on chip, bit by bit
        if rel_lds == "walk":
            eng.set_debug_switch("REL_LEVELS", 0)
        elif rel_lds == "apart":
            eng.set_debug_switch("REL_SCRATCH_IN_L", 0)
        elif rel_lds in ("ext", "ext_walk"):  # messages and records in global memory (round 6; rows of 5 .. 16, columns of 3 .. 8 entries, else ignored)
            eng.set_debug_switch("REL_EXT", 1)
            if rel_lds == "ext_walk":
                eng.set_debug_switch("REL_LEVELS", 0)
        elif rel_lds == "lds_only":
            eng.set_debug_switch("REL_EXT", 0)
        else:
            eng.set_debug_switch("REL_LDS", rel_lds)
    return eng


KERNELS = [None, 0, 16, 64, "walk", "apart", "ext", "ext_walk", "lds_only"]  # (see _engine: default = on chip, level by level; the random schedule ignores the switches)


@pytest.mark.gpu
@pytest.mark.parametrize("rel_lds", KERNELS)
@pytest.mark.parametrize("name", CASES)
def test_device_batch_is_a_new_decoder_per_row(name, rel_lds):
    c = load(name)
    eng = _engine(c, rel_lds)
    got = eng.decode_batch(c["synd"])
    assert same(got, c["fresh"])
    if not c["random"]:
        assert np.array_equal(eng.schedule_order(), c["fresh_order_last"])
    import torch
    eng2 = _engine(c, rel_lds)
    t = eng2.decode_batch(torch.from_numpy(c["synd"]).cuda())
    assert same(tuple(x.cpu().numpy() for x in t), c["fresh"])
    eng3 = _engine(c, rel_lds)  # without log-ratios, and a ragged last tile
    d = eng3.decode_batch(c["synd"][:37], want_llr=False)
    assert d[1] is None and np.array_equal(d[0], c["fresh"][0][:37]) and np.array_equal(d[2], c["fresh"][2][:37])


@pytest.mark.gpu
@pytest.mark.parametrize("rel_lds", KERNELS)
@pytest.mark.parametrize("name", CASES)
def test_device_one_row_calls_follow_one_reference_object(name, rel_lds):
    c = load(name)
    eng = _engine(c, rel_lds)
    rows = min(len(c["synd"]), 24)
    for b in range(rows):
        got = eng.decode_batch(c["synd"][b:b + 1])
        want = tuple(x[b:b + 1] for x in c["carried"])
        assert same(got, want), f"row {b}"
        if not c["random"]:
            assert np.array_equal(eng.schedule_order(), c["carried_orders"][b])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["stateful_rel_bb144_ps", "stateful_rnd_bb144_ps_s7", "stateful_rnd_over_rel_ham4_s5"])
def test_mirror_decode_sequence_and_batch(name):
    from ldpc_amd.bp_decoder import BpDecoder
    c = load(name)
    kw = dict(error_channel=list(c["probs"]), max_iter=c["max_iter"], bp_method="ps" if c["bp_method"] == 0 else "ms", ms_scaling_factor=c["alpha"],
              schedule={0: "serial", 2: "serial_relative"}[c["schedule"]], input_vector_type="syndrome")
    if c["random"]:
        kw.update(random_schedule_seed=c["seed"], random_serial_schedule=True)
    d = BpDecoder(c["h"], **kw)
    for b in range(16):
        s = c["synd"][b]
        if not s.any():
            break  # (the fixture ran the C++ decoder on every row; the Python layer's all-zero shortcut, pyx:679-681, would not)
        out = d.decode(s)
        assert np.array_equal(out, c["carried"][0][b]) and d.iter == c["carried"][2][b] and d.converge == c["carried"][3][b]
        assert bits_equal(d.log_prob_ratios, c["carried"][1][b])
        if not c["random"]:
            assert np.array_equal(d.serial_schedule_order, c["carried_orders"][b])
    d2 = BpDecoder(c["h"], **kw)
    nz = c["synd"].any(axis=1)
    out = d2.decode_batch(c["synd"])
    assert np.array_equal(out[nz], c["fresh"][0][nz]) and np.array_equal(d2.iter_batch[nz], c["fresh"][2][nz])
    assert bits_equal(d2.log_prob_ratios_batch[nz], c["fresh"][1][nz])


@pytest.mark.gpu
@pytest.mark.parametrize("code,method,alpha,p,max_iter", [("surface21", 1, 0.625, 0.05, 30), ("bb144", 0, 1.0, 0.05, 50), ("bb144", 1, 0.0, 0.08, 12),
                                                          ("ldpc600", 0, 1.0, 0.04, 10), ("hgp1600", 1, 0.625, 0.02, 20), ("hgp1600", 0, 1.0, 0.025, 8)])
def test_serial_relative_on_chip_kernel_against_the_per_lane_kernel_and_the_checker(code, method, alpha, p, max_iter, oracle_built):
    """A few thousand syndromes of the codes the schedule is used on -- thousands of std::sort calls on keys FULL of ties (min-sum
    posteriors; all-equal priors in iteration 1) -- through bp_relative_lds_kernel (parallel re-enactment of the sort) and
    bp_serial_relative_kernel (sequential restatement per lane): every output bit for bit, the final order included; a sample against
    the CPU checker, which is pinned to the real reference (stateful_rel_*.npz) and to the host's std::sort (test_std_sort_port.py)."""
    from ldpc_amd import codes
    from ldpc_amd.engine import HipBpEngine
    h = {"surface21": lambda: codes.rotated_surface_code_x(21), "bb144": codes.bivariate_bicycle_hx,
         "ldpc600": lambda: codes.regular_ldpc_code(600, 3, 6, seed=4),
         "hgp1600": lambda: codes.hypergraph_product_hx(codes.regular_ldpc_code(n=32, dv=3, dc=4, seed=5))}[code]()  # (state beyond LDS: the EXT form by default)
    m, n = h.shape
    B = 3000 if code not in ("ldpc600", "hgp1600") else 700
    outs = {}
    for lds in (64, "apart", "walk", 16, "ext", "ext_walk", 0):  # on chip: level by level (scratch in the posterior array / apart), bit by bit with one / four syndromes per wavefront; 0: the per-lane kernel
        eng = HipBpEngine(h.indptr, h.indices, n, np.full(n, p), max_iter, method, alpha)
        eng.set_schedule("serial_relative")
        if lds == "walk":
            eng.set_debug_switch("REL_LEVELS", 0)
        elif lds == "apart":
            eng.set_debug_switch("REL_SCRATCH_IN_L", 0)
        elif lds in ("ext", "ext_walk"):
            eng.set_debug_switch("REL_EXT", 1)
            if lds == "ext_walk":
                eng.set_debug_switch("REL_LEVELS", 0)
        else:
            eng.set_debug_switch("REL_LDS", lds)
        s = eng.gen_bsc_syndromes(17, p, shot0=0, shots=B, device="cuda:0").cpu().numpy()
        s[7, 0] = 3  # a byte above 1: never converges
        outs[lds] = eng.decode_batch(s) + (eng.schedule_order(),)
        outs[(lds, "ms")] = eng.last_kernel_ms()
        eng.close()
    for lds in (64, "apart", "walk", 16, "ext", "ext_walk"):
        assert same(outs[lds][:4], outs[0][:4]) and np.array_equal(outs[lds][4], outs[0][4]), lds
    assert not outs[64][3][7]
    o = oracle_built.BpOracle(h, error_rate=p, max_iter=max_iter, bp_method=method, ms_scaling_factor=alpha)
    rows = np.r_[0:40, B - 8:B]
    want = o.decode_serial_relative_batch(s[rows], fresh=True)
    assert same(tuple(x[rows] for x in outs[64][:4]), want[:4]) and np.array_equal(outs[64][4], want[4])
    print(f"[serial_relative {code} method {method}: messages in global memory {outs[('ext', 'ms')]:.1f} ms; on chip {outs[(64, 'ms')]:.1f} ms level by level ({outs[('apart', 'ms')]:.1f} with the scratch apart), {outs[('walk', 'ms')]:.1f} / {outs[(16, 'ms')]:.1f} ms bit by bit "
          f"(64 / 16 lanes per syndrome), per-lane kernel {outs[(0, 'ms')]:.1f} ms for {B} syndromes]")


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["permutation", "repeats"])
def test_serial_relative_on_chip_with_a_starting_order(kind, oracle_built):
    """A serial_schedule_order given by the caller: a permutation of the bits goes level by level; one with repeated bits (the reference
    takes any n bit numbers) has no "position of bit b" and must take the bit-by-bit walk -- both against the per-lane kernel and the checker."""
    from ldpc_amd import codes
    from ldpc_amd.engine import HipBpEngine
    h = codes.bivariate_bicycle_hx()
    m, n = h.shape
    rng = np.random.default_rng(5)
    order = rng.permutation(n).astype(np.int32)
    if kind == "repeats":
        order[10:30] = order[40:60]  # twenty bits twice, twenty never
    p, max_iter, B = 0.06, 9, 500
    outs = {}
    for lds in (None, 0):
        eng = HipBpEngine(h.indptr, h.indices, n, np.full(n, p), max_iter, 1, 0.625)
        eng.set_schedule("serial_relative", order)
        if lds is not None:
            eng.set_debug_switch("REL_LDS", lds)
        s = eng.gen_bsc_syndromes(23, p, shot0=0, shots=B, device="cuda:0").cpu().numpy()
        outs[lds] = eng.decode_batch(s) + (eng.schedule_order(),)
        eng.close()
    assert same(outs[None][:4], outs[0][:4]) and np.array_equal(outs[None][4], outs[0][4])
    o = oracle_built.BpOracle(h, error_rate=p, max_iter=max_iter, bp_method=1, ms_scaling_factor=0.625)
    want = o.decode_serial_relative_batch(s, order_state=order, fresh=True)
    assert same(outs[None][:4], want[:4]) and np.array_equal(outs[None][4], want[4])


@pytest.mark.gpu
@pytest.mark.parametrize("code,method,alpha,p,max_iter", [("surface21", 1, 0.625, 0.05, 30), ("bb144", 0, 1.0, 0.05, 50), ("ldpc600", 0, 1.0, 0.04, 10)])
def test_random_serial_level_kernel_against_the_walking_kernel_and_the_checker(code, method, alpha, p, max_iter, oracle_built):
    """random_serial_schedule: every iteration has its own order (the same for all rows).  The ring of per-iteration orders also goes up
    cut into levels of check-disjoint bits, and bp_serial_level_kernel takes iteration t's levels from row t -- against the kernel that
    walks the order bit by bit (set_serial_kernel(0)) and the checker; two calls in a row (the second starts where the first one's last
    row left the generator, and reuses the ring minus the rows consumed)."""
    from ldpc_amd import codes
    from ldpc_amd.engine import HipBpEngine
    h = {"surface21": lambda: codes.rotated_surface_code_x(21), "bb144": codes.bivariate_bicycle_hx,
         "ldpc600": lambda: codes.regular_ldpc_code(600, 3, 6, seed=4)}[code]()
    m, n = h.shape
    B = 2000
    outs = {}
    for mode in (1, 0, -1):
        eng = HipBpEngine(h.indptr, h.indices, n, np.full(n, p), max_iter, method, alpha)
        eng.set_schedule("serial")
        eng.set_random_serial(True, 77)
        eng.set_serial_kernel(mode)
        s = eng.gen_bsc_syndromes(19, p, shot0=0, shots=B, device="cuda:0").cpu().numpy()
        first = eng.decode_batch(s)
        ms = eng.last_kernel_ms()
        second = eng.decode_batch(s[:64])
        outs[mode] = (first, second, eng.schedule_order(), ms)
        eng.close()
    for mode in (1, -1):
        assert same(outs[mode][0], outs[0][0]) and same(outs[mode][1], outs[0][1]) and np.array_equal(outs[mode][2], outs[0][2]), mode
    o = oracle_built.BpOracle(h, error_rate=p, max_iter=max_iter, bp_method=method, ms_scaling_factor=alpha)
    rows = np.r_[0:40, B - 8:B]
    assert same(tuple(x[rows] for x in outs[1][0]), o.decode_random_serial_batch(s[rows], 77))
    print(f"[random serial {code} method {method}: level kernel {outs[1][3]:.2f} ms, walking kernel {outs[0][3]:.2f} ms, default {outs[-1][3]:.2f} ms for {B} syndromes]")


# ---- SoftInfoBpDecoder with random_serial_schedule (bp.hpp:573-577): the order the object carries is rearranged at the top of
# every iteration that runs by std::shuffle with a NEW std::default_random_engine(random_schedule_seed) -----------------------------

def load_soft(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    m, n = int(z["m"]), int(z["n"])
    h = sp.csr_matrix((np.ones(len(z["col_idx"]), np.uint8), z["col_idx"], z["row_ptr"]), shape=(m, n))
    c = dict(h=h, m=m, n=n, probs=z["channel_probs"], max_iter=int(z["max_iter"]), alpha=float(z["ms_scaling_factor"]), cutoff=float(z["cutoff"]),
             sigma=float(z["sigma"]), seed=int(z["seed"]), soft=z["soft_syndromes"])
    for kind in ("fresh", "carried"):
        c[kind] = (np.unpackbits(z[kind + "_decoding"], axis=1, count=n), z[kind + "_llr"], z[kind + "_iterations"].astype(np.int32),
                   z[kind + "_converge"].astype(bool), z[kind + "_soft_out"])
    c["fresh_order_last"] = z["fresh_order_last"].astype(np.int32)
    c["carried_orders"] = z["carried_orders"].astype(np.int32)
    return c


def same_soft(got, want):
    return same(got, want) and bits_equal(got[4], want[4])


def test_soft_cases_present():
    assert len(SOFT_CASES) >= 3
    its = np.concatenate([load_soft(n)["fresh"][2] for n in SOFT_CASES])
    assert len(np.unique(its)) >= 5  # rows stop at different iterations: the order a row leaves differs from row to row


@pytest.mark.parametrize("name", SOFT_CASES)
def test_oracle_reproduces_the_reference_soft(name, oracle_built):
    c = load_soft(name)
    o = oracle_built.BpOracle(c["h"], error_channel=c["probs"], max_iter=c["max_iter"], bp_method="minimum_sum", ms_scaling_factor=c["alpha"])
    assert same_soft(o.soft_info_decode_random_batch(c["soft"], c["cutoff"], c["sigma"], c["seed"]), c["fresh"])
    orders, _ = oracle_built.shuffle_orders_reseeded(c["seed"], c["n"], c["max_iter"])
    assert np.array_equal(orders[c["fresh"][2][-1] - 1], c["fresh_order_last"])
    order = None  # one object: the order carries over
    for b in range(len(c["soft"])):
        g = o.soft_info_decode_random_batch(c["soft"][b:b + 1], c["cutoff"], c["sigma"], c["seed"], order)
        assert same_soft(g, tuple(x[b:b + 1] for x in c["carried"])), f"row {b}"
        order = oracle_built.shuffle_orders_reseeded(c["seed"], c["n"], int(g[2][0]), order)[1]
        assert np.array_equal(order, c["carried_orders"][b])


def _soft_engine(c, serial_kernel=-1):
    from ldpc_amd.engine import HipBpEngine
    eng = HipBpEngine(c["h"].indptr, c["h"].indices, c["n"], c["probs"], c["max_iter"], 1, c["alpha"])
    eng.set_schedule("serial")
    eng.set_random_serial(True, c["seed"] & 0xffffffff)
    eng.set_serial_kernel(serial_kernel)  # 0: the kernel that walks the order; 1: the level kernel on the ring's per-iteration levels; -1: by the levels' width
    return eng


@pytest.mark.gpu
@pytest.mark.parametrize("serial_kernel", [0, 1])
@pytest.mark.parametrize("name", SOFT_CASES)
def test_device_soft_batch_both_kernels(name, serial_kernel):
    c = load_soft(name)
    eng = _soft_engine(c, serial_kernel)
    assert same_soft(eng.soft_info_decode_batch(c["soft"], c["cutoff"], c["sigma"]), c["fresh"])
    assert np.array_equal(eng.schedule_order(), c["fresh_order_last"])
    eng.close()
    eng3 = _soft_engine(c, serial_kernel)
    for b in range(min(len(c["soft"]), 12)):
        got = eng3.soft_info_decode_batch(c["soft"][b:b + 1], c["cutoff"], c["sigma"])
        assert same_soft(got, tuple(x[b:b + 1] for x in c["carried"])), f"row {b}"
        assert np.array_equal(eng3.schedule_order(), c["carried_orders"][b])
    eng3.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", SOFT_CASES)
def test_device_soft_batch_and_one_row_calls(name):
    c = load_soft(name)
    eng = _soft_engine(c)
    assert same_soft(eng.soft_info_decode_batch(c["soft"], c["cutoff"], c["sigma"]), c["fresh"])
    assert np.array_equal(eng.schedule_order(), c["fresh_order_last"])
    import torch
    eng2 = _soft_engine(c)
    t = eng2.soft_info_decode_batch(torch.from_numpy(c["soft"]).cuda(), c["cutoff"], c["sigma"])
    assert same_soft(tuple(x.cpu().numpy() for x in t), c["fresh"])
    eng3 = _soft_engine(c)  # one object, one row after the other: the order carries over
    for b in range(min(len(c["soft"]), 24)):
        got = eng3.soft_info_decode_batch(c["soft"][b:b + 1], c["cutoff"], c["sigma"])
        assert same_soft(got, tuple(x[b:b + 1] for x in c["carried"])), f"row {b}"
        assert np.array_equal(eng3.schedule_order(), c["carried_orders"][b])


@pytest.mark.gpu
@pytest.mark.parametrize("name", [n for n in SOFT_CASES if "_sneg" not in n])  # (the Python layer refuses seeds below -2, pyx:548)
def test_soft_mirror_decode_sequence_and_batch(name):
    from ldpc_amd.bp_decoder import SoftInfoBpDecoder
    c = load_soft(name)
    kw = dict(error_channel=list(c["probs"]), max_iter=c["max_iter"], ms_scaling_factor=c["alpha"], cutoff=c["cutoff"], sigma=c["sigma"],
              random_schedule_seed=c["seed"], random_serial_schedule=True)
    d = SoftInfoBpDecoder(c["h"], **kw)
    for b in range(16):
        out = d.decode(c["soft"][b])
        assert np.array_equal(out, c["carried"][0][b]) and d.iter == c["carried"][2][b] and d.converge == c["carried"][3][b]
        assert bits_equal(d.log_prob_ratios, c["carried"][1][b]) and bits_equal(d.soft_syndrome, c["carried"][4][b])
        assert np.array_equal(d.serial_schedule_order, c["carried_orders"][b])
    d2 = SoftInfoBpDecoder(c["h"], **kw)
    out = d2.decode_batch(c["soft"])
    assert np.array_equal(out, c["fresh"][0]) and np.array_equal(d2.iter_batch, c["fresh"][2])
    assert bits_equal(d2.log_prob_ratios_batch, c["fresh"][1]) and bits_equal(d2.soft_syndrome_batch, c["fresh"][4])


@pytest.mark.gpu
@pytest.mark.parametrize("code", ["bb144", "hgp1600"])
def test_serial_relative_with_nan_keys_on_every_kernel_form(code, oracle_built):
    """Priors of exactly 0 and 1 give posteriors inf - inf = NaN: the comparator of bp.hpp:472-482 is then no strict weak order and the
    reference's result is whatever its std::sort's loops happen to do -- the sequential restatement (oracle/: pinned to the host's std::sort
    on ordered keys) that every kernel form falls back to as soon as a key is NaN.  The on-chip forms -- all in LDS (BB144), messages and
    posteriors in global memory with the ranks in chunks (hgp1600: n > 512) -- against the per-lane kernel and the checker."""
    from ldpc_amd import codes
    from ldpc_amd.engine import HipBpEngine
    h = {"bb144": codes.bivariate_bicycle_hx, "hgp1600": lambda: codes.hypergraph_product_hx(codes.regular_ldpc_code(n=32, dv=3, dc=4, seed=5))}[code]()
    m, n = h.shape
    probs = np.full(n, 0.03)
    row0 = h.indices[h.indptr[0]:h.indptr[1]]  # every bit of check 0 certain, the last one the other way: its posterior is -inf + inf wherever
    probs[row0[:-1]] = 0.0                      # the syndrome bit is 0 (the check's message to it is +inf)
    probs[row0[-1]] = 1.0
    probs[n - 3] = 0.0
    B = 300
    outs = {}
    for form in ("default", "ext_walk", 0):
        eng = HipBpEngine(h.indptr, h.indices, n, probs, 6, 0, 1.0)
        eng.set_schedule("serial_relative")
        if form == "ext_walk":
            eng.set_debug_switch("REL_EXT", 1)
            eng.set_debug_switch("REL_LEVELS", 0)
        elif form == 0:
            eng.set_debug_switch("REL_LDS", 0)
        s = eng.gen_bsc_syndromes(23, 0.03, shot0=0, shots=B, device="cuda:0").cpu().numpy()
        outs[form] = eng.decode_batch(s) + (eng.schedule_order(),)
        eng.close()
    assert np.isnan(outs[0][1]).any()  # (the case is what it says)
    for form in ("default", "ext_walk"):
        assert same(outs[form][:4], outs[0][:4]) and np.array_equal(outs[form][4], outs[0][4]), form
    o = oracle_built.BpOracle(h, error_channel=probs, max_iter=6, bp_method=0)
    rows = np.r_[0:30, B - 1]
    want = o.decode_serial_relative_batch(s[rows], fresh=True)
    assert same(tuple(x[rows] for x in outs[0][:4]), want[:4]) and np.array_equal(outs[0][4], want[4])
