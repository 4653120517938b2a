"""CPU-side checks of the DEVICE math (ldpc_amd/csrc/bp_math.h compiled for the host).

(1) accuracy of tanh_half / log_pos against the host libm and an 80-bit reference;
(2) the oracle with the device routines plugged in (oracle/bp_oracle.c: bp_oracle_set_math) must
    reproduce every product-sum golden fixture: hard decisions / converge / iterations exactly,
    LLRs within the north_star's 1e-5 relative.  This predicts on the CPU what the HIP kernel does.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from golden_util import case_names, load_case, llr_close

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "native", "device_math_host.cpp")
SO = os.path.join(HERE, "native", "libdevice_math_host.so")
HDR = os.path.join(os.path.dirname(HERE), "ldpc_amd", "csrc", "bp_math.h")

_dp = np.ctypeslib.ndpointer(np.float64, flags="C")
_ldp = np.ctypeslib.ndpointer(np.longdouble, flags="C")


@pytest.fixture(scope="module")
def dm():
    if (not os.path.exists(SO)) or os.path.getmtime(SO) < max(os.path.getmtime(SRC), os.path.getmtime(HDR)):
        subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-mfma", "-fPIC", "-shared", "-o", SO, SRC],
                       check=True)
    lib = C.CDLL(SO)
    for f in ("dm_tanh_half_v", "dm_log_pos_v", "dm_log_ratio_v", "libm_tanh_half_v", "libm_log_v",
              "dx_tanh_half_v", "dx_log_v", "dx_log_ratio_v", "libm_log_ratio_v", "dx_log_split_v"):
        getattr(lib, f).argtypes = [C.c_long, _dp, _dp]
    for f in ("ref_tanh_half_v", "ref_log_v"):
        getattr(lib, f).argtypes = [C.c_long, _dp, _ldp]
    for f in ("dm_tanh_half_ptr", "dm_log_ratio_ptr", "dx_tanh_half_ptr", "dx_log_ratio_ptr"):
        getattr(lib, f).restype = C.c_void_p
    return lib


def _ulps(got, ref):
    u = np.spacing(np.abs(ref.astype(np.float64))).astype(np.longdouble)
    return np.abs(got.astype(np.longdouble) - ref) / u


@pytest.mark.parametrize("lo,hi,logspace,max_ulp,max_mismatch", [
    (1e-300, 1e-10, True, 1.1, 1e-3),
    (1e-10, 1e-3, True, 1.6, 1e-3),
    (1e-3, 2.0, False, 2.6, 0.15),
    (2.0, 10.0, False, 1.0, 0.02),
    (10.0, 37.0, False, 0.51, 1e-4),   # saturation-critical: must round like the host libm
    (37.0, 39.0, False, 0.51, 1e-4),
    (39.0, 1000.0, False, 0.51, 0.0),
])
def test_tanh_half_accuracy(dm, lo, hi, logspace, max_ulp, max_mismatch):
    rng = np.random.default_rng(5)
    n = 200_000
    b = np.exp(rng.uniform(np.log(lo), np.log(hi), n)) if logspace else rng.uniform(lo, hi, n)
    b *= rng.choice([-1.0, 1.0], n)
    got, libm, ref = np.empty(n), np.empty(n), np.empty(n, np.longdouble)
    dm.dm_tanh_half_v(n, b, got)
    dm.libm_tanh_half_v(n, b, libm)
    dm.ref_tanh_half_v(n, b, ref)
    assert float(_ulps(got, ref).max()) <= max_ulp
    assert np.mean(got != libm) <= max_mismatch
    assert np.all(np.abs(got) <= 1.0) and np.all(np.sign(got) == np.sign(b))


@pytest.mark.parametrize("lo,hi", [(2.0 ** -54, 2.0 ** 54), (0.5, 2.0), (0.99, 1.01), (1 - 1e-8, 1 + 1e-8)])
def test_log_pos_accuracy(dm, lo, hi):
    rng = np.random.default_rng(6)
    n = 200_000
    q = np.exp(rng.uniform(np.log(lo), np.log(hi), n))
    got, ref = np.empty(n), np.empty(n, np.longdouble)
    dm.dm_log_pos_v(n, q, got)
    dm.ref_log_v(n, q, ref)
    assert float(_ulps(got, ref).max()) <= 1.0


def test_special_values(dm):
    def th(v):
        o = np.empty(1); dm.dm_tanh_half_v(1, np.array([float(v)]), o); return o[0]

    def lr(v):
        o = np.empty(1); dm.dm_log_ratio_v(1, np.array([float(v)]), o); return o[0]
    assert th(0.0) == 0.0 and np.signbit(th(-0.0)) and th(np.inf) == 1.0 and th(-np.inf) == -1.0
    assert np.isnan(th(np.nan)) and th(1e308) == 1.0 and th(-50.0) == -1.0
    assert th(38.0) == np.nextafter(1.0, 0.0) and th(38.2) == 1.0  # same saturation point as tanh()
    assert lr(1.0) == np.inf and lr(-1.0) == -np.inf and lr(0.0) == 0.0 and np.isnan(lr(np.nan))
    assert lr(np.nextafter(1.0, 0.0)) == np.log(2.0 / 2.0 ** -53 - 1.0) or abs(lr(np.nextafter(1.0, 0.0)) - 37.42994775023705) < 1e-13


from golden_util import bits_equal as _bits_equal  # noqa: E402


def test_libm_twins_are_bit_identical_to_the_host_libm(dm):
    """tanh_half_libm / log_libm / ps_log_ratio_libm vs the host's std::tanh / std::log, bit for bit.

    Holds on the x86-64 glibc (>= 2.28, FMA-capable CPU) the golden vectors were captured on; a host
    with a different libm may legitimately differ -- then the goldens (captured outputs) are the pin.
    """
    import platform
    rng = np.random.default_rng(3)
    n = 1_000_000
    b = np.concatenate([rng.uniform(-100, 100, n), rng.uniform(-4, 4, n),
                        np.exp(rng.uniform(-120, 5, n)) * rng.choice([-1.0, 1.0], n),
                        [0.0, -0.0, np.inf, -np.inf, np.nan, 44.0, 43.99, -44.0, 2e-16, 1e-17, 2.0 ** -54, 2.0 ** -55, 4.0, 2.0, -2.0]])
    q = np.concatenate([np.exp(rng.uniform(-37.5, 37.5, n)), rng.uniform(0.9, 1.1, n), rng.uniform(0.25, 4, n),
                        [0.0, np.inf, np.nan, 1.0, 2.0 ** -54, 2.0 ** 54]])
    x = np.concatenate([rng.uniform(-1, 1, n), np.tanh(rng.uniform(-20, 20, n)), rng.uniform(-1e-3, 1e-3, n),
                        [1.0, -1.0, 0.0, -0.0, np.nan, 1 - 2.0 ** -53, -1 + 2.0 ** -53]])
    # arguments around every threshold of the tanh twin's argument reduction and tails (high-word boundaries, k changes)
    edges = []
    for v in (0.5 * np.log(2) / 1, 1.5 * np.log(2), 2.0, 13.0, 13.5, 20 * np.log(2), 44.0, 2.0 ** -54, 2.0 ** -55, 2.0 ** -1022, 5e-324,
              *(k * np.log(2) for k in range(1, 64)), *((k + 0.5) * np.log(2) for k in range(0, 64))):
        e = np.float64(v)
        around = [e]
        for _ in range(40):
            around.append(np.nextafter(around[-1], np.inf))
        lo_ = e
        for _ in range(40):
            lo_ = np.nextafter(lo_, -np.inf)
            around.append(lo_)
        edges.extend(around)
    hw = np.array([0x3fd62e42, 0x3fd62e43, 0x3ff0a2b1, 0x3ff0a2b2, 0x3ff0a2b3, 0x40000000, 0x3fffffff, 0x402a0000, 0x4029ffff, 0x3c900000, 0x3c8fffff,
                   0x40460000, 0x4045ffff], dtype=np.uint64)
    words = np.concatenate([(hw << np.uint64(32)) | np.uint64(lo32) for lo32 in (0, 1, 0xffffffff, 0x80000000)]).view(np.float64)
    edges = np.concatenate([np.array(edges, dtype=np.float64), words])
    b = np.concatenate([b, edges, -edges])
    q_norm = np.concatenate([np.exp(rng.uniform(-37.4, 37.4, n)), rng.uniform(0.9, 1.1, n), [1.0, 2.0 ** -54, 2.0 ** 54, 0.9375, np.nextafter(0.9375, 0),
                             1.064697265625, np.nextafter(1.064697265625, 0), np.nextafter(1.0, 0), np.nextafter(1.0, 2)]])
    ok = True
    for mine, libm, arg in (("dx_tanh_half_v", "libm_tanh_half_v", b), ("dx_log_v", "libm_log_v", q),
                            ("dx_log_ratio_v", "libm_log_ratio_v", x), ("dx_log_split_v", "libm_log_v", q_norm)):
        got, want = np.empty(len(arg)), np.empty(len(arg))
        getattr(dm, mine)(len(arg), arg, got)
        getattr(dm, libm)(len(arg), arg, want)
        ok = ok and _bits_equal(got, want)
    if not ok and not (platform.machine() == "x86_64" and "glibc" in platform.libc_ver()[0]):
        pytest.skip("host libm is not x86-64 glibc; bit-identity is defined against that library")
    assert ok


PS_CASES = [n for n in case_names() if load_case(n)["bp_method"] == "product_sum"]


def _decode_with(dm, c, oracle_built, tanh_ptr, log_ptr):
    o = oracle_built.BpOracle(c["h"], error_channel=c["channel_probs"], max_iter=c["max_iter"],
                              bp_method="product_sum", ms_scaling_factor=c["ms_scaling_factor"])
    o.lib.bp_oracle_set_math.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    o.lib.bp_oracle_set_math(o._h, tanh_ptr, log_ptr)
    return o.decode_batch(c["syndromes"])


@pytest.mark.parametrize("name", PS_CASES)
def test_oracle_with_libm_twins_reproduces_reference_bit_for_bit(dm, name, oracle_built):
    """The kernel's default math (LDPC_MATH=2), emulated on the CPU: reference LLRs to the last bit."""
    c = load_case(name)
    dec, llr, it, cv = _decode_with(dm, c, oracle_built, dm.dx_tanh_half_ptr(), dm.dx_log_ratio_ptr())
    assert np.array_equal(dec, c["decoding"])
    assert np.array_equal(cv, c["converge"])
    assert np.array_equal(it, c["iterations"])
    assert _bits_equal(llr[: len(c["llr"])], c["llr"])


# the one fixture whose reference posterior is EXACTLY 0.0 (prior - 2 atanh(tanh(prior/2)) on a
# weight-1 column under a weight-2 check): only a bit-identical libm reproduces that knife edge
KNIFE_EDGE = {"edge_degree1_empty_ps"}


@pytest.mark.parametrize("name", [n for n in PS_CASES if n not in KNIFE_EDGE])
def test_oracle_with_fast_math_stays_within_tolerance(dm, name, oracle_built):
    """The optional fast routines (LDPC_MATH=1): exact decisions, LLRs within 1e-5 relative."""
    c = load_case(name)
    dec, llr, it, cv = _decode_with(dm, c, oracle_built, dm.dm_tanh_half_ptr(), dm.dm_log_ratio_ptr())
    assert np.array_equal(dec, c["decoding"])
    assert np.array_equal(cv, c["converge"])
    assert np.array_equal(it, c["iterations"])
    assert llr_close(llr[: len(c["llr"])], c["llr"], rtol=1e-5)
