"""A bounded slice of the three differential soaks (tests/fuzz_differential.py, tests/fuzz_streamed.py, tests/fuzz_schedules.py) with FIXED seeds, so the driver's
`-m gpu` run exercises them: random regular codes through every on-chip kernel form and four OSD variants, and through the streamed
kernels (hand-off thresholds, chunked workspaces, ring depths, the two-pass decode), and serial_relative with random starting orders
through every form of the on-chip kernel and the per-lane kernel -- decisions, iterations, flags and log-ratio BITS of
every row against the CPU checker.  The same files run for minutes by hand (`python tests/fuzz_differential.py <seconds> <seed>`)."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [11, 12])
def test_differential_soak_slice(seed, oracle_built):
    import fuzz_differential
    cases = fuzz_differential.run(seconds=20.0, seed=seed, max_cases=40)
    assert cases >= 5, f"only {cases} cases in 20 s"


@pytest.mark.parametrize("seed", [21, 22])
def test_streamed_soak_slice(seed, oracle_built):
    import fuzz_streamed
    cases = fuzz_streamed.run(seconds=20.0, seed=seed, max_cases=16)
    assert cases >= 3, f"only {cases} cases in 20 s"


@pytest.mark.parametrize("seed", [31, 32])
def test_schedules_soak_slice(seed, oracle_built):
    import fuzz_schedules
    cases = fuzz_schedules.run(seconds=20.0, seed=seed, max_cases=60)
    assert cases >= 10, f"only {cases} cases in 20 s"


@pytest.mark.parametrize("seed", [41, 42])
def test_serial_stream_soak_slice(seed, oracle_built):
    """tests/fuzz_serial_stream.py: the streamed serial schedule -- (3,6), (4,8), (3,4), (5,10)-regular and irregular codes, the form built around the
    (6,3) record and the item form, passes, compaction, the per-syndrome kernels, batches whose state is not resident at once."""
    import fuzz_serial_stream
    cases = fuzz_serial_stream.run(seconds=25.0, seed=seed, max_cases=10)
    assert cases >= 2, f"only {cases} cases in 25 s"
