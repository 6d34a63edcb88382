"""Host logic of dronesim_amd/placement.py (which candidate the walk keeps, when it stops), with a scripted clock: no
device needed.  What the placement is worth is measured on the GPU (tests/test_gpu_storage_halo_placement.py, profiles/r03_placement_*)."""
import pytest

torch = pytest.importorskip("torch")

from dronesim_amd import placement  # noqa: E402


def walk(times, shape=(1024, 20), walk_bytes=None):
    """Runs place_rows on the CPU with the i-th candidate 'taking' times[i] microseconds; returns (report, array, passes seen)."""
    seen, log = [], []
    it = iter(times)

    def timer(trial, c, passes):
        trial(c)
        seen.append(passes)
        return next(it)
    touched = []
    out = placement.place_rows("cpu", shape, lambda c: touched.append(c.data_ptr()), report=log,
                               walk_bytes=walk_bytes if walk_bytes is not None else 4 * shape[0] * shape[1] * len(times), timer=timer)
    return log[0], out, touched


def test_walk_stops_at_the_first_clearly_faster_candidate():
    rep, out, touched = walk([165.0, 164.0, 166.0, 142.0, 141.0, 150.0])
    assert rep["candidates"] == 4 and rep["chosen"] == 3 and rep["chosen_pass_us"] == 142.0 and rep["first_pass_us"] == 165.0
    assert out.shape == (1024, 20) and float(out.abs().max()) == 0.0 and out.data_ptr() == touched[3]


def test_a_clearly_slower_candidate_does_not_end_the_walk():
    """There are more than two levels: the first candidate may be the middle one."""
    rep, out, touched = walk([157.0, 166.0, 158.0, 154.0, 158.5])
    assert rep["candidates"] == 5 and rep["chosen"] == 3 and rep["decided_by"] == "the fastest of the walk" and out.data_ptr() == touched[3]
    rep, out, touched = walk([157.0, 165.0, 158.0, 145.0, 120.0])              # ... but a clearly faster one does
    assert rep["candidates"] == 4 and rep["chosen"] == 3 and out.data_ptr() == touched[3]


def test_fastest_of_a_flat_walk_is_kept_when_the_budget_is_spent():
    times = [160.0, 161.0, 158.5, 159.0, 162.0]
    rep, out, touched = walk(times)
    assert rep["candidates"] == 5 and rep["chosen"] == 2 and rep["decided_by"] == "all alike" and out.data_ptr() == touched[2]
    # the budget counts what is held at once: two candidates' worth of bytes -> two candidates
    rep, _, _ = walk(times, walk_bytes=2 * 4 * 1024 * 20)
    assert rep["candidates"] == 2 and rep["chosen"] == 0
    # a long walk reports its first and last four timings only
    rep, _, _ = walk([160.0 + 0.01 * k for k in range(20)])
    assert rep["candidates"] == 20 and "pass_us_first4_last4" in rep and len(rep["pass_us_first4_last4"]) == 8


def test_walk_is_bounded_by_the_free_memory_and_falls_back_to_a_plain_allocation():
    """ADVICE r3 / VERDICT r3 next 9: the walk holds at most WALK_FRACTION of the free device memory; with room for fewer than
    two candidates nothing is searched (a plain zeroed array), and a refused allocation ends the walk with what it has —
    never an exception, never an empty candidate list."""
    nbytes = 4 * 1024 * 20
    log, seen = [], []
    out = placement.place_rows("cpu", (1024, 20), lambda c: seen.append(1), report=log, timer=lambda t, c, p: 100.0,
                               free_bytes=int(3.5 * nbytes / placement.WALK_FRACTION))
    assert log[0]["candidates"] == 3 and log[0]["peak_bytes"] == 3 * nbytes and log[0]["budget_bytes"] < 4 * nbytes
    assert log[0]["seconds"] >= 0 and out.shape == (1024, 20)
    log = []
    out = placement.place_rows("cpu", (1024, 20), lambda c: seen.append(1), report=log, timer=lambda t, c, p: 100.0,
                               free_bytes=int(1.5 * nbytes / placement.WALK_FRACTION))
    assert log[0]["candidates"] == 0 and log[0]["decided_by"].startswith("no walk") and float(out.abs().max()) == 0.0
    assert out.shape == (1024, 20) and log[0]["peak_bytes"] == 0

    class Ctx:                                   # a driver that gives two blocks and then refuses
        handle, device, given = 1, "cpu", 0

        class lib:
            @staticmethod
            def dsim_dev_alloc(h, n, out):
                return -1

            @staticmethod
            def dsim_dev_free(h, p_):
                return 0
    log = []
    out = placement.place_rows("cpu", (1024, 20), lambda c: None, report=log, timer=lambda t, c, p: 100.0, ctx=Ctx())
    assert log[0]["candidates"] == 0 and out.shape == (1024, 20) and float(out.abs().max()) == 0.0      # the very first one refused


def test_a_striding_walk_covers_the_budget_with_fewer_candidates():
    """Round 4: `stride_bytes` of untimed ballast behind every candidate (held, inside the budget): the walk leaves the
    neighbourhood of its first candidates — the 16 GiB region of the state block — with a handful of timed candidates.  The
    ballast counts in `peak_bytes`, never exceeds the budget, and a clearly faster candidate still ends the walk at once."""
    nbytes = 4 * 1024 * 20
    stride = 3 * nbytes
    log = []
    placement.place_rows("cpu", (1024, 20), lambda c: None, report=log, timer=lambda t, c, p: 100.0,
                         free_bytes=int(10.5 * nbytes / placement.WALK_FRACTION), stride_bytes=stride)
    # budget 10.5 candidates' worth: candidate (1) + ballast (3) + candidate + ballast + candidate = 9; a third ballast + candidate would be 13
    assert log[0]["candidates"] == 3 and log[0]["peak_bytes"] == 3 * nbytes + 2 * stride <= log[0]["budget_bytes"]
    assert log[0]["stride_bytes"] == stride
    log = []
    times = iter([160.0, 161.0, 143.0, 150.0])
    placement.place_rows("cpu", (1024, 20), lambda c: None, report=log, timer=lambda t, c, p: next(times),
                         free_bytes=int(100 * nbytes / placement.WALK_FRACTION), stride_bytes=stride)
    assert log[0]["candidates"] == 3 and log[0]["chosen"] == 2 and log[0]["decided_by"].startswith("a candidate clearly faster")
    assert log[0]["peak_bytes"] == 3 * nbytes + 2 * stride
