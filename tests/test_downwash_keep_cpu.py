"""Host logic of the kept candidate lists (dronesim_amd/downwash.py; dsim_downwash_args.keep): which queries of a sequence ask the
library to BUILD and which to REUSE, what the step in between is told, and when the sequence starts over — against a stand-in for
the library that records the argument blocks (no GPU, no compute)."""
import ctypes

import torch

from dronesim_amd import _native as nat
from dronesim_amd.downwash import Downwash


class _Lib:
    def __init__(self, ok=1):
        self.calls, self.ok = [], ok
        self.movers = lambda reuses: 3
        self.half = lambda reuses: 5

    def dsim_downwash_prebin_ok(self, m, nx, ny):
        return 1

    def dsim_downwash_workspace(self, m, nx, ny):
        return 64

    def dsim_downwash_keep_ok(self, m, nx, ny, cell, skin):
        self.asked = (m, nx, ny, cell, skin)
        return self.ok

    def dsim_downwash_keep_workspace(self, n_pad, nx, ny):
        return 128

    def dsim_downwash_keep_stats(self, handle, out, half, ofq, q):
        reuses = sum(1 for c in self.calls if c[0] == 2)
        seen = max(reuses - 1, 0)                                                    # the device is one query behind the host
        out._obj.value, half._obj.value, ofq._obj.value, q._obj.value = self.movers(seen), self.half(seen), seen, reuses
        return 0

    def dsim_downwash(self, handle, stream, n, view, ref, force):
        a = ref._obj
        self.calls.append((a.keep, a.prebinned, round(a.keep_skin, 3), a.cell, bool(a.keep_ws), a.keep_age))
        return 0


class _Ctx:
    def __init__(self, lib):
        self.lib, self.handle, self.device = lib, None, torch.device("cpu")

    def stream_ptr(self):
        return None


class _State:
    def __init__(self, n):
        self.n, self.n_pad, self.version = n, n, 0
        self.pos = torch.rand((3, n)) * 40.0

    def raw_fields(self, f0, nf):
        return self.pos[f0:f0 + nf]

    def view(self):
        return nat.View()


def _dw(keep, ok=1, n=1600):
    lib = _Lib(ok)
    st = _State(n)
    return lib, st, Downwash(_Ctx(lib), st, keep_lists=keep, keep_skin=0.1)


def test_one_query_in_k_builds_and_the_step_in_between_is_told_what_comes_next():
    lib, st, dw = _dw(4)
    seen = []
    for k in range(10):
        dw.compute()
        a = ctypes.cast(dw.bin_next_ptr(), ctypes.POINTER(nat.DownwashArgs)).contents       # what dsim_step(bin_next) reads
        seen.append(a.keep)
    assert [c[0] for c in lib.calls] == [1, 2, 2, 2, 1, 2, 2, 2, 1, 2]                       # BUILD, then K - 1 REUSEs
    assert seen == [2, 2, 2, 1, 2, 2, 2, 1, 2, 2]                                            # the NEXT query's kind, known in time
    assert all(c[2] == 0.1 and c[4] for c in lib.calls)
    assert [c[5] for c in lib.calls] == [0, 1, 2, 3, 0, 1, 2, 3, 0, 1]                       # which REUSE of its lists a query is (the moving skin's ring)
    assert abs(lib.calls[0][3] - 5.1) < 1e-6 and abs(lib.asked[3] - 5.1) < 1e-6              # cells of 5 m + the skin
    assert [c[1] for c in lib.calls] == [0] + [1] * 9                                        # the step vouched for every grid but the first


def test_a_new_grid_starts_over_with_a_build():
    lib, st, dw = _dw(8)
    for _ in range(3):
        dw.compute(); dw.bin_next_ptr()
    dw.invalidate_prebin()                          # the state was written behind the grid: new argument block
    dw.compute()
    assert [c[0] for c in lib.calls] == [1, 2, 2, 1] and lib.calls[3][1] == 0
    dw._box_age = dw._box_refresh                   # the box is re-measured: new grid
    dw.compute()
    assert lib.calls[4][0] == 1


def test_off_where_the_shape_or_the_caller_says_so():
    lib, st, dw = _dw(8, ok=0)                      # a shape the library does not keep lists for
    dw.compute(); dw.compute()
    assert [c[0] for c in lib.calls] == [0, 0] and not lib.calls[0][4]
    for keep in (0, 1):
        lib, st, dw = _dw(keep)
        dw.compute(); dw.compute()
        assert [c[0] for c in lib.calls] == [0, 0] and lib.calls[0][3] == 5.0                # ... and plain 5 m cells
    lib, st, dw = _dw(8)
    dw.compute(torch.rand((3, 2000)) * 40.0, local_offset=100)                               # a world given by the caller: plain query
    assert lib.calls[0][0] == 0


def test_a_fleet_in_motion_shortens_the_period_and_then_suspends_the_lists():
    lib, st, dw = _dw(32)
    since = {"build": 0}

    def movers(reuses):                                  # every drone leaves the skin after six steps
        return 0 if reuses - since["build"] < 6 else 1600
    lib.movers = movers
    kinds = []
    for k in range(40):
        dw.compute(); dw.bin_next_ptr()
        kinds.append(lib.calls[-1][0])
        if kinds[-1] == 1:
            since["build"] = sum(1 for c in lib.calls if c[0] == 2)
    assert kinds[:9] == [1, 2, 2, 2, 2, 2, 2, 2, 1]      # the report of the sixth REUSE is read a query late: the ninth query BUILDs
    gaps = [j - i for i, j in zip([i for i, k in enumerate(kinds) if k == 1][:-1], [i for i, k in enumerate(kinds) if k == 1][1:])]
    assert max(gaps[1:]) <= 8 and min(gaps) >= 5         # the period follows the motion
    lib.movers = lambda reuses: 1600                     # ... and a fleet that leaves the skin at once gets plain queries
    for k in range(30):
        dw.compute(); dw.bin_next_ptr()
    assert [c[0] for c in lib.calls[-8:]] == [0] * 8


def test_a_sixteenth_of_the_fleet_half_way_out_builds_before_anybody_has_left():
    lib, st, dw = _dw(32)
    since = {"build": 0}
    lib.half = lambda reuses: 0 if reuses - since["build"] < 3 else 1600           # the whole fleet half way out after three steps
    kinds = []
    for k in range(24):
        dw.compute(); dw.bin_next_ptr()
        kinds.append(lib.calls[-1][0])
        if kinds[-1] == 1:
            since["build"] = sum(1 for c in lib.calls if c[0] == 2)
    assert kinds[:9] == [1, 2, 2, 2, 2, 1, 2, 2, 1]      # read a query late; the period learns the age of the reading: three
    assert 0 not in kinds                                                          # a period of three or four is still worth it
    lib.half = lambda reuses: 60                                                   # a few stragglers: nothing to act on
    n0 = len(lib.calls)
    for k in range(40):
        dw.compute(); dw.bin_next_ptr()
    assert [c[0] for c in lib.calls[n0:]].count(1) <= 8
