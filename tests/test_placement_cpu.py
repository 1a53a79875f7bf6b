"""The search logic of pytorch_toolbelt_amd.placement.choose_placement, driven without a device (free memory and the gap
allocations are injected): stopping rules, fixed counts for distributed ranks, out-of-memory handling, spacing of the candidates."""

from pytorch_toolbelt_amd.placement import REGION_BYTES, choose_placement

GB = 1 << 30


class _Mem:
    def __init__(self, free):
        self.free, self.spacers, self.made = free, [], 0

    def allocate(self, nbytes):
        def f():
            if self.free < nbytes:
                raise RuntimeError("out of memory")
            self.free -= nbytes
            self.made += 1
            return ("pool", self.made - 1)
        return f

    def spacer(self, n):
        self.free -= n
        self.spacers.append(n)
        return ("gap", n)


def _run(times, free=288 * GB, nbytes=12 * GB, **kw):
    mem = _Mem(free)
    it = iter(times)
    pool, rep = choose_placement(mem.allocate(nbytes), lambda p: next(it), nbytes, "cpu", free_bytes=lambda: mem.free, make_spacer=mem.spacer, **kw)
    return pool, rep, mem


def test_stops_at_the_first_candidate_of_the_fast_class():
    pool, rep, mem = _run([2.30, 2.29, 2.31, 1.99, 9.9])
    assert rep == {"by_candidate": [2.30, 2.29, 2.31, 1.99], "chosen": 3} and pool == ("pool", 3)
    assert mem.spacers == [REGION_BYTES - 12 * GB] * 3          # every candidate one region further


def test_needs_three_candidates_before_it_believes_a_difference():
    pool, rep, _ = _run([2.30, 1.90, 2.00, 9.9])
    assert rep["by_candidate"] == [2.30, 1.90, 2.00] and rep["chosen"] == 1


def test_all_alike_runs_to_max_tries_and_takes_the_best():
    pool, rep, mem = _run([2.0, 2.01, 1.99, 2.02, 2.0], max_tries=5)
    assert len(rep["by_candidate"]) == 5 and rep["chosen"] == 2 and mem.made == 5


def test_memory_running_short_ends_the_search():
    pool, rep, mem = _run([2.3, 2.3, 2.3, 2.3, 2.3, 2.3, 2.3, 2.3], free=100 * GB, max_tries=8)
    assert 2 <= len(rep["by_candidate"]) < 8
    assert all(s > GB for s in mem.spacers)


def test_fixed_count_for_distributed_ranks_even_when_allocation_fails():
    calls = []
    mem = _Mem(30 * GB)
    pool, rep = choose_placement(mem.allocate(12 * GB), lambda p: calls.append(p) or 2.0 + 0.01 * len(calls), 12 * GB, "cpu", fixed_count=5,
                                 free_bytes=lambda: mem.free, make_spacer=mem.spacer)
    assert len(calls) == 5 and len(rep["by_candidate"]) == 5        # every rank measures 5 times, memory or not
    assert rep["chosen"] == 0


def test_first_pool_is_candidate_zero():
    pool, rep, mem = _run([1.0, 2.0, 2.0], first=("mine", 0), max_tries=3)
    assert pool == ("mine", 0) and mem.made == 2
