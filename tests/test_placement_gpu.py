"""choose_placement (pytorch_toolbelt_amd/placement.py): the search keeps the fastest candidate, measures a fixed number when asked
to, and gives everything else back to the driver."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_choose_placement_keeps_the_fastest_and_frees_the_rest():
    from pytorch_toolbelt_amd.placement import choose_placement

    dev = torch.device("cuda:0")
    nbytes = 64 << 20
    made = []

    def allocate():
        t = torch.full((nbytes // 4,), float(len(made)), device=dev)
        made.append(t.data_ptr())
        return t

    script = iter([5.0, 5.1, 4.2, 9.9, 9.9])          # the third candidate is in the "fast class" (4.2 <= 0.89 * 5.1): stop there
    torch.cuda.empty_cache()
    before = torch.cuda.memory_reserved(dev)
    pool, rep = choose_placement(allocate, lambda p: next(script), nbytes, dev, max_tries=5, region_bytes=256 << 20, reserve_bytes=1 << 30)
    assert rep == {"by_candidate": [5.0, 5.1, 4.2], "chosen": 2}
    assert float(pool[0]) == 2.0 and len(made) == 3
    assert torch.cuda.memory_reserved(dev) - before <= 2 * nbytes      # the losers and the spacers went back to the driver
    # fixed count (ranks of a distributed job): exactly that many measurements, whatever they show
    calls = []
    pool2, rep2 = choose_placement(allocate, lambda p: calls.append(1) or 3.0 - len(calls), nbytes, dev, first=pool, fixed_count=4,
                                   region_bytes=256 << 20, reserve_bytes=1 << 30)
    assert len(calls) == 4 and rep2["chosen"] == 3 and float(pool2[0]) == 5.0
