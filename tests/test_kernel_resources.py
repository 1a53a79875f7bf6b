"""Register budgets of the hot kernels, from the compiler's own report (hipcc cross-compiles for gfx950 without a GPU).

The kernels of the benchmarked paths sit right at occupancy cliffs: the fused multiscale kernel at 80 VGPRs (6 waves per SIMD; a
two-register drift once cost its gmean instance 7 %), the band-plan kernel must not touch scratch memory (run-time indexing of a
by-value struct once put 52 B per lane there: 2.58 instead of 2.19 ms per image), the straight-line loss kernels must not spill.
The budgets are read from the -Rpass-analysis=kernel-resource-usage remarks of the session's forced rebuild (tests/conftest.py)."""
import re
import subprocess
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import pytest

import __graft_entry__ as entry

CSRC = Path(entry.CSRC)


def _report(path):
    """Per-kernel resource usage parsed from the compiler's -Rpass-analysis=kernel-resource-usage remarks of one translation unit."""
    kernels, cur = {}, None
    for line in open(path).read().splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = kernels.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+(VGPRs|VGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).split(" [")[0]] = int(m.group(2))
    return kernels


@pytest.fixture(scope="module")
def reports(forced_build):
    d = Path(forced_build["remarks_dir"])      # (the session's one forced rebuild of every translation unit: tests/conftest.py)
    return dict(resample=_report(d / "ptb_resample.hip.txt"), bandplan=_report(d / "ptb_bandplan.hip.txt"), losses=_report(d / "ptb_losses.hip.txt"))


def _find(kernels, *needles):
    hits = {k: v for k, v in kernels.items() if all(n in k for n in needles)}
    assert hits, f"no kernel matching {needles}"
    return hits


def test_fused_multiscale_kernel_stays_on_its_occupancy_step(reports):
    # ms_flip_reduce_kernel<NV = 2, INNER, OUTER, ALIGN, TH = 32>: the cfg5 instances (mean / mean = <2,0,0,*,32>, gmean / gmean = <2,2,2,*,32>)
    for k, r in _find(reports["resample"], "ms_flip_reduce_kernelILi2ELi0ELi0E", "ELi32ELi64EE").items():
        assert r["VGPRs"] <= 80 and r["ScratchSize"] == 0 and r["Occupancy"] >= 6, (k, r)
    for k, r in _find(reports["resample"], "ms_flip_reduce_kernelILi2ELi2ELi2E", "ELi32ELi64EE").items():
        assert r["VGPRs"] <= 80 and r["ScratchSize"] <= 16 and r["Occupancy"] >= 6, (k, r)     # (align_corners = 1 spills 2 registers to stay there)
    # the 128 x 16 tiles that run by default (ptb_set_tunable(15, 128)): the same cliff
    for k, r in _find(reports["resample"], "ms_flip_reduce_kernelILi2ELi0ELi0E", "ELi16ELi128EE").items():
        assert r["VGPRs"] <= 80 and r["ScratchSize"] == 0 and r["Occupancy"] >= 6, (k, r)
    for k, r in _find(reports["resample"], "ms_flip_reduce_kernelILi2ELi2ELi2E", "ELi16ELi128EE").items():
        assert r["VGPRs"] <= 80 and r["ScratchSize"] <= 16 and r["Occupancy"] >= 6, (k, r)
    for k, r in _find(reports["resample"], "ms_flip_reduce_kernel").items():
        assert r["LDS Size"] <= 41 * 1024, (k, r)      # 64 x 64 tiles: 40 KB, 4 workgroups per CU


def test_band_plan_kernel_has_no_scratch(reports):
    hits = _find(reports["bandplan"], "band_plan_kernel")
    for k, r in hits.items():
        assert r["ScratchSize"] == 0 and r.get("VGPRs Spill", 0) == 0, (k, r)
    d4_mean = _find(hits, "ILi8ELi6166440ELi0ELi1E")          # 8 views, D4 codes, linear reduction, fp32 source
    plain = {k: r for k, r in d4_mean.items() if "Lb0EEEv" in k}
    prefetching = {k: r for k, r in d4_mean.items() if "Lb1EEEv" in k}       # the headline instances (ptb_set_tunable key 21, default 2)
    assert len(plain) == 2 and len(prefetching) == 2, sorted(d4_mean)
    for k, r in plain.items():
        assert r["VGPRs"] <= 64 and r["Occupancy"] >= 8, (k, r)     # 8 waves per SIMD: two 1024-thread (four 512-thread) workgroups per CU
    for k, r in prefetching.items():
        # one 1024-thread workgroup per CU (4 waves per SIMD: <= 128 registers) with the next tile in flight; round 6: the 64-row instance keeps TWO
        # sets of LDS tiles (128 KiB: one barrier per covering tile), so the compiler's occupancy figure is LDS-bound at 4
        # (+ the next tile's window values travelling with its views: 118 of the 128 registers a 1024-thread workgroup may use)
        assert r["VGPRs"] <= 120 and r["Occupancy"] >= 4 and r["LDS Size"] <= 128 * 1024, (k, r)
    for k, r in _find(hits, "ILi8ELi6166440ELi0ELi2E").items():    # fp16 source
        if "Lb1EEEv" in k:
            assert r["VGPRs"] <= 104, (k, r)


def test_straight_line_loss_kernels_do_not_spill(reports):
    fused = _find(reports["losses"], "seg_fwd_lean_kernelILi16ELi0ELb1ELb0ELi2ELb0ELb1ELb0E")    # cfg4 fused forward, C = 16 (ptb_set_tunable(12, 0))
    for k, r in fused.items():
        assert r["VGPRs"] <= 128 and r["ScratchSize"] == 0, (k, r)      # (the in-launch tail is inlined: a call would cost a stack and 20 VGPRs)
    for k, r in _find(reports["losses"], "seg_focal_pk_kernelILi16ELb0ELb1ELb0E").items():          # the packed-fp32 instance that runs by default
        assert r["VGPRs"] <= 128 and r["ScratchSize"] == 0 and r["Occupancy"] >= 4, (k, r)
    for k, r in _find(reports["losses"], "softmax_focal_bwd_kernelILi4ELi16ELb1E").items():
        assert r["ScratchSize"] == 0 and r.get("VGPRs Spill", 0) == 0, (k, r)
    for k, r in _find(reports["losses"], "seg_fused_bwd_lean_kernelILi16E").items():
        assert r["ScratchSize"] == 0, (k, r)
    for k, r in _find(reports["losses"], "seg_stats_bwd_kernelILi4ELi16ELb0E").items():      # Dice / Jaccard backward at cfg4
        assert r["VGPRs"] <= 96 and r["ScratchSize"] == 0 and r["Occupancy"] >= 5, (k, r)
