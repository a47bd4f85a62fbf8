"""-m "not gpu": the update kernels' register / scratch / occupancy budgets, as the compiler reports them
(-Rpass-analysis=kernel-resource-usage, written by __graft_entry__.build() to khronos_amd/lib/resource_usage.txt).  A change that
pushes the dominant kernel into scratch memory or below its occupancy is a performance regression that no parity test sees
("shown by the resource-usage guard test")."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "khronos_amd", "lib", "resource_usage.txt")


def _kernels():
    if not os.path.exists(PATH):
        pytest.skip("khronos_amd/lib/resource_usage.txt is written by __graft_entry__.build() when it compiles the HIP library")
    out, cur = {}, None
    for ln in open(PATH):
        m = re.search(r"remark: Function Name: (\S+)", ln)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", ln)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return out


def _pick(kernels, pattern):
    sel = {k: v for k, v in kernels.items() if re.search(pattern, k)}
    assert sel, pattern
    return sel


def test_k_fuse_budget():
    """the dominant kernel (k_fuse<16, 4 | 8, .., 12 waves>): at most 168 VGPRs = 3 waves per SIMD = one 12-wave workgroup per CU, no
    scratch memory; its 61.5 KB of LDS fit beside nothing else of its own, as designed"""
    k = _kernels()
    for name, r in _pick(k, r"^_ZN3khr6k_fuseILi16ELi[48]ELb[01]ELb[01]ELi12ELb0E").items():
        assert r["VGPRs"] <= 168 and r["Occupancy"] >= 3, (name, r)
        assert r["ScratchSize"] == 0 and r["VGPRs Spill"] == 0, (name, r)
        assert r["LDS Size"] <= 64 * 1024, (name, r)


def test_no_update_kernel_uses_scratch_memory():
    """every instantiation of k_tsdf / k_band5 / k_fuse / k_fuse2 keeps its working set in registers (scalar spills to vector lanes are
    fine: v_writelane / v_readlane, no memory).  Measured in round 6: the same k_tsdf with 21 VGPRs in scratch ran 65 us instead of 44."""
    k = _kernels()
    for name, r in _pick(k, r"^_ZN3khr(6k_tsdfI|7k_band5I|6k_fuseI|7k_fuse2I)").items():
        if "Lb1EEEvNS_8FuseArgs" in name and name.startswith("_ZN3khr6k_fuseI") and name.endswith("ELb1EEEvNS_8FuseArgsENS_8FuseListE"):
            continue  # (the DBG instantiation with the in-kernel timeline probe)
        assert r["ScratchSize"] == 0 and r["VGPRs Spill"] == 0, (name, r)


def test_k_tsdf_and_k_band5_keep_their_occupancy():
    """the default update step (khr_kernels_fuse5.h): the voxel kernel at >= 5 waves per SIMD (<= 96 VGPRs), 7 for the 64 x 2 items of
    small frames; the band kernel at 5"""
    k = _kernels()
    for name, r in _pick(k, r"^_ZN3khr6k_tsdfILi4ELb[01]ELi8ELi5ELi0E").items():
        assert r["Occupancy"] >= 5 and r["VGPRs"] <= 96, (name, r)
    for name, r in _pick(k, r"^_ZN3khr6k_tsdfILi8E").items():
        assert r["Occupancy"] >= 6, (name, r)
    for name, r in _pick(k, r"^_ZN3khr7k_band5I").items():
        assert r["Occupancy"] >= 5, (name, r)


def test_k_fuse2_multi_frame_budget():
    """the object maps' / rig ticks' multi-frame update: 4 waves per SIMD (128 VGPRs)"""
    k = _kernels()
    for name, r in _pick(k, r"^_ZN3khr7k_fuse2I.*ELi4ELb1EEEv").items():
        assert r["VGPRs"] <= 128 and r["Occupancy"] >= 4, (name, r)


def test_marching_cubes_lds():
    """count pass <= 26 KB (6 workgroups per CU), emit pass <= 43 KB (3 per CU)"""
    k = _kernels()
    for name, r in _pick(k, r"^_ZN3khr16k_marching_cubesILi16ELb0E").items():
        assert r["LDS Size"] <= 27 * 1024, (name, r)
    for name, r in _pick(k, r"^_ZN3khr16k_marching_cubesILi16ELb1E").items():
        assert r["LDS Size"] <= 43 * 1024, (name, r)
