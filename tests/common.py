"""Shared helpers for the parity tests: one config -> (HIP context, oracle map), frame feeding, and
block-by-block comparison."""
import numpy as np

from khronos_amd import FusionContext, default_config
from khronos_amd.synth import SyntheticStream
from oracle import pyoracle as po

# the float tolerance BASELINE.json states (per-voxel SDF / weight / label within 1e-4)
TOL = 1e-4
# arithmetic mode of the contexts make_pair creates (khr_config.exact_arithmetic); the `arith` fixture (conftest.py) runs a
# test under both.  1: voxel values bit-identical to the oracle (every comparison below is exact).  0 (the relaxed option; the product default is 1):
# every decision of the integrator is still exact, distance / weight carry ~1e-6 relative error, so the few later decisions
# that read those VALUES (occupied = distance < threshold, colour rounding) may differ on voxels that sit on the
# threshold to within that error; such voxels are counted and bounded, not ignored.
EXACT = 1


def make_pair(width=320, height=240, seed=1234, stream_kw=None, **cfg_kw):
    kw = dict(voxel_size=0.1, truncation_distance=0.3, with_semantics=1, with_tracking=1, max_blocks=4096,
              max_frame_pixels=width * height, md_min_cluster_size=20, md_min_separation_distance=2.0, md_max_range=5.0)
    kw.update(cfg_kw)
    kw.setdefault("exact_arithmetic", EXACT)
    cfg = default_config(**kw)
    ctx = FusionContext(cfg)
    ora = po.OracleMap(po.config_from(cfg, 0))
    s = SyntheticStream(width, height, seed=seed, **(stream_kw or {}))
    sen = ctx.make_sensor(width, height, s.fx, s.fy, s.cx, s.cy)
    osen = ora.make_sensor(width, height, s.fx, s.fy, s.cx, s.cy)
    return cfg, ctx, ora, s, sen, osen


def step_both(ctx, ora, sen, osen, fr, motion=False, track=True, use_color=True, use_label=True):
    """One ActiveWindow::spinOnce worth of volumetric work (active_window.cpp:127,203-215) on both."""
    color = fr["rgb"] if use_color else None
    label = fr["label"] if use_label else None
    slot = ctx.upload_frame(sen, fr["stamp"], fr["pose"], fr["depth"], color, label)
    dyn_o = None
    out = {}
    if motion:
        out["n_gpu"] = ctx.detect_motion(slot)
        out["n_ora"], dyn_o, out["seeds_ora"] = ora.detect_motion(osen, fr["stamp"], fr["pose"], fr["depth"])
        out["dyn_ora"] = dyn_o
        out["dyn_gpu"] = ctx.download_frame(slot, fr["depth"].shape, range_image=False, dynamic_image=True)[2]
    ctx.integrate(slot, allocate_blocks=True, use_mask=motion)
    out["ostats"] = ora.integrate(osen, fr["stamp"], fr["pose"], fr["depth"], color, label, mask=dyn_o)
    if track:
        ctx.update_tracking(fr["stamp"])
        ora.update_tracking(fr["stamp"])
    out["slot"] = slot
    return out


def occupancy_threshold(cfg):
    """tracking_integrator.cpp:136-138: negative = multiple of the voxel size"""
    t = cfg.tsdf_occupancy_threshold
    return -t * cfg.voxel_size if t < 0 else t


def compare_maps(ctx, ora, max_blocks=None, rng=None, check_lik=True, cfg=None, exact=None):
    """Block index sets bit-exact; per-voxel fields within TOL; labels and last_observed exact.  exact arithmetic: every
    field bit-identical.  Fast arithmetic: last_occupied / tracking flags derive from `distance < threshold`, a decision
    on a VALUE that carries ~1e-7 of error, so a voxel sitting on the threshold may differ; mismatches are counted in
    worst["borderline"] and bounded by 1e-5 of the compared voxels (expected: 0 at test sizes)."""
    exact = bool(EXACT) if exact is None else exact
    gi, oi = ctx.block_indices(), ora.block_indices()
    assert gi.shape == oi.shape, (gi.shape, oi.shape)
    assert (gi == oi).all(), "block index sets differ"
    sel = np.arange(len(gi))
    if max_blocks is not None and len(gi) > max_blocks:
        rng = rng or np.random.default_rng(0)
        sel = np.sort(rng.choice(len(gi), max_blocks, replace=False))
    worst = {"distance": 0.0, "weight_rel": 0.0, "lik": 0.0, "color": 0, "borderline": 0, "voxels": 0, "color_off_by_one": 0}
    for i in sel:
        idx = gi[i]
        g, o = ctx.download_block(idx), ora.get_block(idx)
        worst["voxels"] += g["distance"].size
        worst["distance"] = max(worst["distance"], float(np.abs(g["distance"] - o["distance"]).max()))
        wr = np.abs(g["weight"] - o["weight"]) / np.maximum(1.0, np.abs(o["weight"]))
        worst["weight_rel"] = max(worst["weight_rel"], float(wr.max()))
        cd = np.abs(g["color"].astype(int) - o["color"].astype(int))
        worst["color"] = max(worst["color"], int(cd.max()))
        worst["color_off_by_one"] += int((cd.max(axis=-1) > 0).sum()) if cd.ndim > 1 else int((cd > 0).sum())
        assert (g["last_observed"] == o["last_observed"]).all(), ("last_observed", idx)
        assert (g["sem_label"] == o["sem_label"]).all(), ("sem_label", idx)
        if exact:
            assert np.array_equal(g["distance"], o["distance"]), ("distance not bit-exact", idx)
            assert np.array_equal(g["weight"], o["weight"]), ("weight not bit-exact", idx)
            assert (g["color"] == o["color"]).all(), ("color", idx)
            assert (g["last_occupied"] == o["last_occupied"]).all(), ("last_occupied", idx)
            assert (g["flags"] == o["flags"]).all(), ("flags", idx, np.flatnonzero(g["flags"] != o["flags"])[:8])
            assert g["block_flags"] == o["block_flags"], ("block_flags", idx, g["block_flags"], o["block_flags"])
        else:
            bad = (g["last_occupied"] != o["last_occupied"]) | (g["flags"] != o["flags"])
            worst["borderline"] += int(bad.sum())
            if g["block_flags"] != o["block_flags"]:
                worst["borderline"] += 1
        if check_lik and g["likelihoods"] is not None:
            worst["lik"] = max(worst["lik"], float(np.abs(g["likelihoods"] - o["likelihoods"]).max()))
    assert worst["distance"] <= TOL, worst
    assert worst["weight_rel"] <= TOL, worst
    assert worst["lik"] <= TOL * 10, worst  # log-likelihood sums grow with observations
    assert worst["color"] <= 1, worst
    assert worst["borderline"] <= max(2, int(1e-5 * worst["voxels"])), worst
    # ALL blocks, every voxel, every layer: one digest per layer on both sides (the per-block loop above only samples large
    # maps -- it is there for readable failures and for the tolerance checks of the relaxed arithmetic)
    if hasattr(ctx, "map_digest") and hasattr(ora, "map_digest"):
        assert_digests_equal(ctx.map_digest(), ora.map_digest(), exact=exact, what="HIP vs oracle")
    return worst, len(gi)


DIGEST_LAYERS = ("distance", "weight", "color", "last_observed", "last_occupied", "flags", "sem_label", "likelihoods", "block_flags",
                 "index", "n_blocks", "reserved")
# layers that are bit-exact even with the relaxed arithmetic (decisions, never values)
DIGEST_DECISION_LAYERS = ("last_observed", "sem_label", "index", "n_blocks", "reserved")


def assert_digests_equal(a, b, exact=True, what=""):
    """whole-map parity: khr_map_digest / orc_map_digest words, every layer (exact arithmetic) or the decision layers"""
    for i, name in enumerate(DIGEST_LAYERS):
        if exact or name in DIGEST_DECISION_LAYERS:
            assert int(a[i]) == int(b[i]), ("whole-map digest differs", what, name, hex(int(a[i])), hex(int(b[i])))


def _mix64(x):
    x = x + np.uint64(0x9E3779B97F4A7C15)
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def np_map_digest(blocks, with_semantics=True):
    """the digest definition of include/khronos_amd.h (khr_map_digest) restated in numpy over per-block downloads
    (`blocks`: iterable of (index, dict as returned by download_block / get_block)): the third, independent implementation the
    device kernel and the oracle's C++ are checked against at small sizes."""
    out = np.zeros(12, np.uint64)
    G, L = np.uint64(0x9E3779B97F4A7C15), np.uint64(0x632BE59BD9B4E019)
    with np.errstate(over="ignore"):
        for idx, b in blocks:
            x, y, z = (int(v) for v in idx)
            key = np.uint64(((x + (1 << 20)) & 0x1FFFFF) | (((y + (1 << 20)) & 0x1FFFFF) << 21) | (((z + (1 << 20)) & 0x1FFFFF) << 42))
            nv = b["distance"].size
            i = np.arange(nv, dtype=np.uint64)

            def term(layer, index, value):
                return _mix64(_mix64(key * G + np.uint64(layer) * L + index) ^ value.astype(np.uint64))
            out[0] += term(0, i, b["distance"].view(np.uint32)).sum(dtype=np.uint64)
            out[1] += term(1, i, b["weight"].view(np.uint32)).sum(dtype=np.uint64)
            out[2] += term(2, i, np.ascontiguousarray(b["color"]).view(np.uint32).reshape(-1)).sum(dtype=np.uint64)
            out[3] += term(3, i, b["last_observed"]).sum(dtype=np.uint64)
            out[4] += term(4, i, b["last_occupied"]).sum(dtype=np.uint64)
            out[5] += term(5, i, b["flags"]).sum(dtype=np.uint64)
            out[6] += term(6, i, b["sem_label"]).sum(dtype=np.uint64)
            if with_semantics and b.get("likelihoods") is not None:
                lik = np.ascontiguousarray(b["likelihoods"], dtype=np.float32)  # [k][voxel]
                K = lik.shape[0]
                out[7] += term(7, np.arange(K * nv, dtype=np.uint64), lik.reshape(-1).view(np.uint32)).sum(dtype=np.uint64)
            out[8] += term(8, np.zeros(1, np.uint64), np.array([b["block_flags"] & 0xF], np.uint64))[0]
            out[9] += _mix64(np.array([key], np.uint64))[0]
            out[10] += np.uint64(1)
    return out


class DeviceArray:
    """a numpy array copied to HBM through the HIP runtime the library itself is linked against (ctypes); for tests of the
    device-pointer entry points that must not pull a second HIP runtime (torch's) into the process."""
    _hip = None

    def __init__(self, arr):
        import ctypes as C
        if DeviceArray._hip is None:
            DeviceArray._hip = C.CDLL("libamdhip64.so")
        arr = np.ascontiguousarray(arr)
        self.ptr = C.c_void_p()
        rc = self._hip.hipMalloc(C.byref(self.ptr), C.c_size_t(arr.nbytes))
        assert rc == 0, rc
        rc = self._hip.hipMemcpy(self.ptr, C.c_void_p(arr.ctypes.data), C.c_size_t(arr.nbytes), 1)
        assert rc == 0, rc

    def data_ptr(self):
        return self.ptr.value

    def read(self, offset_bytes, nbytes):
        """bytes [offset, offset + n) back on the host (blocking)"""
        import ctypes as C
        out = np.empty(nbytes, np.uint8)
        if nbytes:
            rc = self._hip.hipMemcpy(C.c_void_p(out.ctypes.data), C.c_void_p(self.ptr.value + offset_bytes), C.c_size_t(nbytes), 2)
            assert rc == 0, rc
        return out

    def copy_from(self, dst_offset_bytes, src, src_offset_bytes, nbytes):
        """device-to-device copy from another DeviceArray (blocking)"""
        import ctypes as C
        if nbytes:
            rc = self._hip.hipMemcpy(C.c_void_p(self.ptr.value + dst_offset_bytes), C.c_void_p(src.ptr.value + src_offset_bytes), C.c_size_t(nbytes), 3)
            assert rc == 0, rc

    def free(self):
        if self.ptr:
            self._hip.hipFree(self.ptr)
            self.ptr = None


class PinnedArray:
    """a numpy array copied into page-locked host memory (hipHostMalloc through the HIP runtime the library is linked against)"""

    def __init__(self, arr):
        import ctypes as C
        if DeviceArray._hip is None:
            DeviceArray._hip = C.CDLL("libamdhip64.so")
        arr = np.ascontiguousarray(arr)
        self.ptr = C.c_void_p()
        rc = DeviceArray._hip.hipHostMalloc(C.byref(self.ptr), C.c_size_t(arr.nbytes), C.c_uint(0))
        assert rc == 0, rc
        C.memmove(self.ptr, arr.ctypes.data, arr.nbytes)

    def data_ptr(self):
        return self.ptr.value

    def free(self):
        if self.ptr:
            DeviceArray._hip.hipHostFree(self.ptr)
            self.ptr = None
