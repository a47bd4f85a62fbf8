"""GPU parity for the ray verificator: khr_rv_* against the oracle restatement of
khronos::RayVerificator, bit-exact (counts and timestamp lists in ascending ray order)."""
import numpy as np
import pytest

from khronos_amd import RayVerificator
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
T = 1_000_000_000


def _scene(rng, n_poses, n_per_pose):
    """sensor positions on a circle inside an 8 x 6 x 3 m room; every pose sees points on the walls / floor."""
    stamps, src, tgt = [], [], []
    for k in range(n_poses):
        th = 2 * np.pi * k / n_poses
        s = np.array([1.5 * np.cos(th), 1.5 * np.sin(th), 1.5], np.float32)
        d = rng.normal(size=(n_per_pose, 3)).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        # distance to the room box [-4,4] x [-3,3] x [0,3]
        lo, hi = np.array([-4, -3, 0], np.float32), np.array([4, 3, 3], np.float32)
        with np.errstate(divide="ignore"):
            t = np.where(d > 0, (hi - s) / d, (lo - s) / d)
        dist = np.minimum(t.min(1), 5.0).astype(np.float32)
        p = (s + d * dist[:, None]).astype(np.float32)
        stamps += [np.full(n_per_pose, (1 + k) * T, np.uint64)]
        src += [np.repeat(s[None], n_per_pose, 0)]
        tgt += [p]
    return np.concatenate(stamps), np.concatenate(src), np.concatenate(tgt)


def _compare(dev, ora, pts, t0, t1):
    g = dev.check(pts, t0, t1)
    o = ora.check(pts, t0, t1)
    for a, b, name in zip(g, o, ("n_present", "n_absent", "present", "absent")):
        assert a.shape == b.shape and np.array_equal(a, b), name
    return g


@pytest.mark.parametrize("block_size,radial,depth", [(1.0, 0.1, 0.1), (0.5, 0.05, 0.2), (2.0, 0.3, 0.05)])
def test_rayver_parity(block_size, radial, depth):
    rng = np.random.default_rng(11)
    st, src, tgt = _scene(rng, 12, 400)
    dev, ora = RayVerificator(block_size, radial, depth), po.OracleRayVerificator(block_size, radial, depth)
    # two batches: the index is extended, not rebuilt from scratch by the caller
    half = len(st) // 2
    for sl in (slice(0, half), slice(half, None)):
        dev.add_rays(st[sl], src[sl], tgt[sl])
        ora.add_rays(st[sl], src[sl], tgt[sl])
        assert dev.num_pairs() == ora.num_pairs() > 0
    assert dev.num_rays() == len(st)
    # queries: measured surface points (present by their own ray, absent / occluded for others), points half-way along
    # rays (absent), random points, points outside everything
    sel = rng.choice(len(st), 300, replace=False)
    pts = np.concatenate([tgt[sel], 0.5 * (src[sel] + tgt[sel]), rng.uniform([-4, -3, 0], [4, 3, 3], (300, 3)).astype(np.float32),
                          np.array([[50.0, 50.0, 50.0], [-7.5, 0.0, 1.0]], np.float32)]).astype(np.float32)
    g = _compare(dev, ora, pts, 0, 2 ** 64 - 1)
    assert g[0].sum() > 300 and g[1].sum() > 300
    # per-point time windows (ray_background_change_detector.cpp:92 / ray_object_change_detector.cpp:127-134)
    t0 = rng.integers(0, 8, len(pts)).astype(np.uint64) * T
    t1 = t0 + rng.integers(0, 6, len(pts)).astype(np.uint64) * T
    _compare(dev, ora, pts, t0, t1)
    # setDsg(): start over
    dev.clear()
    assert dev.num_rays() == 0 and dev.num_pairs() == 0
    n_p, n_a, pres, absn = dev.check(pts[:10], 0, 2 ** 64 - 1)
    assert n_p.sum() == 0 and n_a.sum() == 0 and len(pres) == 0 and len(absn) == 0


def test_rayver_edge_cases():
    dev, ora = RayVerificator(1.0, 0.1, 0.1), po.OracleRayVerificator(1.0, 0.1, 0.1)
    assert dev.check(np.zeros((3, 3), np.float32), 0, 1)[0].sum() == 0  # no rays yet
    # zero-length ray (left out of the index), axis-parallel rays on block borders, negative coordinates, many rays in one block
    st = np.arange(1, 203, dtype=np.uint64) * T
    src = np.zeros((202, 3), np.float32)
    tgt = np.zeros((202, 3), np.float32)
    tgt[1] = [3.0, 0.0, 0.0]
    tgt[2:] = np.stack([np.full(200, -2.0), np.linspace(-0.4, 0.4, 200), np.full(200, -1.0)], 1)
    dev.add_rays(st, src, tgt)
    ora.add_rays(st, src, tgt)
    assert dev.num_pairs() == ora.num_pairs()
    pts = np.concatenate([tgt, 0.5 * tgt, [[1.0, 0.0, 0.0], [2.0, 0.0, 0.0], [-1.0, 0.0, -0.5]]]).astype(np.float32)
    g = _compare(dev, ora, pts, 0, 2 ** 64 - 1)
    assert g[0].max() > 5  # neighbouring rays of the fan agree on their end points
    with pytest.raises(Exception):
        RayVerificator(0.0, 0.1, 0.1)
    with pytest.raises(Exception):
        RayVerificator(1.0, 0.1, -1.0)


@pytest.mark.parametrize("relative", [True, False])
def test_device_vote_equals_restatement(relative):
    """khr_rv_detect_changes (RayChangeDetector::detectChanges on the device, one wave per point) against the oracle
    restatement applied to the stamp lists of khr_rv_check_stamps, point by point: both search directions, several
    resolutions and windows, relative and count confidences; a span of more than 2048 bins raises the host-fallback flag."""
    rng = np.random.default_rng(5)
    st, src, tgt = _scene(rng, 40, 300)  # stamps 1 .. 40 s
    dev = RayVerificator(1.0, 0.1, 0.1)
    dev.add_rays(st, src, tgt)
    sel = rng.choice(len(st), 200, replace=False)
    pts = np.concatenate([tgt[sel], 0.5 * (src[sel] + tgt[sel]), rng.uniform([-4, -3, 0], [4, 3, 3], (200, 3)).astype(np.float32),
                          np.array([[50.0, 50.0, 50.0]], np.float32)]).astype(np.float32)
    m = len(pts)
    n_p, n_a, pres, absn = dev.check(pts, 0, 2 ** 64 - 1)
    op = np.concatenate([[0], np.cumsum(n_p.astype(np.int64))]).astype(np.int64)
    oa = np.concatenate([[0], np.cumsum(n_a.astype(np.int64))]).astype(np.int64)
    fwd = rng.integers(0, 2, m).astype(bool)
    some = 0
    for res, window in ((1.0, 5), (0.5, 3), (2.5, 1), (0.1, 8)):
        kw = dict(temporal_resolution=res, window_size=window)
        if relative:
            kw.update(absence_confidence=0.4, presence_confidence=0.55)
        else:
            kw.update(use_relative_confidence=False, absence_confidence=2.0, presence_confidence=3.0)
        for direction in (fwd, True, False):
            ca, fp, fl = dev.check_and_vote(pts, 0, 2 ** 64 - 1, direction, **kw)
            assert not (fl & 0x80).any()
            for i in range(m):
                f = bool(direction[i]) if not isinstance(direction, bool) else direction
                ref = po.detect_changes(pres[op[i]:op[i + 1]], absn[oa[i]:oa[i + 1]], f, **kw)
                got = (int(ca[i]) if fl[i] & 1 else None, int(fp[i]) if fl[i] & 2 else None)
                assert got == ref, (i, f, kw, got, ref)
                some += ref[0] is not None or ref[1] is not None
    assert some > 500
    # 1 ms bins over 39 s of observations: more bins than the histogram holds -> flag, no result
    ca, fp, fl = dev.check_and_vote(pts, 0, 2 ** 64 - 1, True, temporal_resolution=0.001)
    multi = (n_p + n_a) > 1
    assert (fl[multi] & 0x80).any() and not (fl[~(fl & 0x80).astype(bool)] & 0x80).any()


def test_background_and_object_change_detectors_equal_restatement():
    """the two callers of the ray verificator (BASELINE configs[4] "change reconciliation"): RayBackgroundChangeDetector::detectChanges
    (ray_background_change_detector.cpp:59-103: every new and every re-observed mesh vertex, check + forward vote) and
    RayObjectChangeDetector::checkObjectObservation (ray_object_change_detector.cpp:117-160: sub-sampled vertices queried before /
    after the object's life time, merged, voted once per direction), as batched device passes of the host mirrors
    (khronos_amd/host/change_detection.cpp), against the per-vertex loops of the reference restated with the oracle."""
    from khronos_amd import host_capi as hc
    rng = np.random.default_rng(21)
    st, src, tgt = _scene(rng, 40, 300)  # stamps 1 .. 40 s
    dev, ora = RayVerificator(1.0, 0.1, 0.1), po.OracleRayVerificator(1.0, 0.1, 0.1)
    dev.add_rays(st, src, tgt)
    ora.add_rays(st, src, tgt)
    kw = dict(temporal_resolution=1.0, window_size=5, use_relative_confidence=True, absence_confidence=0.4, presence_confidence=0.55)
    thr = 5.0
    thr_ns = int(np.float32(thr) * 1e9)  # uint64(config.time_filtering_threshold * 1e9)

    def state(point, earliest):  # checkVertex (:90-103)
        _, _, pres, absn = ora.check(point[None], earliest, 2 ** 64 - 1)
        ca, fp = po.detect_changes(pres, absn, True, **kw)
        return hc.ABSENT if ca is not None else (hc.PERSISTENT if fp is not None else hc.UNOBSERVED)
    # background mesh: measured surface points (present), points in front of surfaces (seen through: absent), unobserved ones
    sel = rng.choice(len(st), 400, replace=False)
    verts = np.concatenate([tgt[sel[:200]], 0.5 * (src[sel[200:]] + tgt[sel[200:]]), np.full((5, 3), 60.0, np.float32)]).astype(np.float32)
    vstamps = np.concatenate([st[sel[:200]], st[sel[200:]], np.full(5, 3 * T, np.uint64)])
    first = 250
    states, _ = hc.background_changes(dev, verts[:first], vstamps[:first], time_filtering_threshold=thr, **kw)
    want = np.array([state(verts[i], int(vstamps[i]) + thr_ns) for i in range(first)], np.uint8)
    assert np.array_equal(states, want)
    assert (want == hc.ABSENT).sum() > 10 and (want == hc.PERSISTENT).sum() > 10 and (want == hc.UNOBSERVED).sum() > 10
    # more rays arrive, the mesh grows, some old vertices were re-observed: new ones get their first state, re-observed ones are recomputed
    st2, src2, tgt2 = _scene(np.random.default_rng(22), 12, 300)
    st2 = st2 + np.uint64(40 * T)
    dev.add_rays(st2, src2, tgt2)
    ora.add_rays(st2, src2, tgt2)
    reobs = [3, 17, 120, 249, 100000]  # (an index beyond the mesh is ignored, :73-75)
    states2, n_changed = hc.background_changes(dev, verts, vstamps, states=states, reobserved=reobs, time_filtering_threshold=thr, **kw)
    want2 = want.copy()
    for i in reobs[:-1]:
        want2[i] = state(verts[i], int(vstamps[i]) + thr_ns)
    want2 = np.concatenate([want2, [state(verts[i], int(vstamps[i]) + thr_ns) for i in range(first, len(verts))]]).astype(np.uint8)
    assert np.array_equal(states2, want2)
    assert n_changed == sum(int(want2[i] != want[i]) for i in reobs[:-1])
    # an object: a blob of surface points seen between 10 s and 20 s, stored in its bounding-box frame
    centre = tgt[sel[0]]
    blob = (centre + rng.normal(scale=0.15, size=(1000, 3))).astype(np.float32)
    b0, b1 = blob.min(0), blob.max(0)
    local = (blob - (np.float32(0.5) * (b0 + b1)).astype(np.float32)).astype(np.float32)
    t_first, t_last, sub = 10 * T, 20 * T, 7
    got = hc.object_change(dev, local, b0, b1, t_first, t_last, time_filtering_threshold=thr, query_subsampling=sub, **kw)
    before = [[], []]
    after = [[], []]
    ctr = np.array([np.float32(0.5) * (b0[d] + b1[d]) for d in range(3)], np.float32)  # BoundingBox::center: 0.5f * (min + max)
    for i in range(0, len(local), sub):
        p = (local[i] + ctr).astype(np.float32)
        _, _, pres, absn = ora.check(p[None], 0, t_first - thr_ns)
        before[0] += list(pres)
        before[1] += list(absn)
        _, _, pres, absn = ora.check(p[None], t_last + thr_ns, 2 ** 64 - 1)
        after[0] += list(pres)
        after[1] += list(absn)
    bca, bfp = po.detect_changes(before[0], before[1], False, **kw)
    aca, afp = po.detect_changes(after[0], after[1], True, **kw)
    assert got == dict(first_absent=bca or 0, last_absent=aca or 0, first_persistent=bfp or 0, last_persistent=afp or 0)
    assert len(before[0]) + len(before[1]) > 20 and len(after[0]) + len(after[1]) > 20
