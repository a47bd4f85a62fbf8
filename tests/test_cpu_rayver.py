"""Oracle known-answer tests for khronos::RayVerificator (ray_verificator.cpp:66-145, 327-349)."""
import numpy as np
import pytest

from oracle import pyoracle as po

T = 1_000_000_000


def _one_ray(block_size=1.0, radial=0.1, depth=0.1):
    rv = po.OracleRayVerificator(block_size, radial, depth)
    # sensor at the origin sees a surface point 4 m down the x axis at t = 5 s
    rv.add_rays([5 * T], [[0.0, 0.5, 0.5]], [[4.0, 0.5, 0.5]])
    return rv


def test_march_marks_every_block_on_the_ray_once():
    rv = _one_ray()
    # samples every 0.25 m from 0.25 m to the first one beyond 4 m (4.25): blocks x = 0..4, each listed once
    assert rv.num_pairs() == 5
    rv2 = po.OracleRayVerificator(0.5, 0.1, 0.1)
    rv2.add_rays([5 * T], [[0.0, 0.25, 0.25]], [[4.0, 0.25, 0.25]])
    assert rv2.num_pairs() == 9  # blocks 0..8 of a 0.5 m grid (4.125 is still in block 8)


def test_check_classifies_present_absent_occluded_and_misses():
    rv = _one_ray()
    pres, absn = rv.check_one([4.0, 0.5, 0.5])                 # the measured surface point itself
    assert list(pres) == [5 * T] and len(absn) == 0
    pres, absn = rv.check_one([4.05, 0.5, 0.5])                # within the depth tolerance
    assert list(pres) == [5 * T] and len(absn) == 0
    pres, absn = rv.check_one([2.0, 0.5, 0.5])                 # the ray passed through: evidence of absence
    assert len(pres) == 0 and list(absn) == [5 * T]
    pres, absn = rv.check_one([2.0, 0.55, 0.5])                # 5 cm off the ray: inside the radial tolerance
    assert list(absn) == [5 * T]
    pres, absn = rv.check_one([2.0, 0.7, 0.5])                 # 20 cm off the ray: no overlap
    assert len(pres) == 0 and len(absn) == 0
    pres, absn = rv.check_one([4.2, 0.5, 0.5])                 # behind the surface (and in a marched block): occluded
    assert len(pres) == 0 and len(absn) == 0
    pres, absn = rv.check_one([2.0, 0.5, 1.5])                 # a block no ray crossed
    assert len(pres) == 0 and len(absn) == 0


def test_time_window_is_inclusive():
    rv = _one_ray()
    p = [2.0, 0.5, 0.5]
    assert len(rv.check_one(p, 0, 5 * T)[1]) == 1 and len(rv.check_one(p, 5 * T, 6 * T)[1]) == 1
    assert len(rv.check_one(p, 0, 5 * T - 1)[1]) == 0 and len(rv.check_one(p, 5 * T + 1, 9 * T)[1]) == 0


def test_results_come_in_ray_order_and_rays_accumulate():
    rv = po.OracleRayVerificator(1.0, 0.1, 0.1)
    src = [[0.0, 0.5, 0.5]] * 3
    rv.add_rays([7 * T, 3 * T], src[:2], [[4.0, 0.5, 0.5], [1.5, 0.5, 0.5]])
    rv.add_rays([9 * T], src[:1], [[2.0, 0.5, 0.5]])
    pres, absn = rv.check_one([2.0, 0.5, 0.5])
    # ray 0 sees through the point, ray 1 stops 0.5 m short of it (occlusion), ray 2 ends on it
    assert list(absn) == [7 * T] and list(pres) == [9 * T]
    pres, absn = rv.check_one([1.5, 0.5, 0.5])
    assert list(absn) == [7 * T, 9 * T] and list(pres) == [3 * T]  # ascending ray index, not ascending time


# ---- RayChangeDetector::detectChanges (time-bin vote over the check results; pure host logic) -----------------------------
S = 1_000_000_000


def _both(present, absent, forward, **kw):
    from khronos_amd.host_capi import detect_changes
    got = detect_changes(present, absent, forward, **kw)
    ref = po.detect_changes(present, absent, forward, **kw)
    assert got == ref, (got, ref, kw)
    return got


def test_change_detector_known_answers():
    """hand-derived cases of ray_change_detector.cpp:66-133 (defaults: 1 s bins, window 5, relative confidences 0.5)."""
    # nothing observed: neither result exists
    assert _both([], [], True) == (None, None)
    # only presence, three bins: the LAST bin visited in the search direction is the furthest persistent one
    assert _both([1 * S, 2 * S + 5, 9 * S], [], True) == (None, 9 * S)
    assert _both([1 * S, 2 * S + 5, 9 * S], [], False) == (None, 1 * S)
    # an absent majority stops the search; what was persistent before it stays
    assert _both([1 * S, 1 * S + 7], [20 * S, 20 * S + 1, 21 * S], True) == (20 * S, 1 * S)
    # backwards the absent bins come first: no persistent observation is reported at all
    assert _both([1 * S, 1 * S + 7], [20 * S, 20 * S + 1, 21 * S], False) == (21 * S, None)
    # the window looks at LATER bins in both directions: bin 3 sees the absences of bins 4..7, bin 8 does not see bin 3
    assert _both([3 * S], [4 * S, 5 * S], True) == (3 * S, None)
    assert _both([8 * S], [3 * S, 3 * S + 1], False) == (3 * S, 8 * S)
    # a tie is neither absent (> 0.5) nor present (> 0.5)
    assert _both([3 * S], [3 * S + 1], True) == (None, None)
    # count mode: thresholds are counts compared with '>'
    kw = dict(use_relative_confidence=False, absence_confidence=2.0, presence_confidence=1.0)
    assert _both([1 * S, 1 * S + 1], [9 * S, 9 * S + 1], True, **kw) == (None, 1 * S)
    assert _both([1 * S, 1 * S + 1], [9 * S, 9 * S + 1, 9 * S + 2], True, **kw) == (9 * S, 1 * S)
    # temporal_resolution 0.1 s is 0.1f * 1e9 = 100000001 ns per bin (float -> double -> integer)
    assert _both([100000001], [], True, temporal_resolution=0.1, window_size=1) == (None, 100000001)
    assert _both([100000000], [], True, temporal_resolution=0.1, window_size=1) == (None, 0)


def test_change_detector_matches_restatement_on_random_series():
    rng = np.random.default_rng(11)
    for trial in range(300):
        n_p, n_a = int(rng.integers(0, 40)), int(rng.integers(0, 40))
        span = int(rng.integers(1, 30)) * S
        present = rng.integers(0, span, n_p).astype(np.uint64)
        absent = rng.integers(0, span, n_a).astype(np.uint64)
        kw = dict(temporal_resolution=float(rng.choice([0.1, 0.5, 1.0, 2.5])), window_size=int(rng.integers(1, 8)))
        if trial % 2:
            kw.update(use_relative_confidence=False, absence_confidence=float(rng.integers(1, 6)), presence_confidence=float(rng.integers(1, 6)))
        else:
            kw.update(absence_confidence=float(rng.uniform(0.1, 0.9)), presence_confidence=float(rng.uniform(0.1, 0.9)))
        for forward in (True, False):
            _both(present, absent, forward, **kw)


def test_change_detector_rejects_bad_configuration():
    from khronos_amd import KhronosAmdError
    from khronos_amd.host_capi import detect_changes
    for kw in (dict(temporal_resolution=0.0), dict(window_size=0), dict(absence_confidence=1.5),
               dict(use_relative_confidence=False, presence_confidence=0.0)):
        with pytest.raises(KhronosAmdError):
            detect_changes([1], [2], True, **kw)
