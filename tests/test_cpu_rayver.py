"""Oracle known-answer tests for khronos::RayVerificator (ray_verificator.cpp:66-145, 327-349)."""
import numpy as np

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
