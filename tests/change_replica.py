"""The two callers of the ray verificator (RayBackgroundChangeDetector::detectChanges, ray_background_change_detector.cpp:59-103;
RayObjectChangeDetector::detectChanges / checkObjectObservation, ray_object_change_detector.cpp:62-160), restated as per-vertex loops
over the CPU oracle's RayVerificator and time-bin vote -- the same loops tests/test_gpu_rayver.py::
test_background_and_object_change_detectors_equal_restatement holds the product's batched host drivers to (written out there),
as functions, so that tests/test_cpu_ref_pin.py can hold them to the reference's OWN detector code.  Test infrastructure."""
import numpy as np

from oracle import pyoracle as po

UNOBSERVED, PERSISTENT, ABSENT = 0, 1, 2


def threshold_ns(time_filtering_threshold):
    """uint64(config.time_filtering_threshold * 1e9): float * double (:57, :60)"""
    return int(np.float32(time_filtering_threshold) * 1e9)


def vertex_state(ora, point, earliest, **vote):
    """checkVertex (ray_background_change_detector.cpp:90-103)"""
    _, _, pres, absn = ora.check(np.asarray(point, np.float32)[None], earliest, 2 ** 64 - 1)
    ca, fp = po.detect_changes(pres, absn, True, **vote)
    return ABSENT if ca is not None else (PERSISTENT if fp is not None else UNOBSERVED)


def background_changes(ora, verts, vstamps, time_filtering_threshold, states=None, reobserved=(), **vote):
    """detectChanges (:59-88): vertices beyond len(states) are new; re-observed ones (indices beyond the mesh ignored, :73-75) are
    recomputed.  -> (states, number of re-observed vertices whose state changed)"""
    thr = threshold_ns(time_filtering_threshold)
    out = [] if states is None else [int(x) for x in states]
    n_prev = len(out)
    for i in range(n_prev, len(verts)):
        out.append(vertex_state(ora, verts[i], int(vstamps[i]) + thr, **vote))
    changed = 0
    for i in reobserved:
        if i >= len(verts):
            continue
        s = vertex_state(ora, verts[i], int(vstamps[i]) + thr, **vote)
        changed += int(s != out[i])
        out[i] = s
    return np.array(out, np.uint8), changed


def object_change(ora, local, bbox_min, bbox_max, t_first, t_last, time_filtering_threshold, query_subsampling, **vote):
    """checkObjectObservation (ray_object_change_detector.cpp:117-160): every query_subsampling-th vertex, moved from the box frame to the
    world (BoundingBox::world_P_center = 0.5f * (min + max)), checked before the object's life and after it; both result sets merged and
    voted once per direction."""
    thr = threshold_ns(time_filtering_threshold)
    local = np.asarray(local, np.float32)
    ctr = np.array([np.float32(0.5) * (np.float32(bbox_min[d]) + np.float32(bbox_max[d])) for d in range(3)], np.float32)
    before, after = [[], []], [[], []]
    for i in range(0, len(local), query_subsampling):
        p = (local[i] + ctr).astype(np.float32)
        _, _, pres, absn = ora.check(p[None], 0, (t_first - thr) % (1 << 64))
        before[0] += list(pres)
        before[1] += list(absn)
        _, _, pres, absn = ora.check(p[None], t_last + thr, 2 ** 64 - 1)
        after[0] += list(pres)
        after[1] += list(absn)
    bca, bfp = po.detect_changes(before[0], before[1], False, **vote)
    aca, afp = po.detect_changes(after[0], after[1], True, **vote)
    return dict(first_absent=bca or 0, last_absent=aca or 0, first_persistent=bfp or 0, last_persistent=afp or 0), (before, after)
