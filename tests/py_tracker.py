"""Independent (second-version) restatement of khronos::MaxIoUTracker / ExternalTracker in Python, written from
max_iou_tracker.cpp / external_tracker.cpp, for the N-version check of the C++ host plugins.  Test infrastructure."""
import numpy as np

f32 = np.float32


def from_seconds(s):
    # hydra::fromSeconds on a float config value: float -> double -> * 1e9 -> uint64
    return int(float(f32(s)) * 1e9)


class Track:
    def __init__(self):
        self.id = 0
        self.is_dynamic = False
        self.is_active = True
        self.confidence = f32(0)
        self.first_seen = 0
        self.last_seen = 0
        self.category = -1
        self.has_semantics = False
        self.observations = []
        self.last_voxels = set()
        self.last_points = np.zeros((0, 3), f32)  # track_by = pixels
        self.last_box = None
        self.last_centroid = np.zeros(3, f32)


def iou_voxels(cluster_voxels, track_voxels):
    inter = f32(0)
    for v in cluster_voxels:
        if v in track_voxels:
            inter = f32(inter + f32(1))
    return f32(inter / f32(f32(len(cluster_voxels) + len(track_voxels)) - inter))


def iou_box(a, b):
    lo = np.maximum(a[0], b[0])
    hi = np.minimum(a[1], b[1])
    if not (hi > lo).all():
        return f32(0)
    d = (hi - lo).astype(f32)
    inter = f32(f32(d[0] * d[1]) * d[2])
    da, db = (a[1] - a[0]).astype(f32), (b[1] - b[0]).astype(f32)
    va, vb = f32(f32(da[0] * da[1]) * da[2]), f32(f32(db[0] * db[1]) * db[2])
    uni = f32(f32(va + vb) - inter)
    return f32(inter / uni) if uni > 0 else f32(0)


class MaxIoUTracker:
    def __init__(self, track_by="voxels", association="assign_cluster", min_semantic_iou=0.5, min_cosine_sim=0.0, min_cross_iou=0.5,
                 max_dynamic_distance=1.0, temporal_window=3.0, min_num_observations=20, voxel_size=0.1):
        self.track_by, self.association = track_by, association
        self.min_semantic_iou, self.min_cross_iou = f32(min_semantic_iou), f32(min_cross_iou)
        self.max_dynamic_distance = f32(max_dynamic_distance)
        self.temporal_window, self.min_num_observations, self.voxel_size = temporal_window, min_num_observations, f32(voxel_size)
        self.tracks, self.next_id, self.stamp = [], 0, 0
        self.cam = None  # track_by = pixels: (world_T_sensor, fx, fy, cx, cy, W, H) of the frame being processed

    # -- measurements --
    def centroid(self, c):
        if self.track_by == "bounding_box":
            return (f32(0.5) * (c["box"][0] + c["box"][1])).astype(f32)
        if self.track_by == "pixels":  # max_iou_tracker.cpp:543-549
            return (c["points"].astype(np.float64).sum(0) / len(c["points"])).astype(f32)
        s = np.zeros(3, f32)
        for v in sorted(c["voxels"]):
            s = (s + (np.array(v, f32) + f32(0.5)) * self.voxel_size).astype(f32)
        return (s / f32(len(c["voxels"]))).astype(f32)

    def iou(self, c, t):
        if self.track_by == "pixels":
            from oracle import np_oracle as npo
            return npo.iou_pixels(c["pixels"], t.last_points, *self.cam)[0]
        if self.track_by == "voxels":
            return iou_voxels(c["voxels"], t.last_voxels)
        return iou_box(t.last_box, c["box"])

    # -- track updates --
    def update(self, c, t, dynamic):
        if self.track_by == "voxels":
            t.last_voxels = set(c["voxels"])
        if self.track_by == "pixels":
            t.last_points = c["points"]
        t.last_box = c["box"]
        if not dynamic and not t.has_semantics and c.get("category") is not None:
            t.has_semantics, t.category = True, c["category"]
        t.last_seen = self.stamp
        t.observations.append((self.stamp, -1 if dynamic else c["id"], c["id"] if dynamic else -1))
        t.confidence = min(f32(f32(len(t.observations)) / f32(self.min_num_observations * 2)), f32(1))

    def new_track(self, c, dynamic):
        t = Track()
        t.is_dynamic, t.id, t.first_seen = dynamic, self.next_id, self.stamp
        self.next_id += 1
        self.tracks.append(t)
        self.update(c, t, dynamic)
        return t

    @staticmethod
    def semantics_match(c, t):
        c_has = c.get("category") is not None
        if c_has != t.has_semantics:
            return False
        if not c_has:
            return True
        return c["category"] == t.category

    def process(self, stamp, semantic, dynamic):
        self.stamp = stamp
        # dynamic association
        used = set()
        for t in self.tracks:
            if not t.is_dynamic:
                continue
            best, best_d, best_c = None, self.max_dynamic_distance, None
            for c in dynamic:
                if c["id"] in used:
                    continue
                cen = self.centroid(c)
                d = (cen - t.last_centroid).astype(f32)
                dist = f32(np.sqrt(f32(f32(f32(d[0] * d[0]) + f32(d[1] * d[1])) + f32(d[2] * d[2]))))
                if dist < best_d:
                    best, best_d, best_c = c, dist, cen
            if best is not None:
                used.add(best["id"])
                self.update(best, t, True)
                t.last_centroid = best_c
        for c in dynamic:
            if c["id"] not in used:
                t = self.new_track(c, True)
                t.last_centroid = self.centroid(c)
        # semantic clusters -> dynamic tracks
        used = set()
        for t in self.tracks:
            if not t.is_dynamic:
                continue
            best, best_iou = None, self.min_cross_iou
            for c in semantic:
                if c["id"] in used:
                    continue
                i = self.iou(c, t)
                if i > best_iou:
                    best, best_iou = c, i
            if best is not None:
                used.add(best["id"])
                if t.last_seen < self.stamp:
                    self.update(best, t, False)
                else:
                    s, _, d = t.observations[-1]
                    t.observations[-1] = (s, best["id"], d)
        if self.association == "assign_cluster":
            for t in self.tracks:
                if t.is_dynamic:
                    continue
                best, best_iou = None, self.min_semantic_iou
                for c in semantic:
                    if c["id"] in used or not self.semantics_match(c, t):
                        continue
                    i = self.iou(c, t)
                    if i > best_iou:
                        best, best_iou = c, i
                if best is not None:
                    used.add(best["id"])
                    self.update(best, t, False)
            for c in semantic:
                if c["id"] not in used:
                    self.new_track(c, False)
        else:
            for c in semantic:
                if c["id"] in used:
                    continue
                done = False
                for t in self.tracks:
                    if t.is_dynamic or not self.semantics_match(c, t):
                        continue
                    if self.iou(c, t) < self.min_semantic_iou:
                        continue
                    done = True
                    used.add(c["id"])
                    self.update(c, t, False)
                    break
                if not done:
                    self.new_track(c, False)
        min_time = (self.stamp - from_seconds(self.temporal_window)) % (1 << 64)
        for t in self.tracks:
            t.is_active = t.last_seen >= min_time


class ExternalTracker:
    def __init__(self, temporal_window=3.0, min_num_observations=20):
        self.temporal_window, self.min_num_observations = temporal_window, min_num_observations
        self.tracks, self.stamp = [], 0

    def update(self, c, t):
        if not t.has_semantics and c.get("category") is not None:
            t.has_semantics, t.category = True, c["category"]
        t.last_seen = self.stamp
        t.observations.append((self.stamp, c["id"], -1))
        t.confidence = min(f32(f32(len(t.observations)) / f32(self.min_num_observations * 2)), f32(1))

    def process(self, stamp, semantic, dynamic):
        self.stamp = stamp
        used = set()
        for t in self.tracks:
            for c in semantic:
                if c["id"] in used:
                    continue
                if t.id == c["id"]:
                    used.add(c["id"])
                    self.update(c, t)
                    break
        for c in semantic:
            if c["id"] not in used:
                t = Track()
                t.id, t.first_seen = c["id"], stamp
                self.tracks.append(t)
                self.update(c, t)
        min_time = (self.stamp - from_seconds(self.temporal_window)) % (1 << 64)
        for t in self.tracks:
            t.is_active = t.last_seen >= min_time
