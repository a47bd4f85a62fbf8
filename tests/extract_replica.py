"""MeshObjectExtractor::extractObject for a static track, restated on the CPU oracle (mesh_object_extractor.cpp:81-118, 174-304,
306-356; configuration of khronos_amd.configs.OBJECT_YAML).  Test infrastructure shared by tests/test_gpu_bench_path.py (the
product's extracted objects against it) and tests/test_cpu_ref_pin.py (it against the reference's OWN extractor code)."""
import numpy as np

f32 = np.float32


def extract_static(e, t, threads=1, min_observations=0.0, obj_cfg=None):
    """MeshObjectExtractor::extractObject for a static track, restated on the oracle (mesh_object_extractor.cpp:81-118,
    174-304, 306-356; configuration of bench.py's OBJECT_YAML).  -> None (no object) or dict(points, bbox_min, bbox_max).
    obj_cfg: settings of the extractor's OWN projective / mesh integrator (mesh_object_extractor.cpp:63-64,239,267) as khr_config
    fields, where they differ from the defaults."""
    from khronos_amd import default_config
    from oracle import pyoracle as po
    if not (t.confidence > f32(0.5)) or t.is_dynamic:
        return None
    by_stamp = {fr["stamp"]: fr for fr in e.frames}
    frames, lo, hi = [], None, None
    for (stamp, sem_id, _dyn_id) in t.observations:
        if sem_id == -1 or stamp not in e.sem:
            continue
        _, oimg, boxes = e.sem[stamp]
        frames.append((by_stamp[stamp], oimg, sem_id))
        if sem_id in boxes:
            b0, b1 = boxes[sem_id]
            lo = b0.copy() if lo is None else np.minimum(lo, b0)
            hi = b1.copy() if hi is None else np.maximum(hi, b1)
    if not frames or lo is None:
        return None
    dim = (hi - lo).astype(f32)
    if f32(f32(dim[0] * dim[1]) * dim[2]) < f32(0.005):
        return None
    center = (f32(0.5) * (lo + hi)).astype(f32)
    vs = f32(max(f32(dim.max()) * f32(0.02), f32(0.0)))
    if not vs > 0:
        return None
    inv = f32(1) / (vs * f32(8))
    mn = np.floor((center - dim) * inv).astype(np.int32)
    mx = np.floor((center + dim) * inv).astype(np.int32)
    blocks = [[x, y, z] for x in range(mn[0], mx[0] + 1) for y in range(mn[1], mx[1] + 1) for z in range(mn[2], mx[2] + 1)]
    ocfg = default_config(voxel_size=float(vs), voxels_per_side=8, truncation_distance=float(vs * f32(2)), with_semantics=1,
                          with_tracking=0, num_labels=2, semantic_mode=1, **(obj_cfg or {}))
    om = po.OracleMap(po.config_from(ocfg, threads))
    try:
        om.allocate_blocks(blocks)
        for fr, oimg, sem_id in frames:
            om.integrate(e.osen, fr["stamp"], fr["pose"], fr["depth"], fr["rgb"], None, object_image=oimg.astype(np.int32),
                         object_id=sem_id, allocate_blocks=False)
        om.object_prune(0.5, float(min_observations))  # (min_object_reconstruction_observations; OBJECT_YAML: 0)
        om.generate_mesh(True, False)
        pts = om.mesh()["points"].astype(f32)
    finally:
        om.close()
    if len(pts) == 0:
        return None  # only_extract_reconstructed_objects
    b0, b1 = pts.min(0), pts.max(0)
    d = (b1 - b0).astype(f32)
    vol = f32(f32(d[0] * d[1]) * d[2])
    if vol > f32(10.0) or vol < f32(0.005):
        return None
    return dict(points=(pts - (f32(0.5) * (b0 + b1)).astype(f32)).astype(f32), bbox_min=b0, bbox_max=b1,
                first_seen=t.first_seen, last_seen=t.last_seen, label=t.category if t.has_semantics else -1)
