"""Second, independent restatement (numpy, vectorised) of the per-voxel projective TSDF / semantic update and
of updateTrackingDuration, used ONLY to cross-check oracle.cpp (N-version check; the reference ships no golden
vectors).  Follows ASSUMPTIONS.md A.1-A.4 and tracking_integrator.cpp:224-246.
TEST INFRASTRUCTURE ONLY."""
import numpy as np

f32 = np.float32


def make_pose(T):
    T = np.asarray(T, np.float64).reshape(4, 4)
    R = T[:3, :3].T.astype(f32)
    t = (-(T[:3, :3].T @ T[:3, 3])).astype(f32)  # NB: the oracle sums column-wise left to right
    t = np.array([-(T[0, r] * T[0, 3] + T[1, r] * T[1, 3] + T[2, r] * T[2, 3]) for r in range(3)], np.float64).astype(f32)
    return R, t


def integrate_block(cfg, sensor, T, depth, label, block_index, dist, weight, lik, sem_valid, sem_label, last_obs, stamp):
    """updates the arrays of one 16^3 block in place; returns (n_upd, n_band)."""
    vps = cfg["voxels_per_side"]
    vs, trunc = f32(cfg["voxel_size"]), f32(cfg["truncation_distance"])
    bs = vs * f32(vps)
    W, H = sensor["width"], sensor["height"]
    fx, fy, cx, cy = (f32(sensor[k]) for k in ("fx", "fy", "cx", "cy"))
    mn, mx = f32(sensor["min_range"]), f32(sensor["max_range"])
    R, t = make_pose(T)
    ix, iy, iz = np.meshgrid(np.arange(vps), np.arange(vps), np.arange(vps), indexing="ij")
    lin = (ix + vps * (iy + vps * iz)).ravel()
    order = np.argsort(lin)
    ix, iy, iz = ix.ravel()[order], iy.ravel()[order], iz.ravel()[order]
    o = np.asarray(block_index, np.int64).astype(f32) * bs
    px = o[0] + (ix.astype(f32) + f32(0.5)) * vs
    py = o[1] + (iy.astype(f32) + f32(0.5)) * vs
    pz = o[2] + (iz.astype(f32) + f32(0.5)) * vs
    pc = [((R[r, 0] * px + R[r, 1] * py) + R[r, 2] * pz) + t[r] for r in range(3)]
    z = pc[2]
    ok = z > 0
    with np.errstate(all="ignore"):
        zs = np.where(ok, z, f32(1))
        ok &= ~((zs < mn) | (zs > mx))
        u = (pc[0] * fx) / zs + cx
        v = (pc[1] * fy) / zs + cy
        ok &= ~((np.ceil(u) >= W) | (np.floor(u) < 0)) & ~((np.ceil(v) >= H) | (np.floor(v) < 0))
        uc, vc = np.where(ok, u, f32(0)), np.where(ok, v, f32(0))
        u0, v0 = np.floor(uc).astype(np.int64), np.floor(vc).astype(np.int64)
        u1, v1 = np.minimum(u0 + 1, W - 1), np.minimum(v0 + 1, H - 1)
        du, dv = uc - u0.astype(f32), vc - v0.astype(f32)
        rng = np.where((depth > 0) & np.isfinite(depth), depth, f32(0)).astype(f32)
        pxs = [(v0, u0), (v1, u0), (v0, u1), (v1, u1)]
        r4 = [rng[a, b] for a, b in pxs]
        w4 = [(f32(1) - du) * (f32(1) - dv), (f32(1) - du) * dv, du * (f32(1) - dv), du * dv]
        mode = cfg["interpolation_method"]
        nearest = np.zeros(len(lin), bool) if mode != 0 else np.ones(len(lin), bool)
        if mode == 2:
            nearest |= (np.maximum.reduce(r4) - np.minimum.reduce(r4)) > f32(cfg["adaptive_max_range_difference"])
        near_idx = np.where(du >= 0.5, 2, 0) + np.where(dv >= 0.5, 1, 0)
        r_near = np.choose(near_idx, r4)
        r_bil = ((w4[0] * r4[0] + w4[1] * r4[1]) + w4[2] * r4[2]) + w4[3] * r4[3]
        ds = np.where(nearest, r_near, r_bil)
        ok &= (ds >= mn) & ~(ds > mx)
        sdf = ds - zs
        ok &= ~(sdf < -trunc)
        band = ok & (np.abs(sdf) < trunc)
        q = vs / zs
        w = (fx * fy) * (q * q)
        w = w / (zs * zs)
        eps = f32(cfg["weight_dropoff_epsilon"]) if cfg["weight_dropoff_epsilon"] > 0 else f32(cfg["weight_dropoff_epsilon"]) * -vs
        wd = np.maximum(w * ((trunc + sdf) / (trunc - eps)), f32(0))
        w = np.where(sdf < -eps, wd, w)
        ok &= w > 0
        band &= ok
        sdf_c = np.maximum(np.minimum(trunc, sdf), -trunc)
        d_new = (dist * weight + sdf_c * w) / (weight + w)
        w_new = np.minimum(weight + w, f32(cfg["max_weight"]))
    dist[ok] = d_new[ok]
    weight[ok] = w_new[ok]
    last_obs[ok] = stamp
    if label is not None and cfg["with_semantics"]:
        K = cfg["num_labels"]
        best = np.where(nearest, near_idx, np.argmax(np.stack(w4), axis=0))  # first maximum
        bv = np.choose(best, [p[0] for p in pxs])
        bu = np.choose(best, [p[1] for p in pxs])
        lab = label[bv, bu]
        sel = band & (lab >= 0) & (lab < K)
        lm = np.log(f32(cfg["label_confidence"]))
        ln = np.log((f32(1) - f32(cfg["label_confidence"])) / f32(K - 1))
        idx = np.flatnonzero(sel)
        lik[:, idx[~sem_valid[idx]]] = 0
        add = np.full((K, len(idx)), ln, f32)
        add[lab[idx], np.arange(len(idx))] = lm
        lik[:, idx] = lik[:, idx] + add
        sem_valid[idx] = True
        sem_label[idx] = np.argmax(lik[:, idx], axis=0)
    return int(ok.sum()), int(band.sum())


def tracking_block(cfg, dist, last_obs, last_occ, flags, stamp):
    """updateTrackingDuration over one block (tracking_integrator.cpp:224-246); flags bit0 active, bit2 to_remove."""
    thr = f32(cfg["tsdf_occupancy_threshold"])
    thr = thr * -f32(cfg["voxel_size"]) if thr < 0 else thr
    last_occ[dist < thr] = stamp
    was = (flags & 1) > 0
    act = (last_obs.astype(np.float64) / 1e9) >= (np.float64(stamp) / 1e9 - np.float64(f32(cfg["temporal_window"])))
    flags[:] = (flags & ~np.uint8(1)) | act.astype(np.uint8)
    flags[was & ~act] |= 4
    return bool(act.any())


# ---- MaxIoUTracker::computeIoUPixels (max_iou_tracker.cpp:578-600) and InstanceForwarding (instance_forwarding.cpp:80-149) ----
def reproject_pixels(points, world_T_sensor, fx, fy, cx, cy, W, H):
    """std::set<Pixel> of :583-593: every point goes through getSensorPose() (AS the reference applies it: the pose is
    named sensor_T_world there but is world_T_sensor) in double, is cast to float and projected with
    Sensor::projectPointToImagePlane(p, int& u, int& v) (ASSUMPTIONS.md A.8: z > 0, pinhole, nearest pixel, inside)."""
    T = np.asarray(world_T_sensor, np.float64).reshape(4, 4)
    p = np.asarray(points, np.float64).reshape(-1, 3)
    ps = (p @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
    out = set()
    f32 = np.float32
    for x, y, z in ps:
        if not z > 0:
            continue
        uf = f32(f32(f32(x * f32(fx)) / z) + f32(cx))
        vf = f32(f32(f32(y * f32(fy)) / z) + f32(cy))
        # std::round: halves away from zero
        u = float(np.floor(abs(float(uf)) + 0.5) * (1 if uf >= 0 else -1))
        v = float(np.floor(abs(float(vf)) + 0.5) * (1 if vf >= 0 else -1))
        if 0 <= u < W and 0 <= v < H:
            out.add((int(u), int(v)))
    return out


def iou_pixels(cluster_pixels, track_points, world_T_sensor, fx, fy, cx, cy, W, H):
    """:595-599; cluster_pixels = list of (u, v), track_points = Track::last_points (n, 3)."""
    rep = reproject_pixels(track_points, world_T_sensor, fx, fy, cx, cy, W, H)
    inter = np.float32(0)
    for px in cluster_pixels:
        if px in rep:
            inter = np.float32(inter + np.float32(1))
    return np.float32(inter / np.float32(np.float32(len(cluster_pixels) + len(track_points)) - inter)), int(inter)


def forward_instances(label, range_image, vertex_map, max_range=0.0, background_ids=()):
    """extractSemanticClusters before the size / volume filters: {id: dict(pixels (column-major scan order), bbox)}."""
    H, W = label.shape
    bg = set(int(b) for b in background_ids)
    out = {}
    for u in range(W):
        col = label[:, u]
        for v in np.flatnonzero(col):
            i = int(col[v])
            if i in bg:
                continue
            if max_range > 0 and range_image[v, u] > max_range:
                continue
            out.setdefault(i, []).append((u, int(v)))
    res = {}
    for i, px in out.items():
        pts = np.array([vertex_map[v, u] for (u, v) in px], np.float32)
        res[i] = dict(id=i, pixels=px, num_pixels=len(px), bbox_min=pts.min(0), bbox_max=pts.max(0))
    return res


# ---- TrackingIntegrator::updateBlockEverFree (tracking_integrator.cpp:168-222) over a whole map ----
def _neighbour_offsets(nn):
    faces = [(-1, 0, 0), (1, 0, 0), (0, -1, 0), (0, 1, 0), (0, 0, -1), (0, 0, 1)]
    edges = [(dx, dy, dz) for dx in (-1, 0, 1) for dy in (-1, 0, 1) for dz in (-1, 0, 1) if abs(dx) + abs(dy) + abs(dz) == 2]
    corners = [(dx, dy, dz) for dx in (-1, 1) for dy in (-1, 1) for dz in (-1, 1)]
    return {6: faces, 18: faces + edges, 26: faces + edges + corners}[nn]  # the RESULT does not depend on the order


def voxel_is_free(cfg, last_occ, last_obs, stamp):
    """voxelIsFree (:248-252): last_occupied strictly older than temporal_buffer (double seconds) and observed at all."""
    return ((last_occ.astype(np.float64) / 1e9) < (np.float64(stamp) / 1e9 - np.float64(f32(cfg["temporal_buffer"])))) & (last_obs != 0)


def ever_free_pass(cfg, blocks, updated, stamp):
    """blocks: {(bx, by, bz): dict(last_obs, last_occ, flags)} (flat arrays, x fastest; flags bit1 = ever_free);
    updated: the block indices the integrator touched in this frame.  Sets the ever_free bit where the voxel is free and all
    `neighbor_connectivity` neighbours are ever-free or free; a neighbour in a block that does not exist fails the test.
    (The reference reads neighbours' ever_free while other threads set it; the outcome is the same either way because a
    voxel that becomes ever-free in this pass is free, ASSUMPTIONS.md B.)"""
    vps = cfg["voxels_per_side"]
    offs = _neighbour_offsets(cfg["neighbor_connectivity"])
    # free-or-ever-free volume of every block BEFORE the pass, as (z, y, x) cubes
    F, free = {}, {}
    for b, d in blocks.items():
        fr = voxel_is_free(cfg, d["last_occ"], d["last_obs"], stamp)
        free[b] = fr
        F[b] = (((d["flags"] & 2) > 0) | fr).reshape(vps, vps, vps)
    for b in updated:
        d = blocks[b]
        pad = np.zeros((vps + 2, vps + 2, vps + 2), bool)  # missing block => False
        for dz in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    nb = (b[0] + dx, b[1] + dy, b[2] + dz)
                    if nb not in F:
                        continue
                    src = F[nb]
                    zs = slice(0, 1) if dz == 1 else (slice(vps - 1, vps) if dz == -1 else slice(0, vps))
                    ys = slice(0, 1) if dy == 1 else (slice(vps - 1, vps) if dy == -1 else slice(0, vps))
                    xs = slice(0, 1) if dx == 1 else (slice(vps - 1, vps) if dx == -1 else slice(0, vps))
                    zd = slice(vps + 1, vps + 2) if dz == 1 else (slice(0, 1) if dz == -1 else slice(1, vps + 1))
                    yd = slice(vps + 1, vps + 2) if dy == 1 else (slice(0, 1) if dy == -1 else slice(1, vps + 1))
                    xd = slice(vps + 1, vps + 2) if dx == 1 else (slice(0, 1) if dx == -1 else slice(1, vps + 1))
                    pad[zd, yd, xd] = src[zs, ys, xs]
        ok = np.ones((vps, vps, vps), bool)
        for (dx, dy, dz) in offs:
            ok &= pad[1 + dz:1 + dz + vps, 1 + dy:1 + dy + vps, 1 + dx:1 + dx + vps]
        new = free[b].reshape(vps, vps, vps) & ok
        d["flags"][new.ravel()] |= 2


# ---- FreeSpaceMotionDetector (free_space_motion_detector.cpp:73-399), second restatement: plain Python sets / dicts ----
def motion_point_map(cfg, sensor, T, depth, blocks):
    """setUpPointMap (:105-203).  blocks: {(bx, by, bz): ever_free bool array [nvox]} of the tracking layer as it is when the
    detector runs (before the frame is integrated).  Returns (point_map {voxel (x, y, z): [pixel index v * W + u, ...]} in
    (v, u) order, seeds {voxel}); pixels the reference skips are absent."""
    vps = cfg["voxels_per_side"]
    vs = f32(cfg["voxel_size"])
    bs = vs * f32(vps)
    bs_inv, vs_inv = f32(1) / bs, f32(1) / vs
    W, H = sensor["width"], sensor["height"]
    fx, fy, cx, cy = (f32(sensor[k]) for k in ("fx", "fy", "cx", "cy"))
    Tm = np.asarray(T, np.float64).reshape(4, 4)
    Rw, tw = Tm[:3, :3].astype(np.float32), Tm[:3, 3].astype(np.float32)
    min_z_world = f32(Tm[2, 3] + np.float64(f32(cfg["md_min_z_coordinate"])))  # :80
    d = np.asarray(depth, np.float32)
    u, v = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32))
    valid = (d > 0) & np.isfinite(d)
    rng = np.where(valid, d, f32(0))  # range_mode 0 (A.2)
    x = ((u - cx) / fx) * d
    y = ((v - cy) / fy) * d
    pw = [((Rw[r, 0] * x + Rw[r, 1] * y) + Rw[r, 2] * d) + tw[r] for r in range(3)]
    pw = [np.where(valid, c, f32(0)) for c in pw]  # invalid pixels carry the vertex (0, 0, 0) (A.2)
    keep = (rng > 0) & ~(rng > f32(cfg["md_max_range"])) & ~(pw[2] < min_z_world)  # :169-176
    bi = [np.floor(c * bs_inv).astype(np.int64) for c in pw]
    point_map, seeds = {}, set()
    for vv, uu in zip(*np.nonzero(keep)):
        b = (int(bi[0][vv, uu]), int(bi[1][vv, uu]), int(bi[2][vv, uu]))
        ef = blocks.get(b)
        if ef is None:  # :180 no tracking block
            continue
        o = [f32(b[k]) * bs for k in range(3)]
        vi = [int(np.floor((pw[k][vv, uu] - o[k]) * vs_inv)) for k in range(3)]
        if min(vi) < 0 or max(vi) >= vps:  # :192-197 (float rounding at a block border): in no voxel
            continue
        g = (b[0] * vps + vi[0], b[1] * vps + vi[1], b[2] * vps + vi[2])
        point_map.setdefault(g, []).append(int(vv) * W + int(uu))
        if ef[vi[0] + vps * (vi[1] + vps * vi[2])]:
            seeds.add(g)  # :198-201
    return point_map, seeds


def motion_clusters(cfg, point_map, seeds, W, H):
    """clusterDynamicVoxels (:205-272) + mergeClusters (:274-355) + applyClusterLevelFilters (:365-379) +
    writeClustersToData (:381-399).  Returns (number of clusters written, dynamic image)."""
    offs = _neighbour_offsets(cfg["md_neighbor_connectivity"])
    closed = set()
    clusters = []
    for seed in sorted(seeds):  # C.1: ascending (x, y, z) stands in for the unordered_set order
        if seed in closed:
            continue
        stack, pixels, voxels = [seed], [], set()
        while stack:
            g = stack.pop()
            if g in closed:
                continue
            closed.add(g)
            if g not in point_map:
                continue
            pixels += point_map[g]
            voxels.add(g)
            for (dx, dy, dz) in offs:
                ng = (g[0] + dx, g[1] + dy, g[2] + dz)
                if ng in seeds:
                    stack.append(ng)
                elif ng in point_map:  # occupied neighbour: appended once per adjacent seed, no closed-set test (:255-265)
                    pixels += point_map[ng]
                    voxels.add(ng)
                    closed.add(ng)
        clusters.append([pixels, voxels])
    n = len(clusters)
    sep = f32(cfg["md_min_separation_distance"])

    def overlap(a, b):  # checkClusterOverlap: integer norm of the int64 difference, truncated (C.2)
        for p in clusters[a][1]:
            for q in clusters[b][1]:
                n2 = (p[0] - q[0]) ** 2 + (p[1] - q[1]) ** 2 + (p[2] - q[2]) ** 2
                if f32(int(np.sqrt(np.float64(n2)))) < sep:
                    return True
        return False

    ov = [[False] * n for _ in range(n)]
    for i in range(n):
        for j in range(i + 1, n):
            ov[i][j] = ov[j][i] = overlap(i, j)
    merged, keep = [False] * n, [False] * n

    def connected(ci, out):  # getConnectedClusters: depth-first over the overlap relation, in index order
        for i in range(n):
            if not merged[i] and ov[ci][i]:
                merged[i] = True
                out.append(i)
                connected(i, out)

    for cur in range(n):
        if merged[cur]:
            continue
        idx = []
        connected(cur, idx)
        for i in idx:
            if i != cur:
                clusters[cur][0] += clusters[i][0]
                clusters[cur][1] |= clusters[i][1]
        keep[cur] = True
    dyn = np.zeros(W * H, np.int32)
    cid, n_out = 1, 0
    for ci in range(n):
        if not keep[ci]:
            continue
        size = len(clusters[ci][0])
        if size < cfg["md_min_cluster_size"] or size > cfg["md_max_cluster_size"]:
            continue
        dyn[np.array(clusters[ci][0], np.int64)] = cid  # later clusters overwrite earlier ones (:388-389)
        if cid < 255:
            cid += 1
        n_out += 1
    return n_out, dyn.reshape(H, W)


# ---- frustum allocation of ProjectiveIntegrator::updateMap(allocate_blocks = true) (ASSUMPTIONS.md A.3) ----
def visible_blocks(cfg, sensor, T):
    """block indices (n, 3) whose centre passes pointIsInViewFrustum(centre in the sensor frame, 0.8660254 * block_size),
    vectorised over the (2n + 1)^3 candidate cube around the camera's block; fp32, the operation order of A.3."""
    vps = cfg["voxels_per_side"]
    bs = f32(cfg["voxel_size"]) * f32(vps)
    bs_inv = f32(1) / bs
    W, H = sensor["width"], sensor["height"]
    fx, fy, cx, cy = (f32(sensor[k]) for k in ("fx", "fy", "cx", "cy"))
    max_range = f32(sensor["max_range"])
    R, t = make_pose(T)
    Tm = np.asarray(T, np.float64).reshape(4, 4)
    tw = Tm[:3, 3].astype(np.float32)
    n = int(np.ceil(max_range * bs_inv)) + 1
    bc = np.floor(tw * bs_inv).astype(np.int64)
    xl, xr = (f32(0) - cx) / fx, (f32(W) - cx) / fx
    yt, yb = (f32(0) - cy) / fy, (f32(H) - cy) / fy
    tl, tr, bl, br = (np.array(v, np.float32) for v in ((xl, yt, 1), (xr, yt, 1), (xl, yb, 1), (xr, yb, 1)))

    def crossn(a, b):
        x = a[1] * b[2] - a[2] * b[1]
        y = a[2] * b[0] - a[0] * b[2]
        z = a[0] * b[1] - a[1] * b[0]
        nn = np.sqrt((x * x + y * y) + z * z, dtype=np.float32)
        return np.array([x / nn, y / nn, z / nn], np.float32)

    normals = [crossn(bl, tl), crossn(tr, br), crossn(tl, tr), crossn(br, bl)]  # left, right, top, bottom (inward)
    infl = f32(0.8660254) * bs
    d = np.arange(-n, n + 1, dtype=np.int64)
    dz, dy, dx = np.meshgrid(d, d, d, indexing="ij")
    b = np.stack([bc[0] + dx.ravel(), bc[1] + dy.ravel(), bc[2] + dz.ravel()], axis=1)
    c = (b.astype(np.float32) + f32(0.5)) * bs
    pc = [((R[r, 0] * c[:, 0] + R[r, 1] * c[:, 1]) + R[r, 2] * c[:, 2]) + t[r] for r in range(3)]
    ok = ~(pc[2] < -infl)
    n2 = (pc[0] * pc[0] + pc[1] * pc[1]) + pc[2] * pc[2]
    lim = max_range + infl
    ok &= ~(n2 > lim * lim)
    for nrm in normals:
        dd = (pc[0] * nrm[0] + pc[1] * nrm[1]) + pc[2] * nrm[2]
        ok &= ~(dd < -infl)
    return b[ok]
