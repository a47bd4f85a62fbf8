"""gloo worker of tests/test_cpu_distributed.py: every rank runs one hash-range shard of the block map through
khronos_amd.distributed.ShardedFusion (frames all-gathered, halo records all-gathered) with the CPU oracle as
the shard backend, and compares its shard with the unsharded oracle fed the same frames."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from khronos_amd.distributed import HALO_WORDS, ShardedFusion  # noqa: E402
from khronos_amd.synth import SyntheticStream  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from test_cpu_oracle import _cfg  # noqa: E402

W, H, N_FRAMES, HALO_CAP = 160, 120, 22, 1024


class OracleShard:
    """khronos_amd.distributed shard interface on top of the CPU oracle (frames kept per camera slot)."""

    def __init__(self, omap, sensor, halo_cap):
        self.m, self.sensor, self.halo_cap = omap, sensor, halo_cap
        self.frames, self.dyn = {}, {}

    def upload(self, cam, stamp, pose, depth, rgb, label):
        self.frames[cam] = (stamp, pose, depth.numpy(), rgb.numpy(), label.numpy())
        self.dyn[cam] = None
        return cam

    def motion_keys(self, cam, slot):
        stamp, pose, depth, _, _ = self.frames[cam]
        keys = self.m.motion_keys(self.sensor, stamp, pose, depth)
        return torch.from_numpy(keys.view(np.int64).reshape(-1)), int((keys >> np.uint64(63)).sum())

    def motion_finish(self, cam, slot, keys):
        h, w = self.sensor.height, self.sensor.width
        n, dyn, _ = self.m.detect_motion_from_keys(keys.numpy().view(np.uint64).reshape(h, w))
        self.dyn[cam] = dyn
        return n

    def dynamic_image(self, cam, slot, n_clusters):
        return torch.from_numpy(np.concatenate([self.dyn[cam].reshape(-1).astype(np.int32), np.array([n_clusters], np.int32)]))

    def image_buffer(self, cam, slot):
        return torch.zeros(self.sensor.height * self.sensor.width + 1, dtype=torch.int32)

    def set_dynamic_image(self, cam, slot, img):
        self.dyn[cam] = img.numpy().reshape(self.sensor.height, self.sensor.width).astype(np.int32)

    def integrate(self, cam, slot, use_mask):
        stamp, pose, depth, rgb, label = self.frames[cam]
        self.m.integrate(self.sensor, stamp, pose, depth, rgb, label, mask=self.dyn[cam] if use_mask else None)

    def tracking_phase(self, stamp, phase):
        self.m.update_tracking_phase(stamp, phase)

    def export_halo(self, stamp):
        return torch.from_numpy(self.m.export_halo(stamp, self.halo_cap).view(np.int64))

    def import_halo(self, gathered):
        self.m.import_halo(gathered.numpy().view(np.uint64))

    def mesh_requests(self, cap):
        return torch.from_numpy(self.m.mesh_halo_requests(cap, True)[0].view(np.int64))

    def mesh_export(self, all_requests, cap_records):
        return torch.from_numpy(self.m.mesh_halo_export(all_requests.numpy().view(np.uint64), cap_records).view(np.int32))

    def mesh_import(self, gathered):
        self.m.mesh_halo_import(gathered.numpy().view(np.uint32))

    def generate_mesh(self):
        self.m.generate_mesh(True, True)

    def archive(self):
        self.removed = self.m.reset_inactive()
        self.m.clear_updated()


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    s = SyntheticStream(W, H, threads=1)
    sen = po.OrcSensor(W, H, s.fx, s.fy, s.cx, s.cy, 0.1, 5.0)
    kw = dict(voxel_size=0.1, truncation_distance=0.3, md_min_cluster_size=5, md_min_separation_distance=2.0, md_max_range=5.0)
    shard = po.OracleMap(_cfg(rank=rank, world_size=world, **kw))
    full = po.OracleMap(_cfg(**kw))
    fusion = ShardedFusion(OracleShard(shard, sen, HALO_CAP), dist, world)
    clusters_total = 0
    mesh_checks = 0
    for i in range(N_FRAMES):
        # rank r renders camera r of the rig; the frames are all-gathered like in bench.py
        yaw = 2 * np.pi * rank / world
        fr = s.render(i, yaw_offset=yaw)
        mine = [torch.from_numpy(fr["depth"]), torch.from_numpy(fr["rgb"]), torch.from_numpy(fr["label"])]
        gathered = []
        for t in mine:
            out = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(out, t)
            gathered.append(out)
        cams = [(s.pose(i, yaw_offset=2 * np.pi * r / world), gathered[0][r], gathered[1][r], gathered[2][r])
                for r in range(world)]
        fusion.tick(fr["stamp"], cams)
        n_full = []
        for pose, d, c, l in cams:  # reference order: motion detection, then integration with the mask
            n, dyn, _ = full.detect_motion(sen, fr["stamp"], pose, d.numpy())
            n_full.append(n)
            full.integrate(sen, fr["stamp"], pose, d.numpy(), c.numpy(), l.numpy(), mask=dyn)
        full.update_tracking(fr["stamp"])
        assert fusion.clusters_last_tick == n_full, (i, fusion.clusters_last_tick, n_full)
        clusters_total += sum(n_full)
        if i % 5 == 4:  # output cadence: mesh (with halo), archival, flag clearing
            fusion.output(req_cap=4096, rec_cap=1024)
            full.generate_mesh(True, True)
            removed_full = full.reset_inactive()
            full.clear_updated()
            # union of the shards' archived blocks == the unsharded archive
            cnt = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
            dist.all_gather(cnt, torch.tensor([len(fusion.shard.removed)], dtype=torch.int64))
            assert sum(int(c) for c in cnt) == len(removed_full), (i, cnt, len(removed_full))
            # union of the shards' meshes == the unsharded mesh (vertex count and coordinate checksum)
            mine = shard.mesh()
            tot = torch.tensor([float(len(mine["points"])), float(mine["points"].astype(np.float64).sum())], dtype=torch.float64)
            dist.all_reduce(tot)
            fm = full.mesh()
            assert int(tot[0]) == len(fm["points"]) and len(fm["points"]) > 0, (i, tot, len(fm["points"]))
            assert abs(float(tot[1]) - float(fm["points"].astype(np.float64).sum())) < 1e-6 * max(1.0, len(fm["points"]))
            mesh_checks += 1
    # every block of the shard equals the unsharded block, field by field (incl. ever_free bits)
    mine = shard.block_indices()
    all_idx = full.block_indices()
    counts = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([len(mine)], dtype=torch.int64))
    assert sum(int(c) for c in counts) == len(all_idx), (counts, len(all_idx))
    assert len(mine) > 0
    ef = 0
    for idx in mine:
        a, b = shard.get_block(idx), full.get_block(idx)
        for k in ("distance", "weight", "last_observed", "last_occupied", "flags", "sem_label"):
            assert np.array_equal(a[k], b[k]), (rank, k, idx)
        ef += int((a["flags"] & 2).sum())
    tot = torch.tensor([ef], dtype=torch.int64)
    dist.all_reduce(tot)
    assert int(tot) > 0, "ever-free never fired: the halo path was not exercised"
    assert clusters_total > 0, "motion detector never fired: the key exchange was not exercised"
    assert mesh_checks >= 3
    if rank == 0:
        print("DIST_OK blocks=%d ever_free=%d clusters=%d" % (len(all_idx), int(tot), clusters_total))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
