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

W, H, N_FRAMES, HALO_CAP = 160, 120, 14, 1024


class OracleShard:
    def __init__(self, omap, sensor, halo_cap):
        self.m, self.sensor, self.halo_cap = omap, sensor, halo_cap

    def integrate(self, stamp, pose, depth, rgb, label):
        self.m.integrate(self.sensor, stamp, pose, depth.numpy(), rgb.numpy(), label.numpy())

    def tracking_phase(self, stamp, phase):
        self.m.update_tracking_phase(stamp, phase)

    def export_halo(self, stamp):
        return torch.from_numpy(self.m.export_halo(stamp, self.halo_cap).view(np.int64))

    def import_halo(self, gathered):
        self.m.import_halo(gathered.numpy().view(np.uint64))


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    s = SyntheticStream(W, H, threads=1)
    sen = po.OrcSensor(W, H, s.fx, s.fy, s.cx, s.cy, 0.1, 5.0)
    shard = po.OracleMap(_cfg(rank=rank, world_size=world, voxel_size=0.2, truncation_distance=0.4))
    full = po.OracleMap(_cfg(voxel_size=0.2, truncation_distance=0.4))
    fusion = ShardedFusion(OracleShard(shard, sen, HALO_CAP), dist, world)
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
        for pose, d, c, l in cams:
            full.integrate(sen, fr["stamp"], pose, d.numpy(), c.numpy(), l.numpy())
        full.update_tracking(fr["stamp"])
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
    if rank == 0:
        print("DIST_OK blocks=%d ever_free=%d" % (len(all_idx), int(tot)))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
