"""Multi-GPU tick orchestration: one process per GPU, the block map sharded by contiguous
hash range, owner-computes.  Per tick the ranks (1) all-gather the camera frames, (2) integrate every frame
into the blocks they own, (3) run the per-voxel tracking update, (4) all-gather fixed-size halo records
(528 B per live block: key + 4096 free-or-ever-free bits), (5) run the ever-free stencil with remote
neighbours served from the gathered records.  The collectives are torch.distributed calls (backend `nccl` =
RCCL over xGMI on the GPUs, `gloo` in the CPU tests); the shard itself is any object with the small
interface below (HipShard here; the tests plug the CPU oracle in to check the protocol).
"""
import numpy as np
import torch

HALO_WORDS = 66


class HipShard:
    """Shard backend over a khronos_amd.FusionContext; halo buffers are HBM-resident torch tensors."""

    def __init__(self, ctx, sensor, halo_cap, device):
        self.ctx, self.sensor, self.halo_cap, self.device = ctx, sensor, halo_cap, device
        self.send = torch.zeros((halo_cap, HALO_WORDS), dtype=torch.int64, device=device)

    def integrate(self, stamp, pose, depth, rgb, label):
        """depth / rgb / label: device tensors."""
        slot = self.ctx.upload_frame_device(self.sensor, stamp, pose, depth.data_ptr(), rgb.data_ptr(), label.data_ptr())
        self.ctx.integrate(slot, allocate_blocks=True, use_mask=False)

    def tracking_phase(self, stamp, phase):
        self.ctx.update_tracking_phase(stamp, phase)

    def export_halo(self, stamp):
        self.ctx.export_halo(self.halo_cap, device_ptr=self.send.data_ptr())
        return self.send

    def import_halo(self, gathered):
        self.ctx.import_halo(n_records=gathered.shape[0], device_ptr=gathered.data_ptr())


class ShardedFusion:
    def __init__(self, shard, dist=None, world_size=1):
        self.shard, self.dist, self.world = shard, dist, world_size
        self._recv = None

    def all_gather(self, t):
        if self.dist is None or self.world == 1:
            return t
        if self._recv is None or self._recv[0].shape != t.shape or self._recv[0].device != t.device:
            self._recv = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(self._recv, t)
        return torch.cat(self._recv, dim=0)

    def tick(self, stamp, cameras):
        """cameras: list of (pose, depth, rgb, label) for ALL cameras of the rig (already gathered)."""
        for pose, depth, rgb, label in cameras:
            self.shard.integrate(stamp, pose, depth, rgb, label)
        self.shard.tracking_phase(stamp, 1)
        if self.world > 1:
            self.shard.import_halo(self.all_gather(self.shard.export_halo(stamp)))
        self.shard.tracking_phase(stamp, 2)
