"""Multi-GPU tick orchestration: one process per GPU, the block map sharded by contiguous
hash range, owner-computes.  Per tick the ranks
  (1) all-gather the camera frames (done by the caller; bench.py prefetches the next tick's gather on its own stream),
  (2) ingest all cameras in one launch; the motion detector's seed test runs in the same pass on each rank's shard, and a
      small all-reduce of the per-camera seed-pixel counts decides which cameras need more: for those the per-pixel voxel
      keys are reduced to the camera's home rank, which clusters them, paints the dynamic image and broadcasts it
      (shard_motion=False: the keys are all-reduced and every rank clusters the identical image),
  (3) allocate / cull for every frame (queued before the count exchange: it does not depend on the masks), then integrate
      every frame into the blocks they own (with the dynamic mask),
  (4) run the per-voxel tracking update,
  (5) all-gather fixed-size halo records (528 B per live block: key + 4096 free-or-ever-free bits),
  (6) run the ever-free stencil with remote neighbours served from the gathered records.
The collectives are torch.distributed calls (backend `nccl` = RCCL over xGMI on the GPUs, `gloo` in the CPU
tests); the shard itself is any object with the small interface below (HipShard here; the tests plug the CPU
oracle in to check the protocol).
"""
import numpy as np
import torch

HALO_WORDS = 66


class HipShard:
    """Shard backend over a khronos_amd.FusionContext; exchange buffers are HBM-resident torch tensors."""

    def __init__(self, ctx, sensor, halo_cap, device, n_cameras=1):
        self.ctx, self.sensor, self.halo_cap, self.device = ctx, sensor, halo_cap, device
        self.send = torch.zeros((halo_cap, HALO_WORDS), dtype=torch.int64, device=device)
        n = sensor.width * sensor.height
        self.keys = [torch.zeros(n, dtype=torch.int64, device=device) for _ in range(n_cameras)]

    def upload(self, cam, stamp, pose, depth, rgb, label):
        """depth / rgb / label: device tensors.  Returns the frame slot."""
        return self.ctx.upload_frame_device(self.sensor, stamp, pose, depth.data_ptr(), rgb.data_ptr(), label.data_ptr())

    # -- the cameras of a tick batched (khr_tick_ingest / khr_tick_integrate) --
    MAX_SPLIT = 8  # frames per split-phase call (khronos_amd.h)

    def tick_ingest(self, stamp, cameras, count_seeds, wait=True):
        """wait=False: nothing blocks; the counts land in self.seed_counts_dev (device int64) and tick_seed_counts()."""
        frames = [self.ctx.make_frame(stamp, pose, depth.data_ptr(), rgb.data_ptr(), label.data_ptr())
                  for (pose, depth, rgb, label) in cameras]
        if getattr(self, "seed_counts_dev", None) is None or self.seed_counts_dev.numel() != len(frames):
            self.seed_counts_dev = torch.zeros(len(frames), dtype=torch.int64, device=self.device)
        return self.ctx.tick_ingest(self.sensor, frames, count_seeds=count_seeds, want_counts=count_seeds and wait,
                                    counts_device_ptr=self.seed_counts_dev.data_ptr())

    def tick_seed_counts(self, n):
        return self.ctx.tick_seed_counts(n)

    def tick_integrate(self, slots, use_mask, phases=3):
        self.ctx.tick_integrate(slots, use_mask=use_mask, phases=phases)

    def motion_keys(self, cam, slot):
        _, n_seed = self.ctx.motion_keys(slot, device_ptr=self.keys[cam].data_ptr())
        return self.keys[cam], n_seed

    def motion_finish(self, cam, slot, keys):
        return self.ctx.detect_motion_from_keys(slot, device_ptr=keys.data_ptr())

    # -- one rank clusters a camera's motion for all (ShardedFusion, shard_motion) --
    def _image_buffer(self, cam):
        n = self.sensor.width * self.sensor.height
        if getattr(self, "_dyn_bufs", None) is None:
            self._dyn_bufs = {}
        if cam not in self._dyn_bufs:
            self._dyn_bufs[cam] = torch.zeros(n + 1, dtype=torch.int32, device=self.device)
        return self._dyn_bufs[cam]

    def dynamic_image(self, cam, slot, n_clusters):
        """the painted dynamic image of `slot` + the cluster count in the last element (device int32 tensor)."""
        buf = self._image_buffer(cam)
        self.ctx.copy_frame_image(slot, 0, buf.data_ptr())
        buf[-1:] = n_clusters
        return buf

    def image_buffer(self, cam, slot):
        return self._image_buffer(cam)

    def set_dynamic_image(self, cam, slot, img):
        self.ctx.set_frame_image(slot, 0, None, device_ptr=img.data_ptr())

    def integrate(self, cam, slot, use_mask):
        self.ctx.integrate(slot, allocate_blocks=True, use_mask=use_mask)

    def tracking_phase(self, stamp, phase):
        self.ctx.update_tracking_phase(stamp, phase)

    def export_halo(self, stamp):
        self.ctx.export_halo(self.halo_cap, device_ptr=self.send.data_ptr())
        return self.send

    def import_halo(self, gathered):
        self.ctx.import_halo(n_records=gathered.shape[0], device_ptr=gathered.data_ptr())

    # -- output stage (mesh halo) --
    def mesh_requests(self, cap):
        if getattr(self, "_req", None) is None or self._req.numel() != cap:
            self._req = torch.zeros(cap, dtype=torch.int64, device=self.device)
        self.ctx.mesh_halo_requests(cap, True, device_ptr=self._req.data_ptr())
        return self._req

    def mesh_export(self, all_requests, cap_records):
        words = self.ctx.mesh_halo_words()
        if getattr(self, "_rec", None) is None or self._rec.shape[0] != cap_records:
            self._rec = torch.zeros((cap_records, words), dtype=torch.int32, device=self.device)
        self.ctx.mesh_halo_export(None, cap_records, req_ptr=all_requests.data_ptr(), n_req=all_requests.numel(),
                                  out_ptr=self._rec.data_ptr())
        return self._rec

    def mesh_import(self, gathered):
        self.ctx.mesh_halo_import(device_ptr=gathered.data_ptr(), n_records=gathered.shape[0])

    def generate_mesh(self):
        self.ctx.generate_mesh(True, True)

    def archive(self):
        self.ctx.reset_inactive_async()
        self.ctx.clear_updated()

    def check_exchange_overflow(self):
        """the exchange buffers (halo_cap, mesh request / record caps) are fixed-size; a rank that had more to send counted it on
        the device (khr_stats.pool_exhausted, sticky) instead of sending it.  That is an error, not a result."""
        n = self.ctx.stats()["pool_exhausted"]
        if n:
            raise RuntimeError("sharded fusion: %d exchange / pool overflows (raise halo_cap / mesh_req_cap / mesh_rec_cap / "
                               "max_blocks): the map of this run is incomplete" % n)


class ShardedFusion:
    def __init__(self, shard, dist=None, world_size=1, motion=True, count_device="cpu", shard_motion=True):
        """shard_motion: the clustering of a camera's motion (seed graph, components, painting: the part of the detector
        that works on the assembled key image, not on the map) runs on the camera's home rank only, which broadcasts the
        painted dynamic image; otherwise every rank clusters every camera's identical key image."""
        self.shard, self.dist, self.world, self.motion = shard, dist, world_size, motion
        self.count_device = count_device
        self.shard_motion = shard_motion and hasattr(shard, "dynamic_image")
        self.rank = dist.get_rank() if (dist is not None and world_size > 1) else 0
        self._recv = None
        self.clusters_last_tick = []

    def all_gather(self, t):
        if self.dist is None or self.world == 1:
            return t
        # one contiguous receive buffer per (shape, dtype): the collective writes into it directly (no list + cat copy)
        key = (tuple(t.shape), t.dtype, t.device)
        if self._recv is None:
            self._recv = {}
        out = self._recv.get(key)
        if out is None:
            out = torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            self._recv[key] = out
        self.dist.all_gather_into_tensor(out, t.contiguous())
        return out

    def tick(self, stamp, cameras):
        """cameras: list of (pose, depth, rgb, label) for ALL cameras of the rig (already gathered)."""
        batched = hasattr(self.shard, "tick_ingest")  # HipShard: per-camera launches folded together
        split = batched and len(cameras) <= self.shard.MAX_SPLIT
        self.clusters_last_tick = [0] * len(cameras)
        counts = None
        if batched:
            # nothing waits here: allocation / culling (independent of the motion masks) is queued behind the ingest, so the
            # device has work while the host collects and exchanges the seed counts
            slots, counts = self.shard.tick_ingest(stamp, cameras, self.motion, wait=not split)
            if split:
                self.shard.tick_integrate(slots, use_mask=self.motion, phases=1)
        else:
            slots = [self.shard.upload(ci, stamp, pose, depth, rgb, label) for ci, (pose, depth, rgb, label) in enumerate(cameras)]
        if self.motion:
            keys = [None] * len(cameras)
            if not batched:
                counts = []
                for ci, slot in enumerate(slots):
                    keys[ci], n = self.shard.motion_keys(ci, slot)
                    counts.append(n)
            exchange = self.dist is not None and self.world > 1
            if exchange and split and self.count_device != "cpu":
                cnt = self.shard.seed_counts_dev  # written by the ingest's publish kernel: no host -> device copy
                self.dist.all_reduce(cnt)
                cnt = cnt.tolist()
            else:
                if split:
                    counts = self.shard.tick_seed_counts(len(cameras))
                if exchange:
                    cnt = torch.tensor(counts, dtype=torch.int64, device=self.count_device)
                    self.dist.all_reduce(cnt)  # which cameras have seeds anywhere
                    cnt = cnt.tolist()
                else:
                    cnt = list(counts)
            for ci, slot in enumerate(slots):
                if cnt[ci] == 0:
                    continue  # no seeds on any rank => no clusters, empty dynamic image
                home = ci % self.world
                if keys[ci] is None:  # batched ingest only counted: the voxel keys are produced when somebody has seeds
                    keys[ci], _ = self.shard.motion_keys(ci, slot)
                if not self.shard_motion:
                    if exchange:
                        self.dist.all_reduce(keys[ci])  # exactly one non-zero contribution per pixel
                    self.clusters_last_tick[ci] = self.shard.motion_finish(ci, slot, keys[ci])
                    continue
                # the camera's home rank assembles the key image, clusters it and paints; everybody else receives the
                # painted image (+ the cluster count in its last element) -- the clustering is the one stage of a tick
                # that every rank would otherwise repeat for every camera
                if exchange:
                    self.dist.reduce(keys[ci], dst=home)
                if self.rank == home:
                    n = self.shard.motion_finish(ci, slot, keys[ci])
                    self.clusters_last_tick[ci] = n
                    img = self.shard.dynamic_image(ci, slot, n) if exchange else None
                else:
                    img = self.shard.image_buffer(ci, slot)
                if exchange:
                    self.dist.broadcast(img, src=home)
                    if self.rank != home:
                        self.shard.set_dynamic_image(ci, slot, img[:-1])
                        # (the count of a foreign camera is only read where that costs no device round trip)
                        self.clusters_last_tick[ci] = int(img[-1]) if img.device.type == "cpu" else None
        if batched:
            self.shard.tick_integrate(slots, use_mask=self.motion, phases=2 if split else 3)
        else:
            for ci, slot in enumerate(slots):
                self.shard.integrate(ci, slot, use_mask=self.motion)
        self.shard.tracking_phase(stamp, 1)
        if self.world > 1:
            self.shard.import_halo(self.all_gather(self.shard.export_halo(stamp)))
        self.shard.tracking_phase(stamp, 2)
        return slots

    def output(self, req_cap=8192, rec_cap=1024):
        """ActiveWindow::extractOutputData, volumetric part (active_window.cpp:217-249): marching cubes on the
        mesh-updated blocks with the neighbours' low planes fetched from their owners (request / response
        all-gathers), then archival and flag clearing."""
        if self.world > 1:
            reqs = self.all_gather(self.shard.mesh_requests(req_cap))
            recs = self.all_gather(self.shard.mesh_export(reqs, rec_cap))
            self.shard.mesh_import(recs)
        self.shard.generate_mesh()
        self.shard.archive()
        if hasattr(self.shard, "check_exchange_overflow"):
            self.shard.check_exchange_overflow()
