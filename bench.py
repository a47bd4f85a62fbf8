#!/usr/bin/env python3
"""bench.py — active-window frames/s (+ Mvoxel-updates/s) of the MI355X fusion path.

A "step" is one ActiveWindow::spinOnce worth of volumetric work on one synthetic RGB-D+label frame per
camera (reference order, active_window.cpp:118-174): motion detection -> projective TSDF / label
integration (with the dynamic mask) -> tracking + ever-free update, and every `--output-every` frames
the output extraction (marching cubes on updated blocks, archival of inactive blocks, flag clearing;
active_window.cpp:217-249 at uHumans2's min_output_separation = 0.4 s).

Workloads (`--config`, BASELINE.json `configs`; explicit --width / --voxel-size / ... override a preset):
  c3 (default, the configuration the metric is quoted on): 1280x720 RGB-D + labels, 2 cm voxels, truncation 6 cm,
     K = 20 labels, MotionDetector + object detection / tracking / extraction on, output every 4th frame (0.4 s).
     (Run on stamps EXACTLY 0.1 s apart, the reference's own rate limit -- last + fromSeconds(0.4f) > stamp, active_window.cpp:158-160 --
     lets only every FIFTH frame through, 0.4f being 0.4000000060; tests/test_cpu_ref_pin.py runs the reference's ActiveWindow and
     shows it.  Every 4th is the nominal cadence and the heavier load, so it is what is timed.)
  c2: 640x480, 5 cm voxels, truncation 15 cm, static background TSDF only (khronos_ros/config/mapper/ground_truth.yaml:56-94:
     no motion detector, no object detector, min_output_separation 0 = mesh + archival every frame).
  c1: 640x480, 5 cm voxels, the projective integrator alone (allocation + TSDF / label update of each frame).
Frames are rendered on the host BEFORE the timed region and are resident in HBM when it starts.  `--preroll` frames are
fused before the warm-up so that the timed steps run on a map / track set in steady state (tracks old enough to leave
the window and be extracted).

N > 1 (one process per GPU, launched by torch.distributed.run): rank r owns camera r of an N-camera rig
and the hash-range shard r of the block map; every tick the N camera frames are all-gathered over RCCL
and each rank integrates all of them into the blocks it owns (owner-computes, weak scaling).

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", choices=["c1", "c2", "c3", "c4", "c5"], default="c3",
                    help="BASELINE.json workload preset (see module docstring); c4 / c5 are the rig geometries (per camera: c4 = c3's, "
                         "c5 = 1920x1080 at 1 cm) and are meant for --gpus N or --emulate-world N (4 / 8 cameras)")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--voxel-size", type=float, default=None)
    ap.add_argument("--num-labels", type=int, default=20)
    ap.add_argument("--max-blocks", type=int, default=40960)
    ap.add_argument("--output-every", type=int, default=None)
    ap.add_argument("--no-motion", action="store_true", default=None)
    ap.add_argument("--no-objects", action="store_true", default=None,
                    help="leave out the object half of the active window (ConnectedSemantics, MaxIoUTracker, MeshObjectExtractor)")
    ap.add_argument("--no-tracking", action="store_true", default=None, help="leave out TrackingIntegrator::updateBlocks (c1)")
    ap.add_argument("--preroll", type=int, default=None,
                    help="frames fused before the warm-up (untimed): brings the map and the tracker to steady state; default for c3 / c4: "
                         "75 - warmup, so that the timed steps always START at frame 75 of the stream whatever --warmup is -- the moving sphere "
                         "is in view (motion seeds, dynamic clusters) in frames 89..102, so the driver's 20 steps contain 6 such frames and "
                         "5 outputs (VERDICT r05 item 1: the r01-r05 window, frames 65..84, contained none); 0 otherwise")
    ap.add_argument("--input", choices=["device", "host"], default="device",
                    help="where a frame lives when khr_process_frame is called: device = resident in HBM before the timed region (the headline: "
                         "BASELINE's metric is quoted with inputs resident); host = in page-locked HOST memory, as the reference's spinOnce receives "
                         "its packet (active_window.cpp:118-125,268-286): khr_process_frame(.., on_device = 0, KHR_PF_INPUT_PINNED), the planes of "
                         "frame i + 1 travel on the context's copy stream (khr_ingest_ahead_host) while frame i is fused")
    ap.add_argument("--lookahead", action="store_true",
                    help="hand the next frame over with khr_ingest_ahead while the current one is fused (N = 1).  Off by default: measured in "
                         "round 4, it does not shorten the step (the host and the main stream meet at the seed count every frame) and "
                         "the early conversion competes with k_fuse for the memory path")
    ap.add_argument("--latency-frames", type=int, default=8,
                    help="frames after the timed region that are run with a device synchronisation after each: per-frame latency "
                         "(the reference's active_window/all scope) beside the pipelined throughput; 0 = skip")
    ap.add_argument("--exact", action="store_true", help="(default) khr_config.exact_arithmetic = 1: values bit-identical to the CPU restatement")
    ap.add_argument("--fast", action="store_true", help="khr_config.exact_arithmetic = 0: decisions exact, distance / weight within ~1e-6 relative")
    ap.add_argument("--buffer-frames", type=int, default=100, help="frame_data_buffer.max_buffer_size (frames kept in HBM per camera)")
    ap.add_argument("--cpu-baseline-frames", type=int, default=-1,
                    help="frames of the same stream timed on the CPU oracle (rank 0, N=1); -1 = auto, 0 = skip")
    ap.add_argument("--output-copy", choices=["none", "device", "host"], default="device",
                    help="what happens to an output's map clone (VolumetricMap::cloneUpdated, active_window.cpp:229, inside the reference's "
                         "active_window/all timer scope): device (default line) = device-side snapshot of the updated blocks at every output "
                         "(what the drop-in's ActiveWindowOutput carries); none = the timed step hands out no clone; host = snapshot + "
                         "download of the layers a host consumer reads + mesh fetch")
    ap.add_argument("--host-fields", choices=["tsdf", "all"], default="tsdf",
                    help="--output-copy host: the layers the host consumer downloads per updated block -- tsdf = distance + weight (8 B per "
                         "voxel: what a TSDF / places consumer reads; colour and labels travel with the mesh vertices), all = the six "
                         "layers of khr_download_updated (25 B per voxel)")
    ap.add_argument("--no-extra-streams", action="store_true",
                    help="default c3 run only: do not append the c1 / c2 streams and the output-copy variants (each a short sub-run of this script)")
    ap.add_argument("--no-roofline-timers", action="store_true")
    ap.add_argument("--all-timers", action="store_true", help="HIP-event timers on every kernel group (slower host path)")
    ap.add_argument("--frame-times", action="store_true", help="debug: synchronise and print per-frame wall times")
    ap.add_argument("--mesh-req-cap", type=int, default=None, help="mesh halo requests all-gathered per rank (N > 1); default 16384 (c5: 131072)")
    ap.add_argument("--mesh-rec-cap", type=int, default=None, help="mesh halo records all-gathered per rank (N > 1); default 2048 (c5: 32768)")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="development: ONE process plays rank 0 of an N-rank sharded run (all N cameras rendered locally, no "
                         "collectives) to measure the per-rank tick cost on a 1-GPU box; the JSON line is marked emulation")
    ap.add_argument("--sender-ingest", action="store_true",
                    help="sharded tick with sender-side ingest (kdist_tick_own): a rank converts only its own camera's frame, the ranks "
                         "all-gather the CONVERTED planes and adopt them in place; with --emulate-world the other cameras' planes are "
                         "converted before the timed region (they stand for the all-gather's receive buffer).  Needs --dist-host cxx")
    ap.add_argument("--dist-host", choices=["cxx", "torch"], default="cxx",
                    help="N > 1: who issues the tick's collectives: cxx = RCCL from libkhronos_amd_host.so (kdist_*, the product "
                         "path), torch = khronos_amd/distributed.py over torch.distributed (the protocol test harness)")
    ap.add_argument("--halo-cap", type=int, default=None, help="halo records all-gathered per rank and tick (N > 1); default 8192 (c5: 65536)")
    a = ap.parse_args()
    preset = {"c3": dict(width=1280, height=720, voxel_size=0.02, output_every=4, no_motion=False, no_objects=False, no_tracking=False, preroll=max(0, 75 - a.warmup)),
              "c2": dict(width=640, height=480, voxel_size=0.05, output_every=1, no_motion=True, no_objects=True, no_tracking=False, preroll=0),
              "c1": dict(width=640, height=480, voxel_size=0.05, output_every=0, no_motion=True, no_objects=True, no_tracking=True, preroll=0),
              # BASELINE configs[3]: 4-camera rig, per camera the c3 stream; configs[4]: 8 cameras 1920x1080 at 1 cm (~8x the
              # blocks per camera: larger exchange buffers; the pre-roll is shorter because a tick is ~10x a c3 frame)
              "c4": dict(width=1280, height=720, voxel_size=0.02, output_every=4, no_motion=False, no_objects=False, no_tracking=False, preroll=max(0, 75 - a.warmup)),
              "c5": dict(width=1920, height=1080, voxel_size=0.01, output_every=4, no_motion=False, no_objects=False, no_tracking=False, preroll=40,
                         mesh_req_cap=131072, mesh_rec_cap=32768, halo_cap=65536)}[a.config]
    for k, v in preset.items():
        if getattr(a, k) is None:
            setattr(a, k, v)
    for k, v in dict(mesh_req_cap=16384, mesh_rec_cap=2048, halo_cap=8192).items():
        if getattr(a, k) is None:
            setattr(a, k, v)
    if a.config == "c5" and a.max_blocks == 40960:
        a.max_blocks = 65536
    return a


from khronos_amd.configs import OBJECT_YAML  # noqa: E402  (`active_window:` YAML of the object half, uHumans2.yaml:35-100)


from khronos_amd.bench_line import compact_line  # noqa: E402  (torch-free: tests/test_cpu_bench_line.py imports it without this script)


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback for the fusion path)")
    # debug switches to exercise the N > 1 code path on a 1-GPU box (all ranks on device 0, gloo):
    backend = os.environ.get("KHR_BENCH_BACKEND", "nccl")
    if os.environ.get("KHR_BENCH_SAME_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    emu = args.emulate_world if (args.emulate_world > 1 and world == 1) else 0
    if emu:
        world = emu  # shard parameters, cameras and tick structure of an N-rank run; dist stays None

    import __graft_entry__ as entry
    if rank == 0:
        entry.build()
    if dist is not None:
        dist.barrier()
    from khronos_amd import FusionContext, default_config
    from khronos_amd.synth import SyntheticStream

    W, H, vs = args.width, args.height, args.voxel_size
    K = args.num_labels
    # stream layout: [pre-roll | warm-up | timed steps | latency frames]; only world == 1 runs pre-roll / latency frames
    pre = args.preroll if world == 1 else 0
    lat = args.latency_frames if (world == 1 and not args.frame_times) else 0
    w0 = pre                      # first warm-up frame
    t0i = pre + args.warmup       # first timed frame
    t1i = t0i + args.steps        # first latency frame
    # N > 1: a few more ticks behind the timed region with HIP events around every collective (kdist_profile): the line says
    # what each exchange of the tick costs without those events sitting inside the timed steps
    prof_ticks = 8 if (world > 1 and not emu) else 0
    n_total = t1i + lat + prof_ticks
    trunc = 3.0 * vs
    cfg = default_config(
        voxel_size=vs, truncation_distance=trunc, voxels_per_side=16, with_semantics=1, with_tracking=1,
        exact_arithmetic=0 if args.fast else 1,
        num_labels=K, max_blocks=args.max_blocks, max_frame_pixels=W * H,
        # buffered frames stay resident in the device ring (FrameDataBuffer role); every rank holds all cameras' frames
        num_frame_slots=max(2, world) if args.no_objects else world * (args.buffer_frames + 1) + 64,  # + frames held by detached extractions
        max_mesh_vertices=48 << 20,
        # khronos_ros/config/mapper/uHumans2.yaml:52-57
        md_min_cluster_size=500, md_min_separation_distance=2.0, md_max_range=5.0,
        device=local_rank, rank=rank, world_size=world)
    ctx = FusionContext(cfg)
    # one explicit (non-default) HIP stream shared by torch / RCCL and the fusion kernels
    # (highest priority, like the context's own streams: the window's streams must not share a hardware queue with the extraction
    # workers' -- khronos_amd.hip::createStream; KHR_STREAM_PRIORITY=0 switches both off for the A/B)
    stream = torch.cuda.Stream(device=local_rank, priority=-1 if os.environ.get("KHR_WINDOW_PRIORITY") == "1" else 0)
    ctx.set_stream(stream.cuda_stream)

    # object half of the active window: the reference plugins configured as in khronos_ros/config/mapper/uHumans2.yaml:60-100
    # (object labels = the scene's primitives 7..19; this rank's own camera is detected / tracked / extracted here)
    pipe = None
    if not args.no_objects:
        from khronos_amd.host_capi import ObjectPipeline
        pipe = ObjectPipeline(ctx, OBJECT_YAML % dict(vs=args.voxel_size, trunc=3 * args.voxel_size, buf=args.buffer_frames))

    # ---- synthetic input, rendered before the timed region, resident in HBM ----------------------
    s = SyntheticStream(W, H, seed=1234)
    sensor = ctx.make_sensor(W, H, s.fx, s.fy, s.cx, s.cy, 0.1, 5.0)
    yaw = 2.0 * math.pi * rank / world
    frames_host = []
    dev = torch.device("cuda", local_rank)
    d_depth, d_rgb, d_label, poses, stamps = [], [], [], [], []
    emu_cams = []
    for i in range(n_total):
        fr = s.render(i, yaw_offset=yaw)
        if rank == 0 and world == 1:
            frames_host.append(fr)
        if emu:
            others = [s.render(i, yaw_offset=2.0 * math.pi * r / world) for r in range(1, world)]
            emu_cams.append([(torch.from_numpy(o["depth"]).to(dev), torch.from_numpy(o["rgb"]).to(dev),
                              torch.from_numpy(o["label"]).to(dev)) for o in others])
        d_depth.append(torch.from_numpy(fr["depth"]).to(dev))
        d_rgb.append(torch.from_numpy(fr["rgb"]).to(dev))
        d_label.append(torch.from_numpy(fr["label"]).to(dev))
        stamps.append(fr["stamp"])
        poses.append([s.pose(i, yaw_offset=2.0 * math.pi * r / world) for r in range(world)])
    if world > 1 and not emu:
        # one packed buffer per frame (depth f32 | label i32 | rgb u8x3) so that a tick needs ONE frame all-gather
        npx = W * H
        packed = []
        for i in range(n_total):
            pk = torch.empty(11 * npx, dtype=torch.uint8, device=dev)
            pk[: 4 * npx] = d_depth[i].view(torch.uint8).reshape(-1)
            pk[4 * npx: 8 * npx] = d_label[i].view(torch.uint8).reshape(-1)
            pk[8 * npx:] = d_rgb[i].reshape(-1)
            packed.append(pk)
        # two receive buffers: the all-gather of tick i + 1 runs on its own stream while tick i is fused (the frames of a
        # tick do not depend on the map, so the exchange is off the critical path as long as it is shorter than a tick)
        g_flat, g_depth, g_label, g_rgb = [], [], [], []
        for _b in range(2):
            gf = torch.empty(world * 11 * npx, dtype=torch.uint8, device=dev)  # flat: the concatenated form every backend accepts
            gp = gf.view(world, 11 * npx)
            g_flat.append(gf)
            g_depth.append([gp[r, : 4 * npx].view(torch.float32).view(H, W) for r in range(world)])
            g_label.append([gp[r, 4 * npx: 8 * npx].view(torch.int32).view(H, W) for r in range(world)])
            g_rgb.append([gp[r, 8 * npx:].view(H, W, 3) for r in range(world)])
        comm_stream = torch.cuda.Stream(device=local_rank)
        gather_work = {}

        def issue_gather(i):
            if i >= n_total or i in gather_work:
                return
            comm_stream.wait_stream(stream)  # the buffer's previous reader (tick i - 2) has been queued on `stream`
            with torch.cuda.stream(comm_stream):
                gather_work[i] = dist.all_gather_into_tensor(g_flat[i % 2], packed[i], async_op=True)
    torch.cuda.synchronize()

    # input descriptors (khr_frame: stamp, pose, HBM pointers) are built before the timed region
    frame_desc = None
    host_input = args.input == "host" and world == 1 and not emu
    h_pinned = []
    if world == 1:
        if host_input:
            # the frames in page-locked host memory (what a frontend's double-buffered input queue holds)
            for i in range(n_total):
                fr = frames_host[i]
                h_pinned.append((torch.from_numpy(np.ascontiguousarray(fr["depth"])).pin_memory(), torch.from_numpy(np.ascontiguousarray(fr["rgb"])).pin_memory(),
                                 torch.from_numpy(np.ascontiguousarray(fr["label"])).pin_memory()))
            # a frontend recycles a handful of page-locked buffers, so every transfer of its steady state comes out of memory the copy
            # engine has moved before; here every synthetic frame has buffers of its own, each used once -- and the FIRST transfer
            # out of a fresh page-locked region runs at ~8 GB/s instead of ~50 (tools/ubench/h2d_rates.hip, first repetition).  One
            # throw-away transfer per buffer, before the timed region, puts the stream into that steady state.
            scratch = [torch.empty_like(t, device=dev) for t in h_pinned[0]]
            for hp in h_pinned:
                for sc, t in zip(scratch, hp):
                    sc.copy_(t, non_blocking=True)
            torch.cuda.synchronize()
            del scratch
            frame_desc = [ctx.make_frame(stamps[i], poses[i][0], h_pinned[i][0].data_ptr(), h_pinned[i][1].data_ptr(), h_pinned[i][2].data_ptr())
                          for i in range(n_total)]
        else:
            frame_desc = [ctx.make_frame(stamps[i], poses[i][0], d_depth[i].data_ptr(), d_rgb[i].data_ptr(), d_label[i].data_ptr())
                          for i in range(n_total)]
    fusion = None
    fusion_cxx = None
    dist_host = "none"
    if world > 1:
        if not emu and args.dist_host == "cxx":  # (the token and the agreement below travel over whatever backend torch.distributed uses)
            # the tick's collectives from C++ (RCCL on the context's stream); torch.distributed only carries the 128-byte
            # rendezvous token and the synthetic frames.  Creation is agreed on collectively: if any rank cannot build
            # its communicator, every rank falls back to the torch harness below.
            from khronos_amd.host_capi import ShardedFusionHost
            tok = torch.zeros(128, dtype=torch.uint8, device=dev)
            if rank == 0:
                tok.copy_(torch.frombuffer(bytearray(ShardedFusionHost.unique_id()), dtype=torch.uint8))
            dist.broadcast(tok, 0)
            ok = torch.ones(1, dtype=torch.int32, device=dev)
            try:
                fusion_cxx = ShardedFusionHost(ctx, sensor, rank, world, bytes(tok.cpu().numpy().tobytes()), n_cameras=world,
                                               halo_cap=args.halo_cap, mesh_req_cap=args.mesh_req_cap,
                                               mesh_rec_cap=args.mesh_rec_cap, motion=not args.no_motion, shard_motion=True)
            except Exception as e:  # noqa: BLE001
                print("rank %d: kdist_create failed (%s); falling back to --dist-host torch" % (rank, e), file=sys.stderr)
                ok.zero_()
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                if fusion_cxx is not None:
                    fusion_cxx.close()
                fusion_cxx = None
                ctx.set_stream(stream.cuda_stream)
            else:
                stream = torch.cuda.ExternalStream(fusion_cxx.stream(), device=dev)  # the tick's stream, for event ordering
                dist_host = "cxx"
                if rank == 0:  # (first line on stderr: what the N-rank run is made of)
                    print("bench.py: %d ranks, one per GPU; the tick's collectives are issued from C++ (libkhronos_amd_host.so, kdist_*) over %s -- "
                          "ncclCommInitRank succeeded on all %d ranks (agreed by a MIN all-reduce); motion exchange %s"
                          % (world, os.environ.get("KDIST_RCCL_LIB", "librccl.so.1"), world,
                             "dense (KDIST_MOTION_DENSE)" if os.environ.get("KDIST_MOTION_DENSE") else "compact (2 bits / pixel up, 1 byte / pixel down)"),
                          file=sys.stderr, flush=True)
        elif emu and args.dist_host == "cxx":
            from khronos_amd.host_capi import ShardedFusionHost
            fusion_cxx = ShardedFusionHost(ctx, sensor, 0, world, None, n_cameras=world, halo_cap=args.halo_cap,
                                           mesh_req_cap=args.mesh_req_cap, mesh_rec_cap=args.mesh_rec_cap, motion=not args.no_motion,
                                           shard_motion=True, emulate=True)
            stream = torch.cuda.ExternalStream(fusion_cxx.stream(), device=dev)
            dist_host = "cxx-emulated"
        if fusion_cxx is None:
            from khronos_amd.distributed import HipShard, ShardedFusion
            with torch.cuda.stream(stream):
                fusion = ShardedFusion(HipShard(ctx, sensor, args.halo_cap, dev, n_cameras=world), dist, world,
                                       motion=not args.no_motion, count_device=dev if backend == "nccl" else "cpu")
            dist_host = "emulated" if emu else "torch"

    sender_ingest = bool(args.sender_ingest) and fusion_cxx is not None
    emu_conv = []
    if sender_ingest and emu:
        # the other cameras' frames as their home ranks would send them: converted BEFORE the timed region, one packed buffer
        # per tick standing for the all-gather's receive buffer (entry 0 = this rank's own camera is not read)
        cb = ctx.converted_bytes(sensor)
        torch.cuda.synchronize()
        for i in range(n_total):
            buf = torch.empty(world * cb, dtype=torch.uint8, device=dev)
            for r in range(1, world):
                dep, rgb, lab = emu_cams[i][r - 1]
                sl, _ = ctx.tick_ingest(sensor, [ctx.make_frame(stamps[i], poses[i][r], dep.data_ptr(), rgb.data_ptr(), lab.data_ptr())],
                                        count_seeds=False)
                ctx.export_converted(sl[0], buf.data_ptr() + r * cb)
            emu_conv.append(buf)
        ctx.sync()
        torch.cuda.synchronize()
    _trace = ctx.lib.khr_host_trace  # no-op unless KHR_HOST_TRACE is set
    _tags = {k: k.encode() for k in ("step_begin", "step_end", "timed_begin", "join_begin", "timed_end")}

    def step(i):
        _trace(_tags["step_begin"])
        with torch.cuda.stream(stream):
            _step(i)
        _trace(_tags["step_end"])

    def _step(i):
        if emu:
            cams = [(d_depth[i], d_rgb[i], d_label[i], poses[i][0])] + [
                (dep, rgb, lab, poses[i][r + 1]) for r, (dep, rgb, lab) in enumerate(emu_cams[i])]
        elif world > 1 and sender_ingest:
            cams = [(d_depth[i], d_rgb[i], d_label[i], poses[i][r]) for r in range(world)]  # (only entry `rank` is read as an image)
        elif world > 1:
            issue_gather(i)
            gather_work.pop(i).wait()  # `stream` waits for this tick's frames
            stream.wait_stream(comm_stream)
            issue_gather(i + 1)        # next tick's frames travel while this tick is fused
            b = i % 2
            cams = [(g_depth[b][r], g_rgb[b][r], g_label[b][r], poses[i][r]) for r in range(world)]
        else:
            cams = [(d_depth[i], d_rgb[i], d_label[i], poses[i][0])]
        out_now = args.output_every > 0 and (i + 1) % args.output_every == 0
        if world > 1:
            # sharded tick: integrate all cameras into the owned blocks, tracking, halo all-gather, ever-free
            obj_slot = None
            if fusion_cxx is not None and sender_ingest:
                slots, clusters, obj_slot = fusion_cxx.tick_own(stamps[i], [
                    ctx.make_frame(stamps[i], pose, dep.data_ptr() if r == rank else 0, rgb.data_ptr() if r == rank else 0,
                                   lab.data_ptr() if r == rank else 0) for r, (dep, rgb, lab, pose) in enumerate(cams)],
                    emu_conv[i].data_ptr() if emu else 0)
            elif fusion_cxx is not None:
                slots, clusters = fusion_cxx.tick(stamps[i], [
                    ctx.make_frame(stamps[i], pose, dep.data_ptr(), rgb.data_ptr(), lab.data_ptr()) for (dep, rgb, lab, pose) in cams])
            else:
                slots = fusion.tick(stamps[i], [(pose, dep, rgb, lab) for (dep, rgb, lab, pose) in cams])
                clusters = fusion.clusters_last_tick
            if obj_slot is None:
                obj_slot = slots[rank]
            if pipe is not None:  # owner-computes for objects: each rank handles its own camera
                pipe.finish_frame()  # tracker association of the previous tick, while this tick's kernels run
                pipe.launch_frame(obj_slot, stamps[i], poses[i][rank], sensor, clusters[rank])
            if out_now:
                if fusion_cxx is not None:
                    fusion_cxx.output()
                else:
                    fusion.output(req_cap=args.mesh_req_cap, rec_cap=args.mesh_rec_cap)
                if pipe is not None:
                    t_e = time.perf_counter()
                    n_obj, n_rm, _ = pipe.extract_inactive()
                    obj_stats[0] += n_obj
                    obj_stats[1] += n_rm
                    obj_stats[2] += time.perf_counter() - t_e
            return
        for ci, (dep, rgb, lab, pose) in enumerate(cams):
            flags = ctx.PF_INPUT_READY  # the synthetic frames are resident and complete before the timed region
            if host_input:
                flags |= ctx.PF_INPUT_PINNED
            if not args.no_motion and world == 1:
                flags |= ctx.PF_MOTION
            if pipe is not None:
                flags |= ctx.PF_OBJECTS
            last = ci == len(cams) - 1
            if last:
                if not args.no_tracking:
                    flags |= ctx.PF_TRACKING  # TrackingIntegrator::updateBlocks once per tick, after all cameras
                if out_now:
                    flags |= ctx.PF_OUTPUT
                    if args.output_copy != "none":
                        flags |= ctx.PF_SNAPSHOT
            _t0 = time.perf_counter()
            # input look-ahead (khr_ingest_ahead[_host]): the frontend's input queue already holds the next packet, so it is handed
            # over BEFORE this frame's call -- its transfer (host frames) and conversion run on the context's other streams across
            # the host's wait for the motion detector's seed count inside khr_process_frame (up to two frames may be in the look-ahead)
            if lookahead and len(cams) == 1 and (flags & ctx.PF_MOTION):
                hand = ctx.ingest_ahead_host if host_input else ctx.ingest_ahead
                if not handed and hand(sensor, frame_desc[i]) is not None:
                    handed.append(i)
                if handed and handed[0] == i and len(handed) < 2 and i + 1 < n_total and hand(sensor, frame_desc[i + 1]) is not None:
                    handed.append(i + 1)
            if handed and handed[0] == i:
                flags |= ctx.PF_INGESTED
                handed.pop(0)
            slot, n_dyn = ctx.process_frame(sensor, frame_desc[i], not host_input, flags)
            dyn_log[i] = n_dyn
            _t1 = time.perf_counter()
            host_t[0] += _t1 - _t0
            if args.output_copy == "host":
                host_consumer_poll()  # (behind this frame's launches: the device works on while the host looks at the previous output)
            if last and out_now and args.output_copy != "none":
                # the output's map clone
                _trace(b"snap_take")
                snap = ctx.take_snapshot()
                copy_stats[0] += 1
                if args.output_copy == "host":
                    # a consumer on the host (the Hydra frontend takes outputs from a queue): the clone's transfer is queued on the
                    # context's copy stream as soon as its block count is known and runs beside the next frames; at most two
                    # outputs are in flight (two sets of pinned buffers); the mesh gather is queued now, collected next frame
                    while len(host_inflight) >= 2 or (host_pending[0] is not None and len(host_inflight) >= 1):
                        host_consumer_retire()
                    if host_pending[0] is not None:
                        host_consumer_begin()  # (waits for the count: the device fell more than an output behind)
                    host_pending[0] = snap
                    host_mesh_copy_wait()  # (the staging block is about to be rewritten)
                    ctx.fetch_mesh_launch()
                    host_mesh_pending[0] = True
                else:
                    # device mode: kept until the NEXT output (a consumer that is one output behind), then dropped
                    if held_snapshot[0] is not None:
                        held_snapshot[0].release()
                    held_snapshot[0] = snap
            if pipe is not None:
                # software pipeline: the tracker association of the previous frame runs on the host while this frame's
                # kernels execute; this frame's voxel-set passes are queued behind them and collected next time
                pipe.finish_frame()
                _t2 = time.perf_counter()
                pipe.launch_frame(slot, stamps[i], pose, sensor, n_dyn)
                _t3 = time.perf_counter()
                host_t[1] += _t2 - _t1
                host_t[2] += _t3 - _t2
                if last and out_now:
                    t_e = time.perf_counter()
                    n_obj, n_rm, _ = pipe.extract_inactive()
                    obj_stats[0] += n_obj
                    obj_stats[1] += n_rm
                    obj_stats[2] += time.perf_counter() - t_e

    copy_stats = [0, 0]       # outputs whose map clone was taken, bytes brought to the host (--output-copy)
    dyn_log = {}              # frame index -> dynamic clusters the motion detector found (frames with seeds run the clustering chain)
    handed = []               # indices of the frames handed over with khr_ingest_ahead[_host], oldest first
    lookahead = (args.lookahead or args.input == "host") and world == 1 and not emu  # (host frames: the copy of frame i + 1 beside frame i)
    held_snapshot = [None]
    # ---- --output-copy host: the pipelined host consumer ----
    host_fields = (("indices", torch.int32, 3), ("distance", torch.float32, 4096), ("weight", torch.float32, 4096)) + (
        (("color", torch.uint8, 4 * 4096), ("last_observed", torch.int64, 4096), ("flags", torch.uint8, 4096), ("sem_label", torch.int32, 4096))
        if args.host_fields == "all" else ())
    host_bytes_per_block = 12 + 4096 * (25 if args.host_fields == "all" else 8)
    host_bufs = []            # two sets of pinned arrays (capacity of a snapshot: 8192 blocks), allocated on first use
    host_inflight = []        # [(snapshot, buffer set)] whose download has begun
    host_pending = [None]     # snapshot whose block count was not known yet when it was taken
    host_mesh_pending = [False]
    host_mesh_bufs = {}
    host_mesh_thread = [None]
    host_next_buf = [0]

    def host_consumer_begin():
        _trace(b"hc_begin_enter")
        if not host_bufs:
            for _b in range(2):
                host_bufs.append({nm: torch.empty(8192 * per, dtype=dt_).pin_memory() for nm, dt_, per in host_fields})
        b = host_next_buf[0]
        host_next_buf[0] ^= 1
        order = ("indices", "distance", "weight", "color", "last_observed", "flags", "sem_label")
        host_pending[0].download_begin([host_bufs[b][k].data_ptr() if k in host_bufs[b] else 0 for k in order], 8192)
        host_inflight.append(host_pending[0])
        host_pending[0] = None
        _trace(b"hc_begin_exit")

    def host_consumer_retire():
        _trace(b"hc_retire_enter")
        snap0 = host_inflight.pop(0)
        nb_ = snap0.download_end()
        copy_stats[1] += nb_ * host_bytes_per_block
        snap0.release()
        _trace(b"hc_retire_exit")

    def host_mesh_copy_wait():
        if host_mesh_thread[0] is not None:
            host_mesh_thread[0].join()
            host_mesh_thread[0] = None

    def host_consumer_poll():
        _trace(b"hc_poll_enter")
        if host_mesh_pending[0]:  # the previous output's mesh: its gather ran right behind that output's kernels
            # collect on this thread (a wait for the gather's ticket), then the CONSUMER's thread copies the 20 MB out of the pinned
            # staging block into its own arrays, block by block in sorted order (khr_fetch_mesh_into: host memcpy, no device access;
            # ctypes drops the GIL) while this thread goes on queueing frames
            import threading
            v_ = ctypes.c_int64(0)
            nv_ = ctx._chk(ctx.lib.khr_fetch_mesh(ctx.h, ctypes.byref(v_)))
            if nv_ > host_mesh_bufs.get("cap", 0):
                cap_ = int(nv_ * 1.5) + 1024
                host_mesh_bufs.update(cap=cap_, points=np.zeros((cap_, 3), np.float32), colors=np.zeros((cap_, 4), np.uint8),
                                      labels=np.zeros(cap_, np.uint32), first_seen=np.zeros(cap_, np.uint64), stamps=np.zeros(cap_, np.uint64))
            if nv_:
                ptrs_ = [host_mesh_bufs[k].ctypes.data_as(ctypes.c_void_p) for k in ("points", "colors", "labels", "first_seen", "stamps")]
                host_mesh_thread[0] = threading.Thread(target=lambda: ctx.lib.khr_fetch_mesh_into(ctx.h, *ptrs_))
                host_mesh_thread[0].start()
            copy_stats[1] += nv_ * (12 + 4 + 4 + 8 + 8)
            host_mesh_pending[0] = False
        _trace(b"hc_poll_mesh_done")
        if host_pending[0] is not None and host_pending[0].poll() and len(host_inflight) < 2:
            host_consumer_begin()
        _trace(b"hc_poll_exit")

    def host_consumer_drain():
        host_consumer_poll()
        host_mesh_copy_wait()
        if host_pending[0] is not None:
            host_consumer_begin()
        while host_inflight:
            host_consumer_retire()
    host_t = [0.0, 0.0, 0.0]  # host seconds in process_frame / finish_frame / launch_frame (incl. warm-up)
    obj_stats = [0, 0, 0.0]  # objects extracted, tracks removed, seconds spent in extraction (timed region and warm-up)

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.output_copy == "host" and world == 1:
        # a consumer that takes the mesh at every output sizes its buffers once: the library's pinned staging block and the
        # consumer's own arrays would otherwise grow (x 1.5) in the middle of the run -- a re-allocation of page-locked memory
        # costs tens of milliseconds (seen as ONE 40 ms step in a 40-step window, profiles/r05_host_consumer_40.txt)
        cap_ = int(min(cfg.max_mesh_vertices, 6 << 20))
        ctx.reserve_mesh_staging(cap_)
        # ... and the snapshot arenas of the outputs it keeps in flight (two being downloaded, one pending, one being taken): an
        # arena that is allocated when a snapshot finds none free is a hipMalloc of 0.8 GB -- ONE 35 - 45 ms step
        ctx.reserve_snapshots(4, fields=63, cap_blocks=int(cfg.max_snapshot_blocks) or 8192)
        host_mesh_bufs.update(cap=cap_, points=np.zeros((cap_, 3), np.float32), colors=np.zeros((cap_, 4), np.uint8),
                              labels=np.zeros(cap_, np.uint32), first_seen=np.zeros(cap_, np.uint64), stamps=np.zeros(cap_, np.uint64))
        for k_ in ("points", "colors", "labels", "first_seen", "stamps"):
            host_mesh_bufs[k_].fill(0)  # (touch the pages)
    # The interpreter's cyclic garbage collector is a property of this harness, not of the path: a generation-2 pass in the middle of
    # the timed steps was a 35 - 60 ms step in the host-consumer windows (profiles/r05_host_consumer_40.txt).  It is run once and
    # switched off HERE, in front of the pre-roll -- not between warm-up and timed steps, where the pause lets the device's clocks
    # drop and the first timed steps pay for it (-4.7 % on the driver's command, tools/runs/r05/run28.sh / run29.sh).
    import gc
    gc.collect()
    gc.disable()
    for i in range(t0i):  # pre-roll + warm-up, untimed
        step(i)
    sync_all()
    copy_before = list(copy_stats)
    obj_before = list(obj_stats)
    if pipe is not None:
        obj_before[0] += 0  # (detached extractions finishing later are attributed to the region that joins them)
    st0 = ctx.stats()
    if not args.no_roofline_timers:
        ctx.timing_reset()
        # default: only the two update kernels carry HIP events (each timed launch costs host time);
        # --all-timers adds the per-kernel breakdown
        ctx.timing_enable(True, None if args.all_timers else ("tsdf", "band"))  # HIP events cost a barrier packet each: only the roofline kernel
    if fusion_cxx is not None and not emu:
        fusion_cxx.profile(False)  # (reset: calls / bytes of the timed steps are counted, nothing is timed)
    _trace(_tags["timed_begin"])
    t0 = time.perf_counter()
    ft = []
    step_t = [t0]  # host clock after every step (no synchronisation: the host meets the device once per frame at the seed count)
    for i in range(t0i, t1i):
        if args.frame_times:
            torch.cuda.synchronize()
            tf = time.perf_counter()
        step(i)
        step_t.append(time.perf_counter())
        if args.frame_times:
            torch.cuda.synchronize()
            ft.append((i, round(1e6 * (time.perf_counter() - tf)), ctx.stats()["n_seeds"]))
    if os.environ.get("KHR_BENCH_HOST_TIMES") and rank == 0:
        print("host us/frame: process_frame %.1f finish_frame %.1f launch_frame %.1f" % tuple(1e6 * t / max(1, n_total) for t in host_t),
              file=sys.stderr)
    if args.frame_times and rank == 0:
        print("frame_times(us, seeds):", ft, file=sys.stderr)
    _trace(_tags["join_begin"])
    t_join0 = time.perf_counter()
    if pipe is not None:
        pipe.finish_frame()
        obj_stats[0] += pipe.join()  # detached object extractions still running on the worker thread / its stream
    if args.output_copy == "host" and world == 1:
        host_consumer_drain()  # the last outputs' transfers end inside the timed region
    sync_all()
    dt = time.perf_counter() - t0
    # how the timed region splits: queueing the K steps (the host runs ahead of the GPU by at most the motion detector's seed
    # count) and the wait at the end for the device and for the detached object extractions that the steps started
    timed_split = {"steps_ms": 1e3 * (t_join0 - t0), "drain_and_join_ms": 1e3 * (t0 + dt - t_join0)}
    # What the reference's own timer would show: its `active_window/all` scope is the spinOnce body (active_window.cpp:121), and with
    # detach_object_extraction (the default of khronos_ros/config) the extraction workers run outside it (:239-242, object_worker_pool.cpp:
    # 115-146).  `value` is stricter: it also waits for the device and for every extraction the timed steps started.
    timed_split["spin_once_bodies_only"] = {"frames_per_s": args.steps / max(t_join0 - t0, 1e-9), "ms_per_step": 1e3 * (t_join0 - t0) / args.steps,
                                            "note": "host wall time of the K step calls alone (the reference's active_window/all scope with detached "
                                                    "object extraction); NOT the headline: `value` includes the drain"}
    timed_split["frames_with_dynamic_clusters"] = sum(1 for i in range(t0i, t1i) if dyn_log.get(i, 0) > 0)
    timed_split["outputs"] = sum(1 for i in range(t0i, t1i) if args.output_every > 0 and (i + 1) % args.output_every == 0)
    timed_split["objects_extracted"] = obj_stats[0] - obj_before[0]
    if len(step_t) >= 9:
        # the window fills during the run (more blocks, more tracks, object extractions beside the frames): the steps get heavier, which
        # is why a longer timed region has a higher ms_per_step (host view, un-synchronised: a step's time is the device's, one frame late)
        sd = np.diff(np.array(step_t)) * 1e3
        q = max(1, len(sd) // 4)
        timed_split["step_ms_host_view"] = {"first_quarter_mean": float(sd[:q].mean()), "last_quarter_mean": float(sd[-q:].mean()),
                                            "median": float(np.median(sd)), "max": float(sd.max()), "argmax_step": int(sd.argmax())}
    _trace(_tags["timed_end"])
    gc.enable()
    ctx.timing_enable(False)
    st1 = ctx.stats()
    obj_timed = [obj_stats[k] - obj_before[k] for k in range(3)]
    copy_timed = [copy_stats[k] - copy_before[k] for k in range(2)]
    # per-frame latency: the same steps with a device synchronisation after each (the reference's active_window/all scope
    # is a per-frame wall time; `value` above is pipelined throughput: the host queues frame i + 1 while frame i executes)
    lat_ms = None
    # the other streaming kernels of the path are timed on these frames (start / stop stamps of their own dispatch packets:
    # nothing is added to the streams) and the unit counts of SURVEY.md 8(d) are read after every frame
    kern_names = ("k_tracking_update", "k_ever_free", "k_mc_count", "k_mc_emit", "k_snapshot_pack")
    kern_units = {"trk_vox": 0, "ef_vox": 0, "mesh_vox": 0, "mesh_vertices": 0, "snap_blocks": 0, "frames": 0, "outputs": 0}
    if lat > 0:
        tl = []
        if world == 1 and not args.no_roofline_timers:
            ctx.timing_enable(True, kern_names)  # (their totals start at zero: they were off until here; k_fuse's totals stay)
        for i in range(t1i, t1i + lat):
            torch.cuda.synchronize()
            ta = time.perf_counter()
            step(i)
            if pipe is not None:
                pipe.finish_frame()
            torch.cuda.synchronize()
            tl.append(1e3 * (time.perf_counter() - ta))
            if world == 1 and not args.no_roofline_timers:
                sk = ctx.stats()
                kern_units["frames"] += 1
                kern_units["trk_vox"] += sk["n_tracking_processed_blocks"] * 4096
                kern_units["ef_vox"] += sk["n_tracking_updated_blocks"] * 4096
                if args.output_every > 0 and (i + 1) % args.output_every == 0:
                    kern_units["outputs"] += 1
                    kern_units["mesh_vox"] += sk["n_mesh_blocks"] * 4096
                    kern_units["mesh_vertices"] += sk["n_mesh_vertices"]
                    if held_snapshot[0] is not None:
                        kern_units["snap_blocks"] += held_snapshot[0].num_blocks()
        if world == 1 and not args.no_roofline_timers:
            kern_ms = {k: ctx.timing_get(k) for k in kern_names}
            ctx.timing_enable(False)
        if pipe is not None:
            pipe.join()
        lat_ms = {"mean": float(np.mean(tl)), "min": float(np.min(tl)), "max": float(np.max(tl)), "frames": len(tl)}
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    rccl_info = None
    if fusion_cxx is not None and not emu:
        counted = fusion_cxx.profile_get()  # the timed steps: calls and bytes (no events)
        fusion_cxx.profile(True)
        p0 = t1i + lat
        for i in range(p0, p0 + prof_ticks):  # every rank runs them: the calls are collective
            step(i)
        if pipe is not None:
            pipe.finish_frame()
            pipe.join()
        sync_all()
        timed_c = fusion_cxx.profile_get()
        fusion_cxx.profile(False)
        rccl_info = {
            "rccl_ranks": world, "library": os.environ.get("KDIST_RCCL_LIB", "librccl.so.1"),
            "frames_exchange": "torch.distributed all_gather_into_tensor on a side stream, one tick ahead (%d B per rank and tick)" % (11 * W * H)
                               if not sender_ingest else "inside the tick (converted planes, kdist_tick_own)",
            "timed_steps": {k: {"calls": v["calls"], "bytes_sent_per_rank": v["bytes_sent"]} for k, v in counted.items() if v["calls"]},
            "profiled_ticks": prof_ticks,
            "mesh_halo_last_output_bytes": fusion_cxx.last_mesh_exchange(),
            "collectives": {k: {"calls": v["calls"], "bytes_sent_per_call": v["bytes_sent"] / v["calls"], "ms_per_call": v["ms"] / v["calls"]}
                            for k, v in timed_c.items() if v["calls"]},
            "note": "ms: HIP events recorded around each collective on the tick's stream during %d extra ticks after the timed region "
                    "(includes the wait for the slowest rank to arrive); rank 0's view" % prof_ticks}

    n_upd = st1["cum_updated_voxels"] - st0["cum_updated_voxels"]
    n_band = st1["cum_band_voxels"] - st0["cum_band_voxels"]
    n_vis = st1["cum_visited_voxels"] - st0["cum_visited_voxels"]
    n_calls = st1["cum_integrate_calls"] - st0["cum_integrate_calls"] + 0
    if dist is not None:
        t = torch.tensor([n_upd, n_band, n_vis], dtype=torch.float64, device=dev)
        dist.all_reduce(t)
        n_upd_all, n_band_all, n_vis_all = [float(x) for x in t.tolist()]
    else:
        n_upd_all, n_band_all, n_vis_all = float(n_upd), float(n_band), float(n_vis)

    frames = args.steps * world  # camera frames fused by the whole job
    # the label names a BASELINE config only when the arguments ARE that config
    ref = {"c3": (1280, 720, 0.02, 4, False, False, False), "c2": (640, 480, 0.05, 1, True, True, False),
           "c1": (640, 480, 0.05, 0, True, True, True), "c4": (1280, 720, 0.02, 4, False, False, False),
           "c5": (1920, 1080, 0.01, 4, False, False, False)}[args.config]
    preset_matches = (W, H, vs, args.output_every, bool(args.no_motion), bool(args.no_objects), bool(args.no_tracking)) == ref and K == 20
    preset_name = ({"c3": "BASELINE configs[2]: ", "c2": "BASELINE configs[1] (synthetic stand-in for the tesse_cd_office replay): ",
                    "c1": "BASELINE configs[0] (projective integrator alone): ",
                    "c4": "BASELINE configs[3] geometry (per camera; %d camera(s) here): " % world,
                    "c5": "BASELINE configs[4] geometry (per camera; %d camera(s) here): " % world}[args.config]) if preset_matches else "custom: "
    fps = frames / dt
    out = {
        "metric": "active-window frames/sec (+ Mvoxel-updates/sec) at %dx%d RGB-D+labels, %g cm voxels" % (W, H, vs * 100),
        "latency_ms_per_frame": lat_ms,
        "timed_region": timed_split,
        "value": fps, "unit": "frames/s", "n_gpus": 1 if emu else world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s%dx%d synthetic RGB-D+labels, %g cm voxels, truncation %g cm, vps 16, K=%d labels, "
                               "MotionDetector %s, tracking integrator %s, object detection / tracking / extraction %s, %s; %d camera(s); "
                               "%d pre-roll + %d warm-up frames before the timed steps; %s arithmetic; %s"
                               % (preset_name, W, H, vs * 100, trunc * 100, K, "off" if args.no_motion else "on",
                                  "off" if args.no_tracking else "on",
                                  "off" if args.no_objects else "on (ConnectedSemantics, MaxIoUTracker, MeshObjectExtractor)",
                                  ("output (mesh, archival%s) every %d frame(s)" % ("" if args.no_objects else ", object extraction", args.output_every))
                                  if args.output_every > 0 else "no output stage", world, pre, args.warmup,
                                  "fast (decisions exact, values ~1e-6)" if args.fast else "exact (bit-identical to the CPU restatement)",
                                  "inputs in page-locked host memory (PCIe in the timed region)" if host_input else "inputs resident in HBM"),
                   "preset": args.config if preset_matches else None,
                   "parallelism": "hash-range block shard x%d, owner-computes, RCCL all-gather of packed frames (prefetched one tick ahead on its own stream) + of 528-B halo records "
                                  "(ever-free), seed-gated all-reduce of per-pixel voxel keys (motion detector), mesh halo per output: all-gather of the bucketed plane requests, "
                                  "all-to-all-v of per-relation face / line / voxel answers to the requester alone (--dist-host torch: all-gather of whole-block records)" % world
                   if world > 1 else "single GPU",
                   "collectives_issued_by": {"cxx": "libkhronos_amd_host.so (kdist_*: rccl calls on the context's HIP stream)",
                                             "torch": "khronos_amd/distributed.py (torch.distributed)", "emulated": "none (emulation, torch harness)",
                                             "cxx-emulated": "libkhronos_amd_host.so (kdist_*, KDIST_EMULATE: rank 0 alone, collectives skipped)",
                                             "none": None}[dist_host],
                   "ingest": ("sender side (kdist_tick_own): every rank converts its own camera's frame, the converted planes (12 B / pixel) "
                              "are all-gathered and adopted in place" + (" -- emulated: the other cameras' planes were converted before the timed region" if emu else ""))
                   if sender_ingest else ("every rank converts every camera's raw frame" if world > 1 else "single camera")},
        "mvoxel_updates_per_s": 1e-6 * n_upd_all / dt,
        **({"rccl": rccl_info} if rccl_info is not None else {}),
        **({"emulation": "rank 0 of a %d-rank sharded run played by one process, no collectives: `value` is what the job would reach "
                         "if communication were free and all ranks were as loaded as rank 0 -- NOT a measured N-GPU number" % world}
           if emu else {}),
        "input_lookahead": bool(lookahead),
        "input": {"where": args.input,
                  "what": "frames resident in HBM before the timed region (khr_process_frame(on_device = 1))" if not host_input else
                          "frames in page-locked HOST memory: khr_process_frame(on_device = 0, KHR_PF_INPUT_PINNED); the planes of frame i + 1 are "
                          "handed over with khr_ingest_ahead_host and travel on the context's copy stream while frame i is fused",
                  "host_bytes_per_frame": (11 * W * H) if host_input else 0,
                  "h2d_GBps_sustained": (11.0 * W * H * fps / 1e9) if host_input else None},
        "output_copy": {"mode": args.output_copy,
                        "what": {"none": "the timed steps hand out no clone of the updated blocks (frames and map stay in HBM)",
                                 "device": "every output takes a device-side snapshot of the updated blocks (khr_snapshot_updated between meshing "
                                           "and archival: VolumetricMap::cloneUpdated, active_window.cpp:229), held until the next output",
                                 "host": "every output takes the device-side snapshot AND a host consumer downloads %s of every updated "
                                         "block plus the mesh -- pipelined: the transfer of output k runs on a copy stream beside the frames "
                                         "after it (khr_snapshot_download_begin / _end, two sets of pinned buffers), the mesh gather is queued "
                                         "behind the output and collected one frame later"
                                         % ("distance + weight (8 B / voxel)" if args.host_fields == "tsdf" else "all six layers (25 B / voxel)")}[args.output_copy],
                        "outputs_in_timed_region": copy_timed[0], "host_bytes_in_timed_region": copy_timed[1],
                        "host_bytes_per_output": (copy_timed[1] / copy_timed[0]) if (copy_timed[0] and args.output_copy == "host") else None},
        "objects": None if pipe is None else {"tracks_at_end": pipe.num_tracks(), "buffered_frames": pipe.num_buffered_frames(),
                                              "objects_extracted": obj_timed[0], "tracks_removed": obj_timed[1],
                                              "extraction_ms_total": 1e3 * obj_timed[2],
                                              "objects_extracted_before_timed_region": obj_before[0]},
        "voxels": {"visited": n_vis_all, "updated": n_upd_all, "band": n_band_all, "allocated_blocks": st1["n_allocated_blocks"],
                   "last_frame_visible_blocks": st1["n_visible_blocks"], "last_frame_tsdf_blocks": st1["n_tsdf_blocks"],
                   "last_frame_fuse_items": st1["n_fuse_items"],
                   "band_overflow": st1["band_overflow"], "last_frame_tracking_blocks": st1["n_tracking_processed_blocks"],
                   "last_frame_touched_blocks": st1["n_tracking_updated_blocks"]},
    }

    # the per-frame host / device meeting, watched (khr_stats: recorded by the library, never printed): waits for the motion detector's seed
    # count over the whole run (pre-roll, warm-up, timed steps, latency frames), the late ones (> 0.5 ms) with the queues' state
    out["seed_wait"] = {"waits": st1["n_seed_waits"], "late_over_500us": st1["n_seed_waits_late"], "max_us": st1["seed_wait_max_us"],
                        "hist_us_50_100_200_500_1k_2k_5k_more": st1["seed_wait_hist"],
                        "last_late": {"us": st1["seed_wait_late_us"], "wait_no": st1["seed_wait_late_frame"], "state_bits": st1["seed_wait_late_state"]},
                        "in_timed_steps": {"waits": st1["n_seed_waits"] - st0["n_seed_waits"], "late_over_500us": st1["n_seed_waits_late"] - st0["n_seed_waits_late"]},
                        "motion_merges_on_device": st1["n_md_device_merges"], "motion_host_walks": st1["n_md_host_walks"],
                        "motion_chains_queued_ahead": st1["n_md_prelaunched"], "motion_chains_repeated": st1["n_md_prelaunch_repeats"]}
    # ---- roofline of the dominant kernel (k_fuse: the fused TSDF / colour / label update), from HIP events on the
    #      kernel's own dispatch packets (hipExtLaunchKernelGGL start / stop events on the kernel's stream) ----
    if not args.no_roofline_timers and rank == 0:
        # the update step = k_fuse (voxel phase; also the band phase when KHR_FUSE_SPLIT=0) + k_band (round 4: the in-band voxels'
        # colour / label / likelihood update as its own launch): both launches count, the algorithmic bytes are those of the step
        ms_fuse, launches = ctx.timing_get("tsdf")
        ms_band, launches_band = ctx.timing_get("band")
        ms = ms_fuse + ms_band
        nl = max(1, launches)
        # ALGORITHMIC bytes of the TSDF-update kernel, SURVEY.md section 8(d):
        #   24 B per updated voxel (R+W distance, R+W weight, W last_observed)
        # + (12 + 8K) B per in-band voxel (R+W colour, R+W K likelihoods, W label)
        # + the images once per launch: depth 4 + label 4 + rgb 3 (+ mask 4 with the motion detector) B per pixel
        px_bytes = 11.0 + (0.0 if args.no_motion else 4.0)
        bytes_total = 24.0 * n_upd + (12.0 + 8.0 * K) * n_band + px_bytes * W * H * nl
        achieved = bytes_total / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("kernel") == "k_fuse" and tj.get("workload") == [W, H, vs]:
                    traffic = tj.get("k_fuse_bytes_per_launch")
                    traffic_src = "profiles/pmc_traffic.json (rocprofv3 --pmc passes of this command, %s; NOT measured in this run)" % tj.get("collected", "?")
            except Exception:
                traffic = None
        out["roofline"] = {"kernel": "k_fuse + k_band (the TSDF / colour / label update step)" if launches_band else "k_fuse", "bound": "hbm",
                           "achieved": achieved, "peak": 8000.0, "k_fuse_avg_us": 1e3 * ms_fuse / nl,
                           "k_band_avg_us": (1e3 * ms_band / max(1, launches_band)) if launches_band else None,
                           "unit": "GB/s", "frac": achieved / 8000.0, "traffic": traffic, "traffic_source": traffic_src,
                           "avg_launch_us": 1e3 * ms / nl, "launches": launches,
                           "algorithmic_bytes_per_launch": bytes_total / nl,
                           "algorithmic_bytes": "24 B x N_upd + (12 + 8K) B x N_band + %g B x W x H per launch (SURVEY.md 8(d))" % px_bytes,
                           "frac_of_measured_copy_peak_6290": achieved / 6290.0}
        # Mvoxel-updates/s = N_upd / sum of the TSDF-update step's time (the whole-frame figure is above)
        out["mvoxel_updates_per_s_tsdf_step"] = 1e-6 * n_upd / (ms * 1e-3) if ms > 0 else None
        kern = {}
        for name in ("tsdf", "band", "tracking", "ever_free", "alloc", "motion_pixels", "mesh", "parse"):
            m_, n_ = ctx.timing_get(name)
            kern["fuse" if name == "tsdf" else name] = {"ms_total": m_, "launches": n_}
        out["kernel_ms"] = kern
        if world == 1 and lat > 0 and kern_units["frames"]:
            # ---- the path's other streaming kernels against the same roofline (SURVEY.md 8(d) bytes per unit) ----
            def krec(kernel, timers, nbytes, formula, counts, per):
                ms_ = sum(kern_ms[t][0] for t in timers)
                n_ = max(kern_ms[timers[0]][1], 1)
                if ms_ <= 0:
                    return None
                us = 1e3 * ms_ / n_
                ach = nbytes / n_ / (us * 1e-6) / 1e9
                return {"kernel": kernel, "bound": "hbm", "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0,
                        "avg_launch_us": us, "launches": n_, "per": per, "algorithmic_bytes_per_launch": nbytes / n_,
                        "algorithmic_bytes": formula, "counts_per_launch": {k: v / n_ for k, v in counts.items()},
                        **({"passes_us": {t: 1e3 * kern_ms[t][0] / max(kern_ms[t][1], 1) for t in timers}} if len(timers) > 1 else {})}
            ku = kern_units
            nv_, nf_ = ku["mesh_vertices"], ku["mesh_vertices"] / 3.0
            recs = [
                krec("k_tracking_update", ("k_tracking_update",), 31.0 * ku["trk_vox"],
                     "31 B x N_alloc, N_alloc = voxels of the blocks the pass has to visit (the blocks that provably cannot change are "
                     "skipped by k_tracking_select and cost nothing)", {"N_alloc_vox": ku["trk_vox"]}, "frame"),
                krec("k_ever_free", ("k_ever_free",), 18.0 * ku["ef_vox"], "18 B x N_updblk_vox", {"N_updblk_vox": ku["ef_vox"]}, "frame"),
                krec("k_marching_cubes", ("k_mc_count", "k_mc_emit"), 8.0 * ku["mesh_vox"] + 48.0 * nv_ + 12.0 * nf_,
                     "8 B x N_meshblk_vox + 48 B x N_vertices + 12 B x N_faces (count pass + emit pass together; the soup has N_faces = N_vertices / 3)",
                     {"N_meshblk_vox": ku["mesh_vox"], "N_vertices": nv_, "N_faces": nf_}, "output"),
                krec("k_snapshot_pack", ("k_snapshot_pack",), 2.0 * 25.0 * 4096.0 * ku["snap_blocks"],
                     "2 x 25 B x voxels of the updated blocks (read + write of distance, weight, colour, label, last_observed, flags: "
                     "VolumetricMap::cloneUpdated, active_window.cpp:229; not a row of SURVEY.md 8(d))",
                     {"N_snapshot_blocks": ku["snap_blocks"]}, "output") if ku["snap_blocks"] else None,
            ]
            out["kernel_rooflines"] = {"measured_on": "the %d latency frames after the timed region (%d outputs); HIP start / stop stamps of each kernel's own "
                                                      "dispatch packet (hipExtLaunchKernelGGL)" % (ku["frames"], ku["outputs"]),
                                       "kernels": [r for r in recs if r]}

    # ---- CPU baseline: the oracle on a bounded sample of the same stream (rank 0, N = 1 only) ----
    nb = args.cpu_baseline_frames
    if rank == 0 and world == 1 and nb != 0:
        from oracle import pyoracle as po
        cores = os.cpu_count() or 1
        if nb < 0:
            nb = max(4, min(16, args.steps)) if args.config == "c3" else min(24, args.steps)
        ocfg = po.config_from(cfg, cores)
        ora = po.OracleMap(ocfg)
        osen = ora.make_sensor(W, H, s.fx, s.fy, s.cx, s.cy, 0.1, 5.0)
        tc = 0.0
        upd = 0
        # The SAME frames the GPU steps were timed on: the oracle first fuses the pre-roll and warm-up frames untimed (the
        # window is then in the same steady state), then frames t0i .. t0i + nb - 1 are timed.  When that would take too
        # long (custom runs with long pre-rolls) the sample falls back to the first frames of the stream and says so.
        same_frames = (t0i + nb) <= len(frames_host) and t0i <= 128
        first_timed = t0i if same_frames else min(2, nb - 1)
        last_frame = first_timed + nb if same_frames else nb
        # per-stage wall time under the reference's timer names (hydra::timing scopes of active_window.cpp:121-256,
        # free_space_motion_detector.cpp:74, connected_semantics.cpp:60, max_iou_tracker.cpp:199, tracking_integrator.cpp:72)
        stage_names = ("motion_detection/all", "active_window/update_map (projective integrator)", "integration/tracking",
                       "object_detection/all", "tracking/all (voxel sets)", "active_window/extract_output (mesh, archival)")
        stage = dict.fromkeys(stage_names, 0.0)

        def cpu_frame(i, acc):
            fr = frames_host[i]
            tt = [time.perf_counter()]
            dyn = None
            if not args.no_motion:
                _, dyn, _ = ora.detect_motion(osen, fr["stamp"], fr["pose"], fr["depth"])
            tt.append(time.perf_counter())
            stc = ora.integrate(osen, fr["stamp"], fr["pose"], fr["depth"], fr["rgb"], fr["label"], mask=dyn)
            tt.append(time.perf_counter())
            if not args.no_tracking:
                ora.update_tracking(fr["stamp"])
            tt.append(time.perf_counter())
            t_obj = t_trk = 0.0
            if pipe is not None and acc is not None:
                # object half on the CPU: ConnectedSemantics + the tracker's voxel sets (single-threaded in the reference
                # too); the association itself is negligible and not timed.  (Stateless per frame: skipped on the
                # untimed lead-in.)
                _, oimg, _ = ora.detect_objects(osen, fr["stamp"], fr["pose"], fr["depth"], fr["label"], list(range(7, 20)), use_3d=True,
                                                grid_size=0.1, max_range=5.0, min_cluster_size=50, use_full_connectivity=True)
                t_obj = time.perf_counter() - tt[-1]
                tv = time.perf_counter()
                ora.cluster_voxels(osen, fr["stamp"], fr["pose"], fr["depth"], oimg, 0.2)
                if dyn is not None and dyn.any():
                    ora.cluster_voxels(osen, fr["stamp"], fr["pose"], fr["depth"], dyn, 0.2)
                t_trk = time.perf_counter() - tv
            to = time.perf_counter()
            if args.output_every > 0 and (i + 1) % args.output_every == 0:
                ora.generate_mesh(True, True)
                if not args.no_tracking:
                    ora.reset_inactive()
                ora.clear_updated()
            t_end = time.perf_counter()
            if acc is not None:
                for k, v in zip(stage_names, (tt[1] - tt[0], tt[2] - tt[1], tt[3] - tt[2], t_obj, t_trk, t_end - to)):
                    acc[k] += v
            return t_end - tt[0], stc["n_updated_voxels"]

        for i in range(first_timed):
            cpu_frame(i, None)
        # thread-count sweep on the first frames of the sample (the integrators spawn and join their threads
        # per call, as the reference's do -- tracking_integrator.cpp:83-90 -- so "all cores" is not the fastest setting on a
        # many-core host): 2 frames per setting, then the rest of the sample at all cores
        sweep = {}
        i = first_timed
        sweep_counts = [t for t in (8, 32, 64) if t < cores]
        if same_frames and nb >= 2 * len(sweep_counts) + 4:
            for tcount in sweep_counts:
                ora.set_threads(tcount)
                ts = 0.0
                for _k in range(2):
                    dt_, _u = cpu_frame(i, None)
                    ts += dt_
                    i += 1
                sweep[str(tcount)] = 2.0 / ts
            ora.set_threads(cores)
        n_all = 0
        for i in range(i, last_frame):
            dt_, u_ = cpu_frame(i, stage)
            tc += dt_
            upd += u_
            n_all += 1
        n_timed = n_all
        cpu_fps = n_timed / tc
        sweep[str(cores)] = cpu_fps
        best_threads = max(sweep, key=lambda k: sweep[k])
        out["cpu_baseline"] = {"value": cpu_fps, "unit": "frames/s", "cores": cores, "kind": "port",
                               "sample": "frames %d..%d of the same stream%s (CPU restatement of the reference path, %d threads for "
                                         "the volumetric part%s; reference itself not buildable offline)"
                                         % (last_frame - n_timed, last_frame - 1,
                                            " = %d of the frames the GPU steps were timed on, after the same lead-in frames" % n_timed
                                            if same_frames else " (the window is still filling: fewer blocks per frame than in the GPU's timed steps)",
                                            cores, ", object detection + voxel sets on 1 thread as in the reference" if pipe is not None else ""),
                               "mvoxel_updates_per_s": 1e-6 * upd / tc,
                               "stage_ms_per_frame": {k: 1e3 * v / n_timed for k, v in stage.items()},
                               "thread_sweep_frames_per_s": sweep,
                               "best": {"threads": int(best_threads), "frames_per_s": sweep[best_threads],
                                        "note": "2 frames per setting below all cores; worker threads are spawned and joined per integrator call, as in the reference"}}
        out["speedup_vs_cpu"] = fps / cpu_fps
        out["speedup_vs_cpu_best_thread_count"] = fps / sweep[best_threads]

    if held_snapshot[0] is not None:
        held_snapshot[0].release()
    if args.output_copy == "host" and world == 1:
        host_consumer_drain()
    ctx.close()
    # ---- the other streams north_star asks for, and what an output's map clone costs: short sub-runs of this script,
    #      appended to the default c3 line so that one driver invocation carries all of them ----
    if (rank == 0 and world == 1 and not emu and not args.no_extra_streams and args.config == "c3" and preset_matches
            and args.output_copy == "device" and not args.fast):
        import subprocess
        extra = {}
        common_args = ["--steps", str(args.steps), "--warmup", str(args.warmup), "--no-extra-streams", "--latency-frames", "0"]
        runs = {"c1": ["--config", "c1"], "c2": ["--config", "c2"],
                "c3_output_copy_none": ["--config", "c3", "--output-copy", "none", "--cpu-baseline-frames", "0"],
                "c3_output_copy_host": ["--config", "c3", "--output-copy", "host", "--cpu-baseline-frames", "0"],
                "c3_output_copy_host_all_layers": ["--config", "c3", "--output-copy", "host", "--host-fields", "all", "--cpu-baseline-frames", "0"],
                # what a drop-in sees: the input arrives in host memory; and input from host + output to host
                "c3_input_host": ["--config", "c3", "--input", "host", "--cpu-baseline-frames", "0"],
                "c3_io_host": ["--config", "c3", "--input", "host", "--output-copy", "host", "--cpu-baseline-frames", "0"]}
        for name, extra_args in runs.items():
            r = None
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__)] + extra_args + common_args, capture_output=True, text=True,
                                   timeout=600)
                j = json.loads([ln for ln in r.stderr.strip().splitlines() if ln.startswith("{")][-1])  # (the full record is on stderr)
                keep = {k: j.get(k) for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "mvoxel_updates_per_s",
                                              "speedup_vs_cpu", "output_copy", "input")}
                keep["workload"] = j["config"]["workload"]
                if "roofline" in j:
                    keep["roofline"] = {k: j["roofline"].get(k) for k in ("kernel", "achieved", "peak", "unit", "frac", "avg_launch_us")}
                if "cpu_baseline" in j:
                    keep["cpu_baseline"] = j["cpu_baseline"]
                extra[name] = keep
            except Exception as e:  # noqa: BLE001
                extra[name] = {"error": "%s: %s" % (type(e).__name__, e), "stderr_tail": (r.stderr[-600:] if r is not None else None)}
        # khronos::ActiveWindow::spinOnce itself (the C++ plugin class of libkhronos_amd_host.so), same stream, same window
        r = None
        try:
            import tempfile
            from khronos_amd.configs import ACTIVE_WINDOW_YAML
            with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as tf:
                tf.write(ACTIVE_WINDOW_YAML % dict(vs=args.voxel_size, trunc=3 * args.voxel_size, buf=args.buffer_frames, labels=K, max_blocks=args.max_blocks))
            demo = os.path.join(ROOT, "khronos_amd", "lib", "aw_demo")
            r = subprocess.run([demo, "--bench", tf.name, str(W), str(H), str(pre), str(args.warmup), str(args.steps)], capture_output=True, text=True, timeout=900)
            os.unlink(tf.name)
            j = json.loads(r.stdout.strip().splitlines()[-1])
            j["value"] = j["frames_per_s"]
            j["vs_headline"] = j["frames_per_s"] / fps
            extra["cxx_active_window"] = j
        except Exception as e:  # noqa: BLE001
            extra["cxx_active_window"] = {"error": "%s: %s" % (type(e).__name__, e), "stderr_tail": (r.stderr[-600:] if r is not None else None)}
        out["streams"] = extra
    if rank == 0:
        # The driver parses the LAST stdout line: a compact record (< 6 KB).  Everything else -- sub-streams, per-kernel rooflines,
        # stage tables, notes -- goes to bench_detail.json next to this script and to stderr.
        detail_path = os.environ.get("KHR_BENCH_DETAIL", os.path.join(ROOT, "bench_detail.json"))
        if not args.no_extra_streams or os.environ.get("KHR_BENCH_DETAIL"):
            try:
                with open(detail_path, "w") as f:
                    json.dump(out, f, indent=1)
            except OSError as e:
                print("bench_detail.json not written: %s" % e, file=sys.stderr)
        print(json.dumps(out), file=sys.stderr)
        sys.stderr.flush()
        print(json.dumps(compact_line(out)))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
