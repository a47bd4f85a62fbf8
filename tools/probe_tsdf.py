"""development probe: per-workgroup timeline of k_tsdf_update (KHR_DEBUG=8)."""
import os, sys, time
os.environ.setdefault("KHR_DEBUG", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from khronos_amd import FusionContext, default_config
from khronos_amd.synth import SyntheticStream
W, H, vs = 1280, 720, 0.02
cfg = default_config(voxel_size=vs, truncation_distance=3 * vs, with_semantics=1, num_labels=20, max_blocks=40960,
                     max_frame_pixels=W * H, max_mesh_vertices=1 << 20)
ctx = FusionContext(cfg)
s = SyntheticStream(W, H)
sen = ctx.make_sensor(W, H, s.fx, s.fy, s.cx, s.cy)
ctx.timing_enable(True, ["tsdf"])
for i in range(12):
    if i == 6:
        ctx.timing_reset()
    if i == 11:
        ctx.stats(); time.sleep(0.05)  # isolate the last launch in time
    fr = s.render(i)
    slot = ctx.upload_frame(sen, fr["stamp"], fr["pose"], fr["depth"], fr["rgb"], fr["label"])
    ctx.integrate(slot)
    ctx.update_tracking(fr["stamp"])
st = ctx.stats()
n = min(4096, st["n_tsdf_blocks"] * 4)
buf = np.zeros(4096 * 4 * 8, np.uint64)
ctx.lib.khr_debug_read(ctx.h, buf.ctypes.data, buf.size)
d = buf.reshape(4096, 4, 8)[:n].astype(np.int64)
# keep the entries of the LAST launch only (earlier frames had more work items; their probes are stale)
xcc0 = d[:, 0, 5] & 0xf
keep = np.zeros(len(d), bool)
for x in range(8):
    sel = xcc0 == x
    if sel.any():
        last = d[sel, 0, 3].max()  # the latest end on this XCD belongs to the last launch
        keep |= sel & (last - d[:, 0, 0] < 5_000_000)
print("entries", n, "of the last launch", int(keep.sum()))
d = d[keep]
n = len(d)
ms, cnt = ctx.timing_get("tsdf")
print("k_tsdf_update by its dispatch events: %.1f us avg over %d launches" % (1e3 * ms / max(cnt, 1), cnt))
xcc_all = d[:, 0, 5] & 0xf
start = np.zeros((n, 4), np.int64); p1 = start.copy(); rec = start.copy(); end = start.copy()
for x in range(8):
    sel = xcc_all == x
    if not sel.any():
        continue
    t0 = d[sel][:, :, 0].min()
    for arr, k in ((start, 0), (p1, 1), (rec, 2), (end, 3)):
        arr[sel] = d[sel][:, :, k] - t0
print("blocks", n, "kernel span (cycles, s_memtime @100MHz?)", end.max())
print("wave start  min/med/max", start.min(), np.median(start), start.max())
print("wave dur    min/med/max", (end - start).min(), np.median(end - start), (end - start).max())
print("pass1 dur   med", np.median(p1 - start), " rec dur med", np.median(rec - p1), " pass2 dur med", np.median(end - rec))
hw = d[:, 0, 4]
xcc = d[:, 0, 5] & 0xf
cu = (hw >> 8) & 0xf
se = (hw >> 13) & 0x7
sh = (hw >> 12) & 1
simd = (d[:, :, 4] >> 4) & 3
key = xcc * 1000 + se * 100 + sh * 50 + cu
u, cnt = np.unique(key, return_counts=True)
print("distinct (xcc,se,sh,cu):", len(u), "blocks per CU min/med/max", cnt.min(), np.median(cnt), cnt.max())
print("xcc histogram", np.bincount(xcc))
print("simd of the 4 waves of first 8 blocks", simd[:8].tolist())
# concurrency over time
ev = sorted([(t, 1) for t in start[:, 0]] + [(t, -1) for t in end[:, 0]])
c = 0; mx = 0
for t, e in ev:
    c += e; mx = max(mx, c)
print("max concurrent workgroups", mx)
order = np.argsort(start[:, 0])
print("start times of every 100th block:", start[order[::100], 0].tolist())
