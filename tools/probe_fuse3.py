"""development probe (round 5): per-wave timelines of the last k_fuse3 / k_band3 launch (KHR_FUSE_DBG=64: every wave leaves
{entry, exit (100 MHz realtime), items or rounds, records} in the context's debug buffer)."""
import os, sys
os.environ.setdefault("KHR_FUSE_DBG", "64")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from khronos_amd import FusionContext, default_config
from khronos_amd.synth import SyntheticStream

W, H, vs = 1280, 720, 0.02
n = int(sys.argv[1]) if len(sys.argv) > 1 else 26
cfg = default_config(voxel_size=vs, truncation_distance=3 * vs, with_semantics=1, with_tracking=1, num_labels=20, max_blocks=40960,
                     max_frame_pixels=W * H)
ctx = FusionContext(cfg)
s = SyntheticStream(W, H, seed=1234)
sen = ctx.make_sensor(W, H, s.fx, s.fy, s.cx, s.cy, 0.1, 5.0)
for i in range(n):
    fr = s.render(i)
    slot = ctx.upload_frame(sen, fr["stamp"], fr["pose"], fr["depth"], fr["rgb"], fr["label"])
    ctx.integrate(slot)
    ctx.update_tracking(fr["stamp"])
ctx.sync()
ctx.timing_reset()
ctx.timing_enable(True, ("tsdf", "band"))
fr = s.render(n)
slot = ctx.upload_frame(sen, fr["stamp"], fr["pose"], fr["depth"], fr["rgb"], fr["label"])
ctx.integrate(slot)
ctx.update_tracking(fr["stamp"])
ctx.sync()
print("last launch: k_fuse3 %.1f us, k_band3 %.1f us (dispatch-packet events)" % (1e3 * ctx.timing_get("tsdf")[0], 1e3 * ctx.timing_get("band")[0]))
st = ctx.stats()
print("items", st["n_fuse_items"], "n_upd", st["n_updated_voxels"], "n_band", st["n_band_voxels"], "band_overflow", st["band_overflow"])
buf = np.zeros(4096 * 4 * 12, np.uint64)
ctx.lib.khr_debug_read(ctx.h, buf.ctypes.data, buf.size)
q = [0, 10, 50, 90, 99, 100]
for name, lo in (("k_fuse3", 0), ("k_band3", 8192)):
    b = buf[lo * 4:(lo + 8192) * 4].reshape(-1, 4)
    b = b[b[:, 1] > 0]
    if not len(b):
        print(name, "no waves recorded")
        continue
    t0 = b[:, 0].min()
    start = (b[:, 0] - t0).astype(np.float64) / 100.0  # us
    end = (b[:, 1] - t0).astype(np.float64) / 100.0
    dur = end - start
    cnt = (b[:, 2] & np.uint64(0xffffffff)).astype(np.float64)
    units = (b[:, 2] >> np.uint64(32)).astype(np.float64)
    print(name, "waves", len(b), "kernel span (first entry -> last exit) %.1f us" % end.max())
    print("  entry pct", np.percentile(start, q))
    print("  exit  pct", np.percentile(end, q))
    print("  lifetime pct", np.percentile(dur, q), "mean %.2f" % dur.mean())
    print("  work units per wave pct", np.percentile(cnt, q), "total", cnt.sum(), "records", b[:, 3].sum())
    if units.sum() > 0:
        print("  band rounds run inside the kernel: per wave pct", np.percentile(units, q), "total", units.sum())
    wpw = 8 if name == "k_fuse3" else 4
    nwg = len(b) // wpw
    if nwg * wpw == len(b):
        we = end.reshape(nwg, wpw).max(1)
        print("  per-workgroup finish pct", np.percentile(we, q), "by XCD (wg % 8):", [round(float(we[np.arange(nwg) % 8 == x].mean()), 1) for x in range(8)])
    w = cnt > 0
    if w.any():
        print("  us per unit (waves with work): pct", np.percentile(dur[w] / cnt[w], q), "mean %.2f" % (dur[w] / cnt[w]).mean())
        print("  waves without work: %d, their lifetime mean %.2f us" % ((~w).sum(), dur[~w].mean() if (~w).any() else 0))
