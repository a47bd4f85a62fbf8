"""development probe: per-wave timeline of the last k_fuse launch (KHR_FUSE_DBG=64 selects the instrumented instantiation)."""
import os, sys
os.environ.setdefault("KHR_FUSE_DBG", "64")
os.environ.setdefault("KHR_FUSE_EXACT", "0")  # the instrumented instantiation is the relaxed-arithmetic one
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
print("last launch: k_fuse %.1f us, k_band %.1f us (dispatch-packet events)" % (1e3 * ctx.timing_get("tsdf")[0], 1e3 * ctx.timing_get("band")[0]))
st = ctx.stats()
buf = np.zeros(4096 * 4 * 12, np.uint64)
ctx.lib.khr_debug_read(ctx.h, buf.ctypes.data, buf.size)
b = buf[:4096 * 4 * 8].reshape(-1, 8)
acc = buf[4096 * 4 * 8:].reshape(-1, 4)[b[:, 1] > 0].astype(np.float64)
b = b[b[:, 1] > 0]
t0 = b[:, 0].min()
start = (b[:, 0] - t0).astype(np.float64)
end = (b[:, 1] - t0).astype(np.float64)
dur = end - start
band = b[:, 2].astype(np.float64)
items, rounds, recs, imax = b[:, 3], b[:, 4], b[:, 5], b[:, 6].astype(np.float64)
F = 2100.0  # s_memtime ticks per us: the counter runs at about the shader clock on gfx950 (durations only: XCDs have different bases)
print("tsdf blocks", st["n_tsdf_blocks"], "n_upd", st["n_updated_voxels"], "n_band", st["n_band_voxels"])
print("waves", len(b))
q = [0, 10, 50, 90, 99, 100]
print("dur    pct", np.percentile(dur, q) / F, "mean", dur.mean() / F)
print("band   pct", np.percentile(band, q) / F, "mean", band.mean() / F)
print("items  pct", np.percentile(items, q), "rounds pct", np.percentile(rounds, q), "recs total", recs.sum())
print("max item pct", np.percentile(imax, q) / F)
tr = rounds > 0
if tr.any():
    print("band ticks per round (waves with rounds): mean", (band[tr] / rounds[tr]).mean() / F, "us; pct", np.percentile(band[tr] / rounds[tr], q) / F)
nb = dur - band
print("non-band per item: mean", (nb / np.maximum(items, 1)).mean() / F, "us")
late = np.argsort(dur)[-8:]
for i in late:
    print("longest waves: dur %.1f band %.1f items %d rounds %d recs %d maxitem %.1f" % (dur[i] / F, band[i] / F, items[i], rounds[i], recs[i], imax[i] / F))


# ---- per-workgroup view (LDS item queue: WPW waves per workgroup, one workgroup per CU) ----
WPW = int(os.environ.get("KHR_PROBE_WPW", "12"))
nwg = len(dur) // WPW
if nwg * WPW == len(dur):
    dw = dur.reshape(nwg, WPW) / F
    iw = items.reshape(nwg, WPW)
    rw = rounds.reshape(nwg, WPW)
    wg_max = dw.max(1)
    wg_mean = dw.mean(1)
    print("workgroups", nwg, "per-WG max dur pct", np.percentile(wg_max, q), "mean of per-WG mean", wg_mean.mean())
    print("per-WG (max - min) wave dur pct", np.percentile(dw.max(1) - dw.min(1), q))
    print("per-WG items pct", np.percentile(iw.sum(1), q), "rounds pct", np.percentile(rw.sum(1), q))
    xcd = np.arange(nwg) % 8
    for x in range(8):
        print("xcd", x, "mean WG max dur %.1f" % wg_max[xcd == x].mean(), "items %.1f rounds %.1f" % (iw[xcd == x].sum(1).mean(), rw[xcd == x].sum(1).mean()))
    worst = np.argsort(wg_max)[-5:]
    for w in worst:
        print("slowest WG %d (xcd %d): max %.1f mean %.1f items %d rounds %d" % (w, w % 8, wg_max[w], wg_mean[w], iw[w].sum(), rw[w].sum()))
    best = np.argsort(wg_max)[:3]
    for w in best:
        print("fastest WG %d (xcd %d): max %.1f mean %.1f items %d rounds %d" % (w, w % 8, wg_max[w], wg_mean[w], iw[w].sum(), rw[w].sum()))

# ---- round 3: where does the launch time go?  start / end skew and per-XCD end times ----
print("wave start pct (us after the first wave)", np.percentile(start, q) / F)
print("wave end   pct", np.percentile(end, q) / F)
if nwg * WPW == len(dur):
    sw = (start.reshape(nwg, WPW) / F).min(1)
    ew = (end.reshape(nwg, WPW) / F).max(1)
    # (XCDs have different s_memtime bases: compare within an XCD only)
    for x in range(8):
        s0 = sw[xcd == x].min()
        print("xcd %d: WG start spread %.1f us, WG end (rel. to the xcd's first start) pct" % (x, sw[xcd == x].max() - s0),
              np.round(np.percentile(ew[xcd == x] - s0, q), 1), "records per WG pct", np.percentile(recs.reshape(nwg, WPW).sum(1)[xcd == x], [0, 50, 100]))
    rsum = recs.reshape(nwg, WPW).sum(1)
    print("corr(WG max dur, WG records) = %.2f, corr(WG max dur, WG rounds) = %.2f" % (np.corrcoef(wg_max, rsum)[0, 1], np.corrcoef(wg_max, rw.sum(1))[0, 1]))

# ---- round 4: one clock for all XCDs (s_memrealtime, 10 ns): when do waves enter, when are they past the start-up chain, when do they leave ----
rt = b[:, 7]
entry = (rt & np.uint64(0xffffffff)).astype(np.int64)
exit_ = (rt >> np.uint64(32)).astype(np.int64)
e0 = entry.min()
print("realtime (us after the first wave's entry): entry pct", np.percentile((entry - e0) / 100.0, q))
print("realtime: exit  pct", np.percentile((exit_ - e0) / 100.0, q))
print("realtime: wave lifetime pct", np.percentile((exit_ - entry) / 100.0, q), "mean", ((exit_ - entry) / 100.0).mean())
print("realtime: start-up (entry -> first item's loads issued) = lifetime - dur: pct", np.percentile((exit_ - entry) / 100.0 - dur / F, q))
print("realtime: last exit - first entry = %.1f us" % ((exit_.max() - e0) / 100.0))
