import os, sys
os.environ["KHR_DEBUG"]="16"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from khronos_amd import FusionContext, default_config
from khronos_amd.synth import SyntheticStream
W,H,vs=1280,720,0.02
cfg=default_config(voxel_size=vs, truncation_distance=3*vs, with_semantics=1, num_labels=20, max_blocks=40960, max_frame_pixels=W*H, max_mesh_vertices=1<<20,
                   md_min_cluster_size=500, md_min_separation_distance=2.0, md_max_range=5.0)
ctx=FusionContext(cfg); s=SyntheticStream(W,H,seed=1234); sen=ctx.make_sensor(W,H,s.fx,s.fy,s.cx,s.cy,0.1,5.0)
for i in range(26):
    fr=s.render(i)
    slot=ctx.upload_frame(sen, fr["stamp"], fr["pose"], fr["depth"], fr["rgb"], fr["label"])
    n=ctx.detect_motion(slot)
    ctx.integrate(slot, use_mask=True); ctx.update_tracking(fr["stamp"])
    st=ctx.stats()
    if st["n_seeds"]:
        buf=np.zeros(8,np.uint64); ctx.lib.khr_debug_read(ctx.h, buf.ctypes.data, 8)
        b=buf.astype(np.int64)
        print("frame",i,"S",int(b[6]),"clusters",n,"phases (ticks): init",b[1]-b[0],"edges",b[2]-b[1],"n_edges",b[3],"unite",b[4]-b[2],"flatten",b[5]-b[4],"total",b[5]-b[0])
