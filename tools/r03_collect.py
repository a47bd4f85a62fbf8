#!/usr/bin/env python3
"""copy the summaries of tools/r03_artifacts.sh from gpurun_out/r03art/ into profiles/ (tracked) under round-3 names"""
import json
import os
import shutil

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
A = os.path.join(R, "gpurun_out", "r03art")
P = os.path.join(R, "profiles")
copies = {
    "gpu_tests.txt": "r03_gpu_tests.txt", "bench_c3_1.json": "r03_bench_c3.json", "bench_c3_2.json": "r03_bench_c3_run2.json",
    "bench_c3_3.json": "r03_bench_c3_run3.json", "bench_c3_100steps.json": "r03_bench_c3_100steps.json",
    "bench_c3_noobj.json": "r03_bench_c3_noobj.json", "kernel_stats.csv": "r03_kernel_stats.csv",
    "kernel_stats_timed.csv": "r03_kernel_stats_timed.csv", "k_fuse_durations.txt": "r03_k_fuse_durations.txt",
    "kernel_trace_frames.txt": "r03_kernel_trace_frames.txt", "emu_c5.txt": "r03_emu_c5.txt", "emu_c3.txt": "r03_emu_c3.txt",
    "emu_c4.txt": "r03_emu_c4.txt", "c5_emu8.json": "r03_bench_c5_emu8.json", "c5_n1.json": "r03_bench_c5_n1.json",
    "c3_emu8.json": "r03_bench_c3_emu8.json", "c4_emu4.json": "r03_bench_c4_emu4.json", "tr_c5_8_per_tick.csv": "r03_emu8_c5_kernels_per_tick.csv",
    "tcp_reads.txt": "r03_ubench_tcp_reads.txt", "probe_fuse.txt": "r03_probe_fuse_timeline.txt",
    # tools/r03_artifacts_tick.sh (after the tick's one-launch update / allocation)
    "gpu_tests_tick.txt": "r03_gpu_tests_tick.txt", "tr_c3_8_per_tick.csv": "r03_emu8_c3_kernels_per_tick.csv",
    "tick_union_ab.txt": "r03_tick_union_ab.txt", "queue_atomics.txt": "r03_ubench_queue_atomics.txt",
}
for src, dst in copies.items():
    s = os.path.join(A, src)
    if os.path.exists(s):
        shutil.copyfile(s, os.path.join(P, dst))
    else:
        print("missing", src)
# PMC summary in the form bench.py reads (roofline.traffic)
pmc = json.load(open(os.path.join(A, "k_fuse_pmc.json")))
bench = json.load(open(os.path.join(A, "bench_c3_1.json")))
fetch_kib, write_kib = pmc["FETCH_SIZE"], pmc["WRITE_SIZE"]
out = {
    "kernel": "k_fuse", "workload": [1280, 720, 0.02],
    "collected": "round 3 (k_fuse: per-wave software pipeline, lane <-> record band phase, speculative gated launch), tools/r03_pmc.sh: rocprofv3 "
                 "--kernel-trace --pmc, one counter group per run, bench.py --steps 10 --warmup 5 --preroll 20 --no-objects; per-launch "
                 "averages over the 10 TIMED launches (frames 25..34 of the stream)",
    "k_fuse_bytes_per_launch": (fetch_kib + write_kib) * 1024.0,
    "FETCH_SIZE_KiB_per_launch": fetch_kib, "WRITE_SIZE_KiB_per_launch": write_kib,
    "k_fuse_us_per_launch_under_the_profiler": {k[len("avg_us_pass_"):]: v for k, v in pmc.items() if k.startswith("avg_us_pass_")},
    "note": "FETCH_SIZE / WRITE_SIZE as reported (KiB x 1024).  tools/ubench/tcp_reads.hip (profiles/r03_ubench_tcp_reads.txt): random 64-byte "
            "reads from HBM saturate at 51 G requests/s = 3.3 TB/s of sectors, i.e. 6.5 TB/s if every request moves a 128-byte line -- the "
            "guide's factor 2 on FETCH_SIZE for wide streams may well apply to k_fuse's scattered band reads too; the figure is NOT doubled "
            "here (with every read doubled the total would be %.0f MB)." % ((2 * fetch_kib + write_kib) * 1024.0 / 1e6),
    "other_counters_per_launch": {k: v for k, v in pmc.items() if k not in ("spec", "FETCH_SIZE", "WRITE_SIZE") and not k.startswith("avg_us")},
}
json.dump(out, open(os.path.join(P, "r03_pmc_traffic.json"), "w"), indent=1)
json.dump(out, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)  # the file bench.py labels `roofline.traffic` from
print("k_fuse HBM bytes / launch (PMC): %.1f MB; driver line %.0f fps" % (out["k_fuse_bytes_per_launch"] / 1e6, bench["value"]))
