#!/bin/bash
# the driver's command N times (default 30) on one box; per run the headline and the library's seed-wait watchdog
# (khr_stats: waits, late ones, histogram, queue state of the last late one) -> gpurun_out/seed_latency.txt
N=${1:-30}; R=$PWD; O=$R/gpurun_out; mkdir -p $O
: > $O/seed_latency_runs.jsonl
for i in $(seq 1 $N); do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0 2> $O/sl.err > /dev/null
  grep '^{' $O/sl.err | tail -n 1 >> $O/seed_latency_runs.jsonl
done
python - "$O" "$N" <<'PY'
import json, sys
O, N = sys.argv[1], int(sys.argv[2])
runs = [json.loads(l) for l in open(O + "/seed_latency_runs.jsonl") if l.strip()]
hist = [0] * 8
with open(O + "/seed_latency.txt", "w") as f:
    f.write("# python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0, %d runs on one box (tools/seed_latency.sh)\n" % len(runs))
    f.write("# per run: frames/s, ms/step, drain ms | seed-count waits of the whole run (95 frames): count, late (> 0.5 ms), longest us, [state bits of the last late one]\n")
    late_runs = 0
    for i, j in enumerate(runs):
        sw = j["seed_wait"]
        for k in range(8):
            hist[k] += sw["hist_us_50_100_200_500_1k_2k_5k_more"][k]
        late_runs += 1 if sw["late_over_500us"] else 0
        f.write("run %2d  %7.1f frames/s  %.4f ms/step  drain %.3f ms | waits %3d  late %d (%d inside the timed steps)  max %5d us%s\n" % (
            i + 1, j["value"], j["ms_per_step"], j["timed_region"]["drain_and_join_ms"], sw["waits"], sw["late_over_500us"],
            sw["in_timed_steps"]["late_over_500us"], sw["max_us"],
            ("  last late: wait no. %d, %d us, state 0x%x" % (sw["last_late"]["wait_no"], sw["last_late"]["us"], sw["last_late"]["state_bits"])) if sw["late_over_500us"] else ""))
    v = sorted(j["value"] for j in runs)
    f.write("# frames/s: min %.0f  median %.0f  max %.0f\n" % (v[0], v[len(v) // 2], v[-1]))
    f.write("# all waits by duration (< 50, < 100, < 200, < 500 us, < 1, < 2, < 5 ms, longer): %s\n" % hist)
    f.write("# runs with at least one late wait: %d of %d\n" % (late_runs, len(runs)))
print(open(O + "/seed_latency.txt").read())
PY
