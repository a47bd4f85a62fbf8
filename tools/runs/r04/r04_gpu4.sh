#!/bin/bash
# round 4, fourth GPU call: what bounds k_band?  ablations of the band update inside k_band only (exact k_fuse ignores the switches),
# team counts, the fused baseline; host consumer with reused mesh buffers; rocprofv3 kernel stats of the split form
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_4
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_snapshot.py -x -q -m gpu > $O/parity.txt 2>&1; echo "parity rc $?" >> $O/rc.txt
B="--steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B > $O/b_$name.json 2> $O/b_$name.err; }
run fused KHR_FUSE_SPLIT=0
run split_s4 KHR_BAND_TEAMS=4
run split_s8 KHR_BAND_TEAMS=8
run split_s2 KHR_BAND_TEAMS=2
run split_s6 KHR_BAND_TEAMS=6
run abl_noimg KHR_FUSE_DBG=1024
run abl_noliklo KHR_FUSE_DBG=2048
run abl_nolikst KHR_FUSE_DBG=4096
run abl_nolik KHR_FUSE_DBG=6144
run abl_nostores KHR_FUSE_DBG=12288
run abl_noparta_loads KHR_FUSE_DBG=17408
run abl_noloads KHR_FUSE_DBG=19456
run abl_nothing KHR_FUSE_DBG=31744
timeout 300 python bench.py $B --output-copy host > $O/b_host.json 2> $O/b_host.err
timeout 300 python bench.py $B --output-copy host --host-fields all > $O/b_host_all.json 2> $O/b_host_all.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_split -o split -- python $GRAFT_REPO_ROOT/bench.py $B > /dev/null 2> $GRAFT_REPO_ROOT/$O/prof_split.err
cd $GRAFT_REPO_ROOT
find $O/prof_split -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/split_kernel_stats.csv
find $O/prof_split -type f -size +20M -delete
cat $O/rc.txt; tail -n 2 $O/parity.txt
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_4/b_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        r = j["roofline"]
        print("%-22s fps %5.0f ms/step %.4f  k_fuse %6.1f us  k_band %s us  total %6.1f  frac %.3f  host B/out %s" % (
            f.split("/")[-1][2:-5], j["value"], j["ms_per_step"], r["k_fuse_avg_us"], ("%6.1f" % r["k_band_avg_us"]) if r["k_band_avg_us"] else "   -  ",
            r["avg_launch_us"], r["frac"], j["output_copy"].get("host_bytes_per_output")))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-600:])
PY
head -14 $O/split_kernel_stats.csv | cut -c1-160
