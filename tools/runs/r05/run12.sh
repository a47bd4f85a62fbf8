#!/bin/bash
# round 5, run 12: what the emit launch spends on the copy of the kept meshes (A/B switch KHR_MC_SPLIT_MOVE), kernel trace
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${RUN12_OUT:-r05_12}; mkdir -p $O
cd $R
for v in 0 1; do
  if [ $v = 1 ]; then export KHR_MC_SPLIT_MOVE=1; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -o run -- python bench.py --steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0 > $O/bench_$v.json 2> $O/bench_$v.err
  f=$(find $O/prof_$v -name "*kernel_stats.csv" | head -1)
  echo "== split_move=$v"; python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n=r['Name']
    if any(k in n for k in ('k_marching','k_mesh_move','k_snapshot_pack','k_fuse<')):
        print("%-60s calls %5s avg %8.1f us min %8.1f max %8.1f" % (n[:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
  find $O/prof_$v -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$v.csv \;
  rm -rf $O/prof_$v
done
