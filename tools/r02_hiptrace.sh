#!/bin/bash
mkdir -p gpurun_out/r02h; O=$PWD/gpurun_out/r02h; R=$PWD
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $O/prof -o r02 -- python $R/bench.py --steps 20 --warmup 5 --cpu-baseline-frames 0 --latency-frames 0 > $O/prof.log 2>&1
tail -2 $O/prof.log | cut -c1-300
ls -la $O/prof/
# keep only the last 25% of the api trace (size)
python - <<PY
import csv,glob
f=glob.glob('$O/prof/*hip_api_trace.csv')[0]
rows=list(csv.reader(open(f)))
print(len(rows), rows[0])
keep=[rows[0]]+rows[int(len(rows)*0.7):]
csv.writer(open(f,'w')).writerows(keep)
PY
