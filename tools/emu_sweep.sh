# per-rank tick cost of an N-rank sharded run, emulated on one GPU (bench.py --emulate-world)
mkdir -p gpurun_out
for n in 1 2 4 8; do
  for obj in "--no-objects" ""; do
    timeout 300 python bench.py --emulate-world $n --steps 30 --warmup 10 --cpu-baseline-frames 0 $obj 2>gpurun_out/emu.err | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=$n', '$obj' or 'full', 'fps', round(j['value']), 'ms/tick', round(j['ms_per_step'],3), 'tsdf_us', round(j['roofline']['avg_launch_us'],1))" || tail -5 gpurun_out/emu.err
  done
done
