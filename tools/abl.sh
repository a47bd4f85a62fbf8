mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for ch in 2 4; do for g in 2048 4096; do
KHR_TSDF_CHUNKS=$ch KHR_TSDF_GRID=$g python bench.py --cpu-baseline-frames 0 --no-motion --output-every 0 > gpurun_out/abl_x.json 2>gpurun_out/abl.err
python -c "
import json; d=json.load(open('gpurun_out/abl_x.json')); print('chunks=$ch grid=$g fps', round(d['value']), {k:(round(1e3*v['ms_total']/max(1,v['launches']),1)) for k,v in d['kernel_ms'].items() if k in ('tsdf','band')})"
done; done
