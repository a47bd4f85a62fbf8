#!/bin/bash
# round 3: the -m gpu suite with the default kernels, then the fusion parity files again under the alternative k_fuse forms
mkdir -p gpurun_out/r03; O=$PWD/gpurun_out/r03
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -x > $O/pytest_default.log 2>&1; tail -3 $O/pytest_default.log | head -2
for spec in "KHR_FUSE_BAND=1" "KHR_FUSE_V=2" "KHR_FUSE_V=2 KHR_FUSE_BAND=1 KHR_FUSE2_CFG=3"; do
  n=$(echo $spec | tr ' =' '__')
  env $spec timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_golden.py -m gpu -q --timeout 300 -x > $O/pytest_$n.log 2>&1
  echo "$spec: $(tail -3 $O/pytest_$n.log | head -1)"
done
