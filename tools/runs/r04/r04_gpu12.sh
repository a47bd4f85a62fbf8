#!/bin/bash
# round 4, session 2: access-pattern micro-benchmark + PMC (TCP group) of k_fuse under the ablation switches
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_12
mkdir -p $O
timeout 120 tools/ubench/band_patterns > $O/band_patterns.txt 2>&1; echo "ubench rc $?" >> $O/rc.txt
bash tools/pmc_fuse.sh "c" base KHR_FUSE_DBG=8 KHR_FUSE_DBG=1 KHR_FUSE_DBG=9 KHR_FUSE_DBG=11 > $O/pmc.log 2>&1; echo "pmc rc $?" >> $O/rc.txt
for k in 1 2 3 4 5; do cp gpurun_out/pmc_fuse_$k/k_fuse_pmc.json $O/k_fuse_pmc_$k.json 2>/dev/null; done
cat $O/rc.txt $O/band_patterns.txt; cat $O/k_fuse_pmc_*.json
