#!/bin/bash
# round 4, session 2: instruction-cache and wait counters of the new k_fuse
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_16
mkdir -p $O
bash tools/pmc_fuse.sh "kac" base > $O/pmc.log 2>&1; echo "pmc rc $?" >> $O/rc.txt
cp gpurun_out/pmc_fuse_1/k_fuse_pmc.json $O/k_fuse_pmc.json
cat $O/rc.txt $O/k_fuse_pmc.json; tail -5 $O/pmc.log
