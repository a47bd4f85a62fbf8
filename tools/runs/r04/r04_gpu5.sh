#!/bin/bash
# round 4, fifth GPU call: common-clock timeline of k_fuse (split and fused): where do the ~25 us between the waves' busy time and the launch go?
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_5
mkdir -p $O
KHR_FUSE_SPLIT=1 timeout 300 python tools/probe_fuse.py 30 > $O/probe_split.txt 2>&1; echo "probe_split rc $?" >> $O/rc.txt
KHR_FUSE_SPLIT=0 timeout 300 python tools/probe_fuse.py 30 > $O/probe_fused.txt 2>&1; echo "probe_fused rc $?" >> $O/rc.txt
cat $O/rc.txt; grep -n "last launch\|^dur\|^band   pct\|realtime\|non-band\|per-WG max\|tsdf blocks" $O/probe_split.txt $O/probe_fused.txt
