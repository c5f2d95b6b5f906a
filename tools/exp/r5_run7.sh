#!/bin/bash
# round 5, GPU call 7b: rocprofv3 material of the final structure (tools/collect_profiles.sh)
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5g; mkdir -p $O
bash tools/collect_profiles.sh r5 > $O/collect.log 2>&1; echo "collect rc=$?" >> $O/rc.txt
cat $O/rc.txt; du -sh gpurun_out
