#!/bin/bash
# round 5, GPU call 7: rocprofv3 material of the final structure (tools/collect_profiles.sh) + the rest of the video_out soak
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5g; mkdir -p $O
bash tools/collect_profiles.sh r5 > $O/collect.log 2>&1; echo "collect rc=$?" >> $O/rc.txt
EFX_GUARD=1 timeout 300 python bench.py --soak video_out 40 > $O/soak_g1_video_out.out 2> $O/soak_g1_video_out.err; echo "soak guard1 video_out x40 rc=$?" >> $O/rc.txt
tail -2 $O/soak_g1_video_out.err
cat $O/rc.txt; ls gpurun_out/prof_r5 | head -50
