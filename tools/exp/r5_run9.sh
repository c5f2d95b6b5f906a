#!/bin/bash
# round 5, GPU call 9: the start-aligned guard build (EFX_GUARD=2: under-reads / under-writes) brought to 100 repetitions per leg
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5i; mkdir -p $O
for cfg in "wide1500k:40" "vmedia_x1024:80" "video_out:94" "fixed_batch_8192:96"; do
  IFS=: read l n <<< "$cfg"
  EFX_GUARD=2 timeout 500 python bench.py --soak $l $n > $O/soak_g2_$l.out 2> $O/soak_g2_$l.err; echo "soak guard2 $l x$n rc=$?" >> $O/rc.txt
  grep "soak " $O/soak_g2_$l.err | tail -1
done
cat $O/rc.txt
