#!/bin/bash
# round 5, run 26: the whole GPU suite under the guard-page allocator (buffers end on the last mapped byte)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5guard3
EFX_GUARD=1 timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r5guard3/suite_guard1.log 2>&1; echo "guard 1 suite rc=$?"; tail -4 gpurun_out/r5guard3/suite_guard1.log | cut -c1-400
