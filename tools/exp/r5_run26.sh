#!/bin/bash
# round 5, run 26: the whole GPU suite under the guard-page allocator (run once with EFX_GUARD=1: buffers end on the last mapped
# byte, once with EFX_GUARD=2: they start on the first)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5guard3
EFX_GUARD=2 timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r5guard3/suite_guard2.log 2>&1; echo "guard 2 suite rc=$?"; tail -4 gpurun_out/r5guard3/suite_guard2.log | cut -c1-400
