#!/bin/bash
# round 5, run 20: the whole GPU suite, smoke() and the default bench line on the tree with the frame-parallel SBC decoder
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5fin4
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5fin4/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r5fin4/smoke.log
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r5fin4/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r5fin4/gpu_tests.log
timeout 900 python bench.py > gpurun_out/r5fin4/bench.json 2> gpurun_out/r5fin4/bench.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/r5fin4/bench.json; tail -3 gpurun_out/r5fin4/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5fin4/bench.json").read().strip().splitlines()[-1])
v = d.get("video_out", {})
print("sbc", json.dumps(v.get("sbc", {}))[:900])
print("demux", json.dumps(v.get("demux", {}))[:300])
PY
