L=$GRAFT_REPO_ROOT/espflix_amd
EFX_LIB=$L/libefx_z.so timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for t in a z a z; do EFX_LIB=$L/libefx_$t.so python bench.py --steps 20 --no-cpu-baseline --no-fixed-batch --no-video-out --sustained-steps 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); o=d['other_workloads']; print('[$t]', round(d['value']), {k:round(v['frames_per_s']) for k,v in o.items()})"; done
