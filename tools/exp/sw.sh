python tools/exp/env_sweep.py "T=new"
FLAGS=36 python tools/exp/env_sweep.py "T=new36"
timeout 700 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
