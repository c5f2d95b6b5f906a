"""development aid: tests/_build/parse_harness over 13 generator flavours x 16 streams with 0-3 bit flips each (CPU only).
   python tools/dbg/parse_soak.py   -> totals; a mismatch is printed and the stream kept in /tmp/soak_fail_*.es"""
import sys, subprocess, numpy as np, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from espflix_amd import gen
H='/root/repo/tests/_build/parse_harness'
tot=dict(slices=0,unseen=0,phantom=0,fail=0,runs=0)
rng=np.random.default_rng(3)
import re
for fl in [0,1,2,4,8,16,36,64,68,128,132,256,260]:
    b=gen.Batch(1000, 16, 6, 12, fl)
    for k in range(16):
        es=b.es(k).copy()
        flips=int(rng.integers(0,4))
        for _ in range(flips):
            at=int(rng.integers(100, es.size-100))
            if es[at-3:at+4].min()==0: continue
            es[at]^=1<<int(rng.integers(0,8))
        es.tofile('/tmp/soak.es')
        p=subprocess.run([H,'/tmp/soak.es'],capture_output=True,text=True)
        tot['runs']+=1
        if p.returncode:
            tot['fail']+=1; print('FAIL fl',fl,'k',k,'flips',flips,p.stderr.strip()[:200]); es.tofile(f'/tmp/soak_fail_{fl}_{k}.es')
        else:
            m=re.match(r"OK slices=(\d+) .* unseen=(\d+) phantom=(\d+)",p.stdout)
            tot['slices']+=int(m.group(1)); tot['unseen']+=int(m.group(2)); tot['phantom']+=int(m.group(3))
print(tot)
