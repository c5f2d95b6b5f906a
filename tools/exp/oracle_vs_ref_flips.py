"""development aid (build container only: needs oracle/_ref): single-bit flips in the video payload of TS-wrapped synthetic streams,
the oracle against the UNMODIFIED reference decoder (8 s limit per play)"""
import sys, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import oracle
from espflix_amd import gen
import subprocess, tempfile, os
def ref(ts):
    with tempfile.TemporaryDirectory() as td:
        src=os.path.join(td,'in.ts'); ts.tofile(src)
        p=subprocess.run([os.path.join(oracle.REF_DIR,'efx_ref_decode'),'decode',src,'-','flush'],stderr=subprocess.PIPE,stdout=subprocess.DEVNULL,text=True,timeout=8)
        rows=[l.split() for l in p.stderr.splitlines() if l.startswith('F ')]
        return np.array([int(r[3],16) for r in rows],dtype=np.uint64), np.array([int(r[2]) for r in rows],dtype=np.int64)
b=gen.Batch(0,8,6)
rng=np.random.default_rng(7)
agree=differ=0; bad=[]
N=int(sys.argv[1]) if len(sys.argv)>1 else 100
for t in range(N):
    k=t%8
    ts=np.frombuffer(b.ts(k),dtype=np.uint8).copy()
    while True:
        pos=int(rng.integers(188,ts.size-188))
        if pos%188>=40: break
    bit=int(rng.integers(0,8))
    ts[pos]^=1<<bit
    n,oh,opts,_=oracle.decode(ts,1,True)
    try:
        rh,rpts=ref(ts)
    except Exception as e:
        bad.append((k,pos,bit,'ref failed',str(e)[:60])); differ+=1; continue
    if n==len(rh) and np.array_equal(oh,rh) and np.array_equal(opts,rpts): agree+=1
    else:
        differ+=1; bad.append((k,pos,bit,n,len(rh), int((oh[:min(n,len(rh))]!=rh[:min(n,len(rh))]).sum())))
print('flips',N,'oracle == reference',agree,'differ',differ)
print('reference hung or crashed', sum(1 for x in bad if x[3]=='ref failed'), 'finished but differs', sum(1 for x in bad if x[3]!='ref failed'))
