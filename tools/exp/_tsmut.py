import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np
import espflix_amd as efx, oracle, common
from espflix_amd import gen
efx.load_library()
rng=np.random.default_rng(21)
streams=[];desc=[]
for k in range(96):
    b=gen.Batch(300+k,1,5,12,[0,4,128,256][k%4])
    if k%3==0: ts=bytes(b.ts(0))
    else: ts=common.hostile_ts(b.es(0).tobytes(), 50+k, noise=(k%3==1))
    pk=[ts[i:i+188] for i in range(0,len(ts)-187,188)]
    op=int(rng.integers(0,7)); i=int(rng.integers(1,len(pk)-1))
    if op==0: d='drop packet'; del pk[i]
    elif op==1: d='duplicate packet'; pk.insert(i,pk[i])
    elif op==2: j=int(rng.integers(1,len(pk)-1)); d='swap packets'; pk[i],pk[j]=pk[j],pk[i]
    elif op==3: d='bad sync byte'; pk[i]=b'\x48'+pk[i][1:]
    elif op==4: d='truncate stream mid packet'; pk=pk[:i]+[pk[i][:100]]
    elif op==5: d='clear payload flag'; p=bytearray(pk[i]); p[3]&=~0x10; pk[i]=bytes(p)
    else: d='set PUSI on a continuation packet'; p=bytearray(pk[i]); p[1]|=0x40; pk[i]=bytes(p)
    streams.append(np.frombuffer(b''.join(pk),dtype=np.uint8)); desc.append(d+(' gen-ts' if k%3==0 else ' hostile'))
dec=efx.Decoder(len(streams),8,2,max_stream_bytes=sum(len(s) for s in streams)+8192)
dec.upload(streams,efx.FORMAT_TS); dec.decode(); h=dec.frame_hashes()
bad=0
for i,s in enumerate(streams):
    n=dec.picture_count(i); st=dec.stream_status(i)
    on,oh,opts,_=oracle.decode(s,1,True)
    got=[int(h[i,dec.picture_slot(p)]) for p in range(max(0,n-2),n)]
    want=[int(x) for x in oh][max(0,on-2):]
    gpts=[dec.picture_pts(i,p) for p in range(n)]
    same=(n==on and got==want and gpts==[int(x) for x in opts][:n]) or (n==0 and on<=1)
    if not same:
        print(('UNFLAGGED ' if st==0 else 'flagged   ')+desc[i],'status',st,'pics',n,on,[a==b for a,b in zip(got,want)], 'pts ok' if gpts==[int(x) for x in opts][:n] else 'pts differ')
        bad+= st==0
print('unflagged deviations:',bad,'of',len(streams))
