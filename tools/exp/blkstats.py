"""development aid (round 6): coded-block / coefficient statistics of bench streams from the oracle parse trace, per wave of 64 blocks in k_recon order (profiles/r6_recon_vmem.md)"""
import sys, ctypes as C, numpy as np, collections
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import oracle
from espflix_amd import gen
ZZ=[0,1,8,16,9,2,3,10,17,24,32,25,18,11,4,5,12,19,26,33,40,48,41,34,27,20,13,6,7,14,21,28,35,42,49,56,57,50,43,36,29,22,15,23,30,37,44,51,58,59,52,45,38,31,39,46,53,60,61,54,47,55,62,63]
def stats(sid, flags=0):
    es = gen.Batch(sid,1,12,12,flags).es(0)
    pics=[]  # per picture: dict mb -> [ (blk, [positions]) ]
    cur={'pic':-1,'mb':None}
    blocks=collections.defaultdict(lambda: collections.defaultdict(list))  # pic -> (mb,blk) -> positions
    mbinfo=collections.defaultdict(dict)
    def cb(_u,kind,a,b,c,e):
        if kind==0: cur['pic']=a
        elif kind==1: cur['mb']=a; mbinfo[cur['pic']][a]=b
        elif kind==2: blocks[cur['pic']][(cur['mb'],a)].append(b)
    FN=C.CFUNCTYPE(None,C.c_void_p,C.c_int,C.c_int,C.c_int,C.c_int,C.c_int)
    fn=FN(cb); L=oracle.lib(); L.efxo_set_trace.argtypes=[FN,C.c_void_p]; L.efxo_set_trace(fn,None)
    d=np.ascontiguousarray(es,dtype=np.uint8)
    L.efxo_decode(d.ctypes.data,d.size,0,1,None,None,None,0); L.efxo_set_trace(FN(),None)
    return blocks, mbinfo
def block_order():
    order=[]
    for b in range(1584):
        mbrow=b//132; rr=b%132
        if rr<88:
            h=rr>=44; r2=rr-44*h; blk=2*h+(r2&1); mbx=r2>>1
        else:
            h=rr>=110; blk=4+h; mbx=rr-88-22*h
        order.append((mbrow*22+mbx,blk))
    return order
order=block_order()
for sid in (0,1,5,1000):
    blocks,mbinfo=stats(sid)
    for pic in (0,1,6,11):
        bl=blocks[pic]
        cnts=[len(bl.get(k,[])) for k in order]
        cn=np.array(cnts)
        nz=(cn>0).mean()
        # per wave stats
        waves=[order[i:i+64] for i in range(0,1584,64)]
        tot=[sum(len(bl.get(k,[])) for k in w) for w in waves]
        def conf(w,lim):
            for k in w:
                for p in bl.get(k,[]):
                    z=ZZ[p]
                    if (z>>3)>=lim or (z&7)>=lim: return False
            return True
        c4=np.mean([conf(w,4) for w in waves]); c6=np.mean([conf(w,6) for w in waves])
        colocc=np.mean([len({ZZ[p]&7 for k in w for p in bl.get(k,[])}) for w in waves])
        maxc=[max(len(bl.get(k,[])) for k in w) for w in waves]
        print(f"stream {sid} pic {pic}: coded blocks {nz:.2f} mean cnt {cn.mean():.2f} max {cn.max()} | per wave total mean {np.mean(tot):.0f} max {max(tot)} rounds(192) {np.mean([ (t+191)//192 for t in tot]):.2f} maxcnt mean {np.mean(maxc):.1f} | waves all<4x4 {c4:.2f} all<6x6 {c6:.2f} cols occupied {colocc:.1f}")
