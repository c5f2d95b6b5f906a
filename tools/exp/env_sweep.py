"""development aid: pipelined / serial stage times under environment variants, same box:
   python tools/exp/env_sweep.py "" "EFX_RECON2=1" "EFX_PARSE_WG_CAP=128" ...   (each spec: space-separated KEY=VALUE pairs)"""
import os,sys,time,subprocess
sys.path.insert(0,'.')
if not os.environ.get("CHILD"):
    for spec in sys.argv[1:]:
        env=dict(os.environ,CHILD="1",SPEC=spec)
        for kv in spec.split():
            k,v=kv.split('='); env[k]=v
        subprocess.run([sys.executable,sys.argv[0]],env=env)
    sys.exit(0)
import espflix_amd as efx
from espflix_amd import gen
flags=int(os.environ.get("FLAGS","0"))
NS=int(os.environ.get("STREAMS","1024"))
b = gen.Batch(0, NS, 12, 12, flags)
blobs = [b.es(k) for k in range(NS)]
dec = efx.Decoder(max_streams=NS, max_pictures=12, ring_depth=2)
dec.upload(blobs, efx.FORMAT_ES)
dec.decode(); dec.decode(); dec.set_timing(True)
for _ in range(10): dec.decode()
t = dec.timing(); ser=(t.index_ms,t.parse_ms,t.recon_ms)
dec.set_timing(True); dec.sync(); t0=time.perf_counter()
NIT=max(20,200*1024//NS)
for _ in range(NIT): dec.decode(sync=False)
dec.sync(); dt=(time.perf_counter()-t0)/NIT
t=dec.timing()
print('[%s] serial index %.3f parse %.3f recon %.3f | pipelined parse %.3f recon %.3f step %.3f ms = %.2f M frames/s'%(os.environ["SPEC"],*ser,t.parse_ms,t.recon_ms,dt*1e3,12*NS/dt/1e6))
