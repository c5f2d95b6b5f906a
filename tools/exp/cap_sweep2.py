import os,sys,time,subprocess
sys.path.insert(0,'.')
if len(sys.argv)>1 and not os.environ.get("CHILD"):
    for spec in sys.argv[1:]:
        tag,cap,always=spec.split(':')
        env=dict(os.environ,CHILD="1",EFX_PARSE_WG_CAP=cap,EFX_LIB=os.path.join(os.getcwd(),"espflix_amd/libefx_%s.so"%tag))
        if always=='1': env['EFX_CAP_ALWAYS']='1'
        subprocess.run([sys.executable,sys.argv[0],spec],env=env)
    sys.exit(0)
import espflix_amd as efx
from espflix_amd import gen
b = gen.Batch(0, 1024, 12)
blobs = [b.es(k) for k in range(1024)]
dec = efx.Decoder(max_streams=1024, max_pictures=12, ring_depth=2)
dec.upload(blobs, efx.FORMAT_ES)
dec.decode(); dec.set_timing(True)
for _ in range(10): dec.decode()
t = dec.timing(); ser=(t.index_ms,t.parse_ms,t.recon_ms)
dec.set_timing(True); dec.sync(); t0=time.perf_counter()
for _ in range(200): dec.decode(sync=False)
dec.sync(); dt=(time.perf_counter()-t0)/200
t=dec.timing()
print(sys.argv[1],'serial index %.3f parse %.3f recon %.3f | pipelined parse %.3f recon %.3f step %.3f ms = %.2f M frames/s'%(*ser,t.parse_ms,t.recon_ms,dt*1e3,12288/dt/1e6))
