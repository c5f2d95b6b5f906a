"""development aid: serial and pipelined stage times of libefx_<tag>.so builds WITHOUT the parity gate (for ablation
builds whose output is wrong on purpose): python tools/exp/time_variants.py tag [tag ...]"""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 2 or (len(sys.argv) == 2 and not os.environ.get("EFX_LIB_CHILD")):
    for tag in sys.argv[1:]:
        env = dict(os.environ, EFX_LIB=os.path.join(os.path.dirname(sys.argv[0]), "..", "..", "espflix_amd", f"libefx_{tag}.so"), EFX_LIB_CHILD="1")
        subprocess.run([sys.executable, sys.argv[0], tag], env=env)
    sys.exit(0)
import espflix_amd as efx
from espflix_amd import gen
tag = sys.argv[1]
b = gen.Batch(0, 1024, 12)
blobs = [b.es(k) for k in range(1024)]
dec = efx.Decoder(max_streams=1024, max_pictures=12, ring_depth=2)
dec.upload(blobs, efx.FORMAT_ES)
dec.decode()
dec.set_timing(True)
for _ in range(10):
    dec.decode()
t = dec.timing()
ser = (t.index_ms, t.parse_ms, t.recon_ms)
dec.set_timing(True)
dec.sync()
t0 = time.perf_counter()
for _ in range(100):
    dec.decode(sync=False)
dec.sync()
dt = (time.perf_counter() - t0) / 100
print(tag, 'serial index %.3f parse %.3f recon %.3f | pipelined step %.3f ms = %.2f M frames/s' % (*ser, dt * 1e3, 12288 / dt / 1e6))
