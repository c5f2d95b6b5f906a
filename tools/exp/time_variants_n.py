"""development aid: like time_variants.py for a batch of N streams (env N, default 8192): python tools/exp/time_variants_n.py tag [tag ...]"""
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
N = int(os.environ.get("N", 8192))
flags = int(os.environ.get("FLAGS", 0))
b = gen.Batch(0, N, 12, 12, flags)
blobs = [b.es(k) for k in range(N)]
dec = efx.Decoder(max_streams=N, max_pictures=12, ring_depth=2)
dec.upload(blobs, efx.FORMAT_ES)
for _ in range(3):
    dec.decode()
dec.set_timing(True)
for _ in range(4):
    dec.decode()
t = dec.timing()
ser = (t.index_ms, t.parse_ms, t.recon_ms)
dec.sync()
steps = max(4, 40960 // N)
t0 = time.perf_counter()
for _ in range(steps):
    dec.decode(sync=False)
dec.sync()
dt = (time.perf_counter() - t0) / steps
print(tag, N, 'streams: serial index %.3f parse %.3f recon %.3f (groups %d halves %d) | pipelined step %.3f ms = %.2f M frames/s' % (*ser, t.groups, t.parse_halves, dt * 1e3, 12 * N / dt / 1e6))
