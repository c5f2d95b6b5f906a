"""development aid: N decode calls of the bench batch (1024 streams x GOP 12), one at a time, NO parity gate -- the command the
counter passes of tools/exp/pmc_variants.sh profile (ablation builds compute wrong pixels on purpose).  EFX_LIB selects the build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import espflix_amd as efx
from espflix_amd import gen
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
b = gen.Batch(0, 1024, 12)
dec = efx.Decoder(max_streams=1024, max_pictures=12, ring_depth=2)
dec.upload([b.es(k) for k in range(1024)], efx.FORMAT_ES)
for _ in range(n):
    dec.decode()
dec.close()
