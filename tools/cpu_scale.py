import sys, os, subprocess, tempfile
sys.path.insert(0, os.getcwd())
from espflix_amd import gen
b = gen.Batch(0, 256, 12, 12, 0)
td = tempfile.mkdtemp()
lst = os.path.join(td, "l.txt")
with open(lst, "w") as f:
    for i in range(256):
        p = os.path.join(td, f"{i}.ts"); b.ts(i).tofile(p); f.write(p + "\n")
for w, rep in ((1, 2), (8, 8), (32, 16), (64, 32), (128, 32), (256, 32)):
    r = subprocess.run(["oracle/_ref/efx_ref_decode", "bench", str(w), lst, str(rep)], stderr=subprocess.PIPE, text=True, timeout=120)
    print(r.stderr.strip().splitlines()[-1])
print(open("/proc/cpuinfo").read().count("processor"), os.cpu_count(), len(os.sched_getaffinity(0)))
try:
    print(open("/sys/fs/cgroup/cpu.max").read())
except Exception as e:
    print(e)
