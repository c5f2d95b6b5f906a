"""development aid (round 6): interleaved A/B timing of libefx_<tag>.so builds on one box, WITHOUT the parity gate (ablation
and sensitivity builds compute wrong pixels on purpose; correctness is pytest's job).

    python tools/exp/abtime.py [--rounds R] [--shape gop12|wide] [--pictures P] tag [tag ...]

Per tag and round one child process: 30 decodes one call at a time (median stage times from the library's HIP events)
and 200 back-to-back calls with the launch structure pinned as bench.py pins it (one reconstruction group per call).
Prints every round and, per tag, the medians over the rounds -- box-to-box and minute-to-minute drift is 1-2 %, the
rounds are interleaved so that it hits every tag alike."""
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def child(tag, shape, pictures):
    import numpy as np
    import espflix_amd as efx
    from espflix_amd import gen
    cache = f"/dev/shm/abtime_{shape}_{pictures}.npz"
    if os.path.exists(cache):
        z = np.load(cache)
        blobs = [z[f"s{k}"] for k in range(1024)]
    else:
        b = gen.Batch(0, 1024, pictures, 12, (4 | 32) if shape == "wide" else 0)
        blobs = [b.es(k) for k in range(1024)]
        np.savez(cache, **{f"s{k}": v for k, v in enumerate(blobs)})
    dec = efx.Decoder(max_streams=1024, max_pictures=pictures, ring_depth=2)
    dec.upload(blobs, efx.FORMAT_ES)
    dec.decode()
    ser = []
    for _ in range(30):
        dec.set_timing(True)
        dec.decode()
        t = dec.timing()
        ser.append((t.index_ms, t.parse_ms, t.recon_ms))
    med = [statistics.median(x[i] for x in ser) for i in range(3)]
    dec.set_timing(False)
    dec.set_option(efx.OPT_GROUPS, 1)
    steps = []
    for _ in range(3):
        dec.sync()
        t0 = time.perf_counter()
        for _ in range(100):
            dec.decode(sync=False)
        dec.sync()
        steps.append((time.perf_counter() - t0) / 100)
    dt = min(steps)
    print("RESULT %s %.4f %.4f %.4f %.4f" % (tag, med[0], med[1], med[2], dt * 1e3), flush=True)


def main():
    args = sys.argv[1:]
    rounds, shape, pictures = 3, "gop12", 12
    while args and args[0].startswith("--"):
        if args[0] == "--rounds":
            rounds = int(args[1])
        elif args[0] == "--shape":
            shape = args[1]
        elif args[0] == "--pictures":
            pictures = int(args[1])
        args = args[2:]
    if os.environ.get("ABTIME_CHILD"):
        child(args[0], shape, pictures)
        return
    res = {t: [] for t in args}
    for r in range(rounds):
        for tag in args:
            lib = os.path.join(ROOT, "espflix_amd", "libefx.so" if tag == "lib" else f"libefx_{tag}.so")
            env = dict(os.environ, EFX_LIB=lib, ABTIME_CHILD="1")
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--shape", shape, "--pictures", str(pictures), tag], env=env,
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
            if not line:
                print(tag, "FAILED", p.stderr[-400:])
                continue
            v = [float(x) for x in line[0].split()[2:]]
            res[tag].append(v)
            print("round %d %-16s serial index %.3f parse %.3f recon %.3f | back to back %.3f ms = %.2f M frames/s" % (r, tag, *v, 1024 * pictures / v[3] / 1e3))
    print("---- medians over %d rounds (%s, %d pictures)" % (rounds, shape, pictures))
    base = None
    for tag in args:
        if not res[tag]:
            continue
        m = [statistics.median(x[i] for x in res[tag]) for i in range(4)]
        base = base or m
        print("%-16s serial index %.3f parse %.3f recon %.3f (%+.1f %%) | back to back %.3f ms (%+.1f %%) = %.2f M frames/s" % (
            tag, m[0], m[1], m[2], 100 * (m[2] / base[2] - 1), m[3], 100 * (m[3] / base[3] - 1), 1024 * pictures / m[3] / 1e3))


if __name__ == "__main__":
    main()
