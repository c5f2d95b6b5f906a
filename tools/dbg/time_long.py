import subprocess, time, os
env = dict(os.environ, EFX_DROPIN_TIMEOUT="15")
for exe in ("tests/_build/espflix_dropin_long_old", "tests/_build/espflix_dropin_long") * 4:
    t = time.time(); r = subprocess.run([exe], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=env); print(exe, "%.2f s rc %d" % (time.time() - t, r.returncode))
