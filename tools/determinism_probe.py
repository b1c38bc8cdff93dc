"""Run-to-run determinism probe: the same one-step run several times per V-cycle variant, pairwise bitwise comparison."""
import os, subprocess, sys, tempfile
import numpy as np
here = os.path.dirname(os.path.abspath(__file__))
worker = os.path.join(here, "..", "tests", "mp_strip_worker.py")
k, nx = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1, 512)
tmp = tempfile.mkdtemp()
res = {}
for tag, extra, reps in (("fused", {}, 3), ("unfused", {"HDG_MG_NO_FUSE": "1"}, 2), ("plain", {"HDG_MG_NO_FUSE": "1", "HDG_MG_NO_TAIL": "1"}, 2)):
    for r in range(reps):
        out = os.path.join(tmp, f"{tag}{r}.npz")
        p = subprocess.run([sys.executable, worker, "0", "1", "unused", str(k), str(nx), "1", out], env=dict(os.environ, **extra),
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        assert p.returncode == 0, p.stdout.decode()[-2000:]
        res[(tag, r)] = np.load(out)
keys = list(res)
for a in range(len(keys)):
    for b in range(a + 1, len(keys)):
        d = {n: float(np.max(np.abs(res[keys[a]][n] - res[keys[b]][n]))) for n in ("Q", "p", "lam", "its")}
        print(keys[a], keys[b], d)
