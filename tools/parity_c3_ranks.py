"""One-off check at the benchmark size: 2 and 4 ranks (shared-memory transport on one GPU) against one rank, k = 2, 1024^2,
one HDG-IMEX step.  usage: python tools/parity_c3_ranks.py [nranks ...]"""
import os, subprocess, sys, tempfile, uuid
import numpy as np
here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(here)
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
from test_gpu_multirank import _assemble  # noqa: E402

k, nx, nsteps = 2, 1024, 1
worker = os.path.join(root, "tests", "mp_strip_worker.py")
tmp = tempfile.mkdtemp()
ref = None
for P in [1] + [int(a) for a in sys.argv[1:] or ["2", "4"]]:
    token = "/hdg_par_" + uuid.uuid4().hex[:10]
    outs = [os.path.join(tmp, f"p{P}_r{r}.npz") for r in range(P)]
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(P), token, str(k), str(nx), str(nsteps), outs[r]],
                              env=dict(os.environ, HDG_FLOW_CHECK="1", HDG_DEBUG="1"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(P)]
    logs = [p.communicate(timeout=900)[0].decode(errors="replace") for p in procs]
    assert all(p.returncode == 0 for p in procs), logs[0][-2000:]
    parts = [np.load(o) for o in outs]
    Q, p, lam = _assemble(parts, k, nx) if P > 1 else (parts[0]["Q"], parts[0]["p"], parts[0]["lam"])
    fc = [ln for ln in logs[0].splitlines() if ln.startswith("[flow check]")]
    if ref is None:
        ref = (Q, p, lam)
        print(f"P=1: its {parts[0]['its']}")
    else:
        rel = lambda a, b: np.max(np.abs(a - b)) / np.max(np.abs(b))
        print(f"P={P}: rel. max deviation from one rank  Q {rel(Q, ref[0]):.2e}  p {rel(p, ref[1]):.2e}  lambda {rel(lam, ref[2]):.2e};  its {parts[0]['its']};  {fc[0] if fc else ''}", flush=True)
