"""Communication census of one strip-partitioned step: 2 ranks on the one GPU (shm transport), HDG_DEBUG
prints the number of halo exchanges / all-reduces / all-gathers issued by rank 0."""
import os, subprocess, sys, uuid, tempfile
here = os.path.dirname(os.path.abspath(__file__))
worker = os.path.join(here, "..", "tests", "mp_strip_worker.py")
k, nx, nsteps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
token = "/hdg_census_" + uuid.uuid4().hex[:10]
tmp = tempfile.mkdtemp()
procs = [subprocess.Popen([sys.executable, worker, str(r), "2", token, str(k), str(nx), str(nsteps), os.path.join(tmp, f"r{r}.npz")],
                          env=dict(os.environ, HDG_DEBUG="1"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
for r, p in enumerate(procs):
    o, _ = p.communicate(timeout=600)
    txt = o.decode(errors="replace")
    for line in txt.splitlines():
        if line.startswith("[comm]") or line.startswith("[flow check]"):
            print(f"k={k} nx={nx} steps={nsteps}:", line)
    if p.returncode != 0:
        print("rank", r, "failed", txt[-1500:])
