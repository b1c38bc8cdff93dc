# per-degree comparison of the two tentative-velocity Krylov methods (nx = 512)
for k in 1 2 3 4; do for sol in 1 0; do
  timeout -k 10 300 python bench.py --degree $k --nx 512 --steps 3 --no-cpu-baseline --tent-solver $sol > gpurun_out/d.log 2>&1
  python - "$k" "$sol" <<PY
import json,sys
d=json.loads(open("gpurun_out/d.log").read().strip().splitlines()[-1])
it=d["config"]["krylov_iterations_avg"]
print("k=%s solver=%s: %.1f ms/step %.1f MDOF/s tentative its %.1f CG its %.1f"%(sys.argv[1],sys.argv[2],d["ms_per_step"],d["value"],it["tentative"],it["pressure"]))
PY
done; done
