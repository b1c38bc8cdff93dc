# scan the Chebyshev ellipse fraction / upper interval factor / hand-over threshold of the tentative-velocity solver
# usage: ELLS="0.2 0.3" FHIS="1.3" HANDS="0 0.6" ARGS="--degree 2 --nx 512" bash tools/ellscan.sh
for hand in ${HANDS:-0}; do for ell in ${ELLS:-0.3 0.5}; do for fhi in ${FHIS:-1.3}; do
  HDG_CHEB_HANDOVER=$hand HDG_CHEB_ELL=$ell HDG_CHEB_FHI=$fhi python bench.py --steps ${STEPS:-4} --warmup 2 --no-cpu-baseline $ARGS > gpurun_out/ell.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/ell.json")); print("$ARGS hand", $hand, "ell", $ell, "fhi", $fhi, round(d["ms_per_step"],2), round(d["config"]["krylov_iterations_avg"]["tentative"],2), round(d["timers"]["tentative_velocity_solve"]["avg_ms"],2))
PY
done; done; done
