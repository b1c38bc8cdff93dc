# k = 3, 4: GMRES(8) (default) against the Chebyshev iteration with hand-over (matrix-core lift + vector-kernel step)
for K in 3 4; do for NX in ${NXS:-512}; do
for cfgs in ${CFGS:-0.3:0.4:1.3 0.3:0.6:1.3 0.2:0.5:1.3}; do   # ellipse:hand-over:f_hi
  set -- $(echo $cfgs | tr : " ")
  HDG_CHEB_ELL=$1 HDG_CHEB_HANDOVER=$2 HDG_CHEB_FHI=$3 python bench.py --degree $K --nx $NX --steps 5 --warmup 2 --no-cpu-baseline --tent-solver 1 > gpurun_out/k34.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/k34.json")); print("k=$K nx=$NX ell $1 hand $2 fhi $3", round(d["value"],1), round(d["ms_per_step"],2), round(d["config"]["krylov_iterations_avg"]["tentative"],2), round(d["timers"]["tentative_velocity_solve"]["avg_ms"],2))
PY
done
python bench.py --degree $K --nx $NX --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/k34.json 2>/dev/null
python - <<PY
import json
d=json.load(open("gpurun_out/k34.json")); print("k=$K nx=$NX GMRES default", round(d["value"],1), round(d["ms_per_step"],2), round(d["config"]["krylov_iterations_avg"]["tentative"],2), round(d["timers"]["tentative_velocity_solve"]["avg_ms"],2))
PY
done; done
