# old (ellipse 0.5, no hand-over) vs new defaults of the tentative-velocity Chebyshev iteration: MDOF/s, ms/step, iterations, ms per solve
for args in ${CASES:-"--degree 2 --nx 1024" "--degree 2 --nx 512" "--degree 2 --nx 256" "--degree 2 --nx 128" "--degree 1 --nx 256" "--degree 1 --nx 1024"}; do
  for mode in new old; do
    if [ $mode = old ]; then export HDG_CHEB_ELL=0.5 HDG_CHEB_HANDOVER=0; else unset HDG_CHEB_ELL HDG_CHEB_HANDOVER; fi
    python bench.py --steps 6 --warmup 2 --no-cpu-baseline $args > gpurun_out/cmp.json 2>/dev/null
    python - <<PY
import json
d=json.load(open("gpurun_out/cmp.json")); print("$mode", "$args", round(d["value"],1), round(d["ms_per_step"],2), round(d["config"]["krylov_iterations_avg"]["tentative"],2), round(d["timers"]["tentative_velocity_solve"]["avg_ms"],2))
PY
  done
done
