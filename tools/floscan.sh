for flo in 0.9 0.95 1.0; do for fhi in 1.15 1.2 1.3; do
  HDG_CHEB_FLO=$flo HDG_CHEB_FHI=$fhi python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/flo.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/flo.json")); print("flo", $flo, "fhi", $fhi, round(d["value"],1), round(d["ms_per_step"],2), round(d["config"]["krylov_iterations_avg"]["tentative"],2), round(d["timers"]["tentative_velocity_solve"]["avg_ms"],2))
PY
done; done
