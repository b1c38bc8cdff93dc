# A/B of two builds of the library on one box: usage  LIB=build/var/lib_x.so [ARGS="--degree 2 --nx 512"] bash tools/ab_lib.sh
for rep in 1 2 3; do for v in default variant; do
  if [ $v = default ]; then unset HDG_LIB_PATH; else export HDG_LIB_PATH=$PWD/$LIB; fi
  python bench.py --steps 6 --warmup 2 --no-cpu-baseline $ARGS > gpurun_out/ab.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/ab.json")); r=d["roofline"]
print("$v $ARGS", round(d["value"],1), round(d["ms_per_step"],2), "tent", round(d["timers"]["tentative_velocity_solve"]["avg_ms"],3), "press", round(d["timers"]["pressure_solve"]["avg_ms"],3), "adv", round(r["ms_per_launch"]*1e3,1), "lift", [round(v["ms"]*1e3,1) for k,v in r["other_kernels"].items() if "Cheb" in k])
PY
done; done
