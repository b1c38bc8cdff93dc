# builds: e.g.  for v in ADV_NT=6 LIFT_NT=3 ADV_PIPE=0; do hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -DHDG_$v -o build/var/lib_${v/=/_}.so incompressibleeulerhdg_amd/csrc/hdg_engine.hip -L/opt/rocm/lib -lrccl -lrt -lpthread; done
for v in default ADV_NT_6 ADV_NT_5 ADV_NT_3 LIFT_NT_6 LIFT_NT_5 LIFT_NT_3 ADV_PIPE_0 default; do
  if [ $v = default ]; then unset HDG_LIB_PATH; else export HDG_LIB_PATH=$PWD/build/var/lib_$v.so; fi
  python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/var.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/var.json")); r=d["roofline"]
print("$v", round(d["ms_per_step"],2), "tent", round(d["timers"]["tentative_velocity_solve"]["avg_ms"],2), "adv", round(r["ms_per_launch"]*1e3,1), "lift", [round(v["ms"]*1e3,1) for k,v in r["other_kernels"].items() if "Cheb" in k])
PY
done
