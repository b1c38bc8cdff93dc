#!/bin/bash
# usage (GPU box, repo root): bash tools/other_configs.sh OUTDIR
# The BASELINE configurations other than C3 on ONE MI355X (last table): per-step times from the
# PerformanceLog summary the driver prints (label `timestep`), bench lines for the IMEX cases.
O=${1:-gpurun_out/other}
mkdir -p $O
D="python -m incompressibleeulerhdg_amd.driver --output="
# C1: HDG implicit, k=1, 16x16, dt 0.05, 20 steps, projection on / off (monolithic)
$D --timestepper implicit --degree 1 --nx 16 --dt 0.05 --tfinal 1.0 --use_projection_method --fused > $O/c1_proj.log 2>&1
$D --timestepper implicit --degree 1 --nx 16 --dt 0.05 --tfinal 1.0 --fused > $O/c1_mono.log 2>&1
# C2: HDG-IMEX SSP2(3,3,2), k=1, 256^2
python bench.py --degree 1 --nx 256 --steps 10 --warmup 2 --no-cpu-baseline > $O/c2_bench.json 2> $O/c2_bench.err
# C4's problem on one GPU: HDG implicit, k=3, 512^2, projection, dt = 0.25/nx, 5 steps
$D --timestepper implicit --degree 3 --nx 512 --dt 0.00048828125 --tfinal 0.00244140625 --use_projection_method --fused > $O/c4_proj.log 2>&1
# C5's mesh on one GPU: HDG-IMEX k=4, 2048^2
python bench.py --degree 4 --nx 2048 --steps 1 --warmup 1 --no-cpu-baseline > $O/c5_bench.json 2> $O/c5_bench.err
# shear flow on the periodic square with tracer (f-2 / f-3 rows): k=2, 256^2, 10 steps
$D --problem shear --timestepper imex_ssp2_332 --degree 2 --nx 256 --dt 0.006135923 --tfinal 0.06135923 --use_projection_method --tracer_advection --fused > $O/shear_tracer.log 2>&1
grep -H "timestep\|error" $O/*.log | head -40
python - <<PY
import json
for f in ("c2_bench.json", "c5_bench.json"):
    try:
        d = json.load(open("$O/" + f)); print(f, round(d["value"], 1), "MDOF/s", round(d["ms_per_step"], 2), "ms/step")
    except Exception as e:
        print(f, "failed", e)
PY
