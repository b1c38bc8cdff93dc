# scan the row padding of the cell arrays (HDG_ROW_PAD) at one bench configuration: ms/step, tentative solve, pressure solve
# usage: PADS="0 1 2" ARGS="--degree 3 --nx 512" bash tools/padscan.sh      (pad "auto" = the engine's own rule)
for pad in ${PADS:-auto 0 1 2 3 5 8}; do
  if [ "$pad" = auto ]; then unset HDG_ROW_PAD; else export HDG_ROW_PAD=$pad; fi
  python bench.py --steps 4 --warmup 2 --no-cpu-baseline $ARGS > gpurun_out/pad_$pad.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/pad_$pad.json")); print("pad", "$pad", "$ARGS", round(d["value"],1), round(d["ms_per_step"],2), round(d["timers"]["tentative_velocity_solve"]["avg_ms"],2), round(d["timers"]["pressure_solve"]["avg_ms"],2))
PY
done
