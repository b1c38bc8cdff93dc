run() { # k nx extra-env...
  k=$1; nx=$2; shift 2
  env "$@" timeout -k 10 300 python bench.py --degree $k --nx $nx --steps 3 --no-cpu-baseline > gpurun_out/hy.log 2>&1
  python - "$k $nx $*" <<PY
import json,sys
try:
    d=json.loads(open("gpurun_out/hy.log").read().strip().splitlines()[-1])
    o=d["roofline"]["other_kernels"]; adv=[v["ms"] for kk,v in o.items() if kk.startswith("k_adv")]
    advms=adv[0] if adv else d["roofline"]["ms_per_launch"]
    print(sys.argv[1], ": %.1f ms/step, %.1f MDOF/s, its"%(d["ms_per_step"],d["value"]), round(d["config"]["krylov_iterations_avg"]["tentative"],1), "adv us %.1f"%(advms*1e3))
except Exception as e:
    print(sys.argv[1], "FAILED", open("gpurun_out/hy.log").read()[-300:])
PY
}
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -2
HDG_ADV_SPLIT=1:4 timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "adv" 2>&1 | tail -2
for k in 3 4; do run $k 512 HDG_ADV_SPLIT_FROM=9; run $k 512 HDG_ADV_SPLIT_FROM=3; done
run 2 1024 HDG_ADV_SPLIT_FROM=9
run 2 1024 HDG_ADV_SPLIT_FROM=2
run 1 1024 HDG_ADV_SPLIT_FROM=9
run 1 1024 HDG_ADV_SPLIT_FROM=1
