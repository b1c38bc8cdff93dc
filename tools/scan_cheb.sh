run() { # k nx extra-env...
  k=$1; nx=$2; shift 2
  env "$@" timeout -k 10 300 python bench.py --degree $k --nx $nx --steps 3 --no-cpu-baseline > gpurun_out/hy.log 2>&1
  python - "$k $nx $*" <<PY
import json,sys
try:
    d=json.loads(open("gpurun_out/hy.log").read().strip().splitlines()[-1])
    print(sys.argv[1], ": %.1f ms/step, %.1f MDOF/s, its"%(d["ms_per_step"],d["value"]), round(d["config"]["krylov_iterations_avg"]["tentative"],1))
except Exception as e:
    print(sys.argv[1], "FAILED", open("gpurun_out/hy.log").read()[-300:])
PY
}
for k in 1 2 3 4; do for nx in 128 512; do run $k $nx A=1; done; done
run 1 1024 A=1
run 2 1024 A=1
run 2 1024 HDG_CHEB_ELL=0.6
