#!/bin/bash
run() {
  label=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  env "${envs[@]}" python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$label', '%.1f MDOF/s %.2f ms/step' % (d['value'], d['ms_per_step']), 'its %.2f' % d['config']['krylov_iterations_avg']['tentative'], 'tent %.2f ms press %.2f ms' % (d['timers']['tentative_velocity_solve']['avg_ms'], d['timers']['pressure_solve']['avg_ms']))"
}
C3="--steps 10 --warmup 5"
for mk in 2 4 6 8 10; do
  run "c3 handover=0.3 min_k=$mk" HDG_CHEB_HANDOVER=0.3 HDG_CHEB_MIN_K=$mk -- $C3
done
run "c3 handover=0.3 min_k=4 smax=5" HDG_CHEB_HANDOVER=0.3 HDG_CHEB_MIN_K=4 HDG_SSTEP_MAX=5 -- $C3
run "c3 handover=0.3 min_k=4 smax=7" HDG_CHEB_HANDOVER=0.3 HDG_CHEB_MIN_K=4 HDG_SSTEP_MAX=7 -- $C3
run "c3 handover=0.3 pd=1.4" HDG_CHEB_HANDOVER=0.3 HDG_SSTEP_PER_DECADE=1.4 -- $C3
run "c3 handover=0.3 pd=2.0" HDG_CHEB_HANDOVER=0.3 HDG_SSTEP_PER_DECADE=2.0 -- $C3
K34="--steps 12 --warmup 4 --nx 512"
for k in 3 4; do
  for m in 6 8; do run "k$k smax=$m" HDG_SSTEP_MAX=$m -- $K34 --degree $k; done
done
HDG_CHEB_HANDOVER=0.3 HDG_DEBUG=1 python bench.py --no-cpu-baseline --steps 2 --warmup 1 2> gpurun_out/r04e_debug_h03.err > /dev/null
HDG_DEBUG=1 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --nx 512 --degree 4 2> gpurun_out/r04e_debug_k4.err > /dev/null
