#!/bin/bash
# Scan of the tentative-velocity solver parameters around the s-step tail (round 4): ms/step and iterations at C3 and at k = 3, 4 (512^2).
# usage (GPU box): bash tools/sstep_scan.sh > gpurun_out/sstep_scan.log
run() {  # label, env..., -- bench args
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
run "c3 default" X=1 -- $C3
run "c3 tail=gmres" HDG_TAIL_GMRES=1 -- $C3
for e in 0.2 0.4 0.5; do run "c3 ell=$e" HDG_CHEB_ELL=$e -- $C3; done
for h in 0.3 0.45 0.8; do run "c3 handover=$h" HDG_CHEB_HANDOVER=$h -- $C3; done
for p in 1.4 2.0; do run "c3 per_decade=$p" HDG_SSTEP_PER_DECADE=$p -- $C3; done
run "c3 smax=6" HDG_SSTEP_MAX=6 -- $C3
K34="--steps 12 --warmup 4 --nx 512"
for k in 3 4; do
  run "k$k default" X=1 -- $K34 --degree $k
  run "k$k tail=gmres" HDG_TAIL_GMRES=1 -- $K34 --degree $k
  run "k$k handover=0.6" HDG_CHEB_HANDOVER=0.6 -- $K34 --degree $k
  run "k$k handover=0.25" HDG_CHEB_HANDOVER=0.25 -- $K34 --degree $k
done
