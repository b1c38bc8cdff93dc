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
for h in 0.1 0.15 0.2 0.25 0.3 0.35; do
  for m in 6 8; do run "c3 handover=$h smax=$m" HDG_CHEB_HANDOVER=$h HDG_SSTEP_MAX=$m -- $C3; done
done
run "c3 handover=0.25 ell=0.2" HDG_CHEB_HANDOVER=0.25 HDG_CHEB_ELL=0.2 -- $C3
run "c3 handover=0.25 ell=0.25" HDG_CHEB_HANDOVER=0.25 HDG_CHEB_ELL=0.25 -- $C3
run "c3 handover=0.25 ell=0.35" HDG_CHEB_HANDOVER=0.25 HDG_CHEB_ELL=0.35 -- $C3
run "c3 h=0.25 20+5" HDG_CHEB_HANDOVER=0.25 -- --steps 20 --warmup 5
C2="--steps 20 --warmup 5 --nx 256 --degree 1"
run "c2 default" X=1 -- $C2
run "c2 handover=0.3" HDG_CHEB_HANDOVER=0.3 -- $C2
run "c2 handover=0.5" HDG_CHEB_HANDOVER=0.5 -- $C2
