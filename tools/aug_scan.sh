#!/bin/bash
# LGMRES-type augmentation of the s-step cycles (HDG_SSTEP_AUG = 0, 1, 2): C3 and k = 3, 4 at 512^2
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
for a in 0 1 2; do
  run "c3 aug=$a" HDG_SSTEP_AUG=$a -- --steps 10 --warmup 5
  run "k3 aug=$a" HDG_SSTEP_AUG=$a -- --steps 12 --warmup 4 --nx 512 --degree 3
  run "k4 aug=$a" HDG_SSTEP_AUG=$a -- --steps 12 --warmup 4 --nx 512 --degree 4
done
