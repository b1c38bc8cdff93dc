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
for e in 16 64 256; do
  run "c3 20+5 every=$e" HDG_CHEB_EVERY=$e -- --steps 20 --warmup 5
done
run "c3 40+5 every=64" HDG_CHEB_EVERY=64 -- --steps 40 --warmup 5
run "k3 20+5 every=16" HDG_CHEB_EVERY=16 -- --steps 20 --warmup 5 --nx 512 --degree 3
run "k3 20+5 every=64" HDG_CHEB_EVERY=64 -- --steps 20 --warmup 5 --nx 512 --degree 3
run "k4 20+5 every=16" HDG_CHEB_EVERY=16 -- --steps 20 --warmup 5 --nx 512 --degree 4
run "k4 20+5 every=64" HDG_CHEB_EVERY=64 -- --steps 20 --warmup 5 --nx 512 --degree 4
