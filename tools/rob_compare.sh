for cfg in "2 512" "1 256" "1 512" "2 256"; do
  set -- $cfg
  echo "== new defaults k=$1 nx=$2"; python tools/robustness_sweep.py $1 $2 2>/dev/null | grep "solver="
  echo "== old (ell 0.5, no hand-over) k=$1 nx=$2"; HDG_CHEB_ELL=0.5 HDG_CHEB_HANDOVER=0 python tools/robustness_sweep.py $1 $2 2>/dev/null | grep "solver="
done
