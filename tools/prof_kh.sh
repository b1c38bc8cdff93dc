#!/bin/bash
# rocprofv3 kernel statistics of the Kelvin-Helmholtz run on the unit disk (general-mesh path): level 6, k = 2, dt 0.005, 6 steps
# (7 with the warm-up step of tools/kh_bench.py).  usage (GPU box, repo root): bash tools/prof_kh.sh [OUT=gpurun_out/kh_kernel_stats.csv]
OUT=$GRAFT_REPO_ROOT/${1:-gpurun_out/kh_kernel_stats.csv}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_kh
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kh -o s -- python3 $R/tools/kh_bench.py 6 2 6 > $R/gpurun_out/prof_kh.log 2>&1
f=$(find /tmp/prof_kh -name "*kernel_stats.csv" | head -1)
cp "$f" "$OUT"
cd $R
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$OUT")))
tot = sum(float(x["TotalDurationNs"]) for x in rows)
print("total kernel time %.1f ms" % (tot / 1e6))
for x in rows[:22]:
    print(x["Name"][:100].ljust(100), x["Calls"].rjust(6), "%9.1f us %6.2f%%" % (float(x["AverageNs"]) / 1e3, float(x["TotalDurationNs"]) / tot * 100))
PY
grep "disk level" $R/gpurun_out/prof_kh.log
