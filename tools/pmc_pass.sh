#!/bin/bash
# One rocprofv3 counter pass over the kernel micro-benchmarks: tools/pmc_pass.sh NAME "COUNTER ..." [time_kernels args]
# (counters in their own run with --kernel-trace only; the program itself follows `--`).
name=$1; shift; ctrs=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$name
cd /tmp && export TMPDIR=/tmp
rm -rf $out && mkdir -p $out
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/time_kernels.py "$@" > $out/run.log 2>&1
f=$(find $out -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:70]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (k, r["Dispatch_Id"])
    if key not in seen: seen.add(key); cnt[k] += 1
for k in acc:
    if cnt[k] < 5: continue
    print(k, "dispatches", cnt[k], " ".join(f"{c}={v / cnt[k]:.4g}" for c, v in sorted(acc[k].items())))
PY
