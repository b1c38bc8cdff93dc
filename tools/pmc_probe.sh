#!/bin/bash
# usage (GPU box, repo root): bash tools/pmc_probe.sh TAG "COUNTER COUNTER ..." [bench args]
# One --pmc pass of the C3 benchmark (2 timed steps) with the given counters; prints per-kernel averages per launch for
# the dominant kernels.  Counters alone, no other trace domain, the program directly after `--` (see collect_profiles.sh).
TAG=$1; CNT=$2; shift 2
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmcprobe_$TAG
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $O -o p -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@" > $O.log 2>&1
cd $R
python3 - "$O" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counter file"); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in csv.DictReader(open(f[0])):
    name = r["Kernel_Name"].split("(")[0].replace("void hdg::", "").replace("hdg::", "")
    acc[name][r["Counter_Name"]] += float(r["Counter_Value"]); n[name].add(r["Dispatch_Id"])
keys = sorted({c for v in acc.values() for c in v})
want = ("k_adv_apply<2, true>", "k_edge_lift<2, false, 2, true>", "k_trace_post_tile<2, true>", "k_trace_pre_tile<2>", "k_trace_pre_tile<2, false>", "k_cg_sr_update", "k_cg_sr_update_r", "k_p1_down<2>", "k_p1_up<2>", "k_edge_lift_pair<2, false, 2, false>", "k_gs_update<32, true, double>", "k_adv_apply<2, false>")
for name in want:
    if name in acc:
        print(name, "launches", len(n[name]), " ".join(f"{c}={acc[name][c] / len(n[name]):.4g}" for c in keys))
PY
rm -rf $O
