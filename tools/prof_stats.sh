cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_tile -o s -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 $ARGS > $GRAFT_REPO_ROOT/gpurun_out/prof_tile.log 2>&1
cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/prof_tile -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:int(__import__("os").environ.get("NROWS", "14"))]:
    print(f"{r['Name'][:70]:70s} {r['Calls']:>5s} {float(r['AverageNs'])/1e3:8.1f} us {r['Percentage']}")
PY
rm -rf gpurun_out/prof_tile
