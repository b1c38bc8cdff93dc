cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_kh -o s -- python3 -m incompressibleeulerhdg_amd.driver --problem kelvinhelmholtz --refinement 6 --degree 2 --dt 0.005 --tfinal 0.03 --timestepper imex_ssp2_332 --use_projection_method --richardson 2 > $GRAFT_REPO_ROOT/gpurun_out/prof_kh.log 2>&1
cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/prof_kh -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print(f"{r['Name'][:70]:70s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:9.1f} us {r['Percentage']}")
PY
rm -rf gpurun_out/prof_kh
