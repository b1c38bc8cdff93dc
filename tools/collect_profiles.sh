#!/bin/bash
# usage (on the GPU box, repo root): bash tools/collect_profiles.sh TAG
# Kernel stats, HBM-traffic PMC passes (FETCH_SIZE and WRITE_SIZE in SEPARATE runs, counters without any other trace
# domain, the program itself after `--`: /opt/skills/guides/MI355X_MICROARCH.md, HBM / rocprofv3 PMC slots) and the
# matrix-core counters of the k = 3, 4 kernels.  tools/summarise_profiles.py turns the raw CSVs under gpurun_out/ into
# the tracked files profiles/<TAG>_* and profiles/pmc_traffic.json (tagged with the hash of the kernel sources).
TAG=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline"
T="python3 $R/tools/time_kernels.py"
# C3: kernel stats of the benchmark itself
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3_stats -o s -- $B --steps 5 --warmup 1 > $O/c3_stats.log 2>&1
# C3: HBM traffic of the kernels inside the benchmark (2 timed steps), separate passes
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/c3_fetch -o f -- $B --steps 2 --warmup 1 > $O/c3_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/c3_write -o w -- $B --steps 2 --warmup 1 > $O/c3_write.log 2>&1
# calibration of FETCH_SIZE / WRITE_SIZE on a kernel with a known byte count (stream triad on velocity vectors)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/cal_fetch -o f -- $T 2 1024 8 > $O/cal_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/cal_write -o w -- $T 2 1024 8 > $O/cal_write.log 2>&1
# k = 3, 4 at 512^2: kernel stats and matrix-core counters
for K in 3 4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/k${K}_stats -o s -- $B --degree $K --nx 512 --steps 3 --warmup 1 > $O/k${K}_stats.log 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/k${K}_mfma -o m -- $B --degree $K --nx 512 --steps 2 --warmup 1 > $O/k${K}_mfma.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/k${K}_fetch -o f -- $B --degree $K --nx 512 --steps 2 --warmup 1 > $O/k${K}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/k${K}_write -o w -- $B --degree $K --nx 512 --steps 2 --warmup 1 > $O/k${K}_write.log 2>&1
done
cd $R
# summarise here: the raw counter / trace CSVs are larger than what gpurun copies back
S=$R/gpurun_out/summary_$TAG
python3 tools/summarise_profiles.py $TAG $S > $R/gpurun_out/summary_$TAG.log 2>&1
tail -25 $R/gpurun_out/summary_$TAG.log
rm -rf $O
# the bench lines of the same build, with roofline.traffic from the file just written (same source hash)
cp $S/pmc_traffic.json $R/profiles/pmc_traffic.json
# (the driver's window: 20 timed steps after 5 warm-up steps -- the first solves of a run still learn their check schedule)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $S/${TAG}_bench_c3.json 2> $R/gpurun_out/bench_c3_$TAG.err
python3 bench.py --degree 3 --nx 512 --steps 20 --warmup 5 --no-cpu-baseline > $S/${TAG}_bench_k3.json 2> $R/gpurun_out/bench_k3_$TAG.err
python3 bench.py --degree 4 --nx 512 --steps 20 --warmup 5 --no-cpu-baseline > $S/${TAG}_bench_k4.json 2> $R/gpurun_out/bench_k4_$TAG.err
ls $S
