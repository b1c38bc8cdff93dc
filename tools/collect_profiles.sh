# usage (on the GPU box, repo root): bash tools/collect_profiles.sh TAG
# kernel stats, the two PMC passes (separate runs, as the guide prescribes) and the default bench line
set -e
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o s -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_${TAG}_fetch -o f -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_${TAG}_write -o w -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_write.log 2>&1
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -c 600 gpurun_out/${TAG}_bench.json
