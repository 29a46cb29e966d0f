#!/bin/bash
# Refresh the judged profile artefacts from ONE bench configuration (run on the GPU box):
#   bash tools/collect_profiles.sh TAG        e.g. TAG=r01_f
# writes gpurun_out/prof_TAG/{TAG_bench.json, TAG_bench_kernel_stats.csv, TAG_pmc_fetch_size.csv,
#        TAG_pmc_write_size.csv, traffic.json}; copy them into profiles/ afterwards.
# Three separate rocprofv3 passes (kernel trace; --pmc FETCH_SIZE; --pmc WRITE_SIZE) as the MI355X guide asks.
TAG=${1:-r01_x}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_$TAG; rm -rf $O; mkdir -p $O
timeout -k 10 600 python3 bench.py > $O/${TAG}_bench.json 2> $O/bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt -o run --output-format csv -- python3 bench.py --steps 5 --no-cpu-baseline > $O/kt.log 2>&1 || exit 1
cp $O/kt/run_kernel_stats.csv $O/${TAG}_bench_kernel_stats.csv
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $O/pf -o run --output-format csv -- python3 bench.py --steps 3 --no-cpu-baseline > $O/pf.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $O/pw -o run --output-format csv -- python3 bench.py --steps 3 --no-cpu-baseline > $O/pw.log 2>&1 || exit 1
python3 tools/make_traffic.py $O/pf/run_counter_collection.csv $O/pw/run_counter_collection.csv $O $TAG
