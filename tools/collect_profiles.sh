#!/bin/bash
# Refresh the judged profile artefacts (run on the GPU box):   bash tools/collect_profiles.sh TAG      e.g. TAG=r02_a
# writes gpurun_out/prof_TAG/:
#   TAG_bench.json                 the default bench line (all legs)
#   TAG_bench_kernel_stats.csv     rocprofv3 --kernel-trace --stats of `bench.py --workload full`
#   TAG_pmc_fetch_size.csv / TAG_pmc_write_size.csv / traffic.json   separate --pmc FETCH_SIZE / WRITE_SIZE passes
#   TAG_sq_full.txt                per-kernel SQ counter summary (two --pmc passes of 8 counters): VALU / LDS / wait cycles
#   TAG_corr_kernel_stats.csv, TAG_corr_traffic.txt, TAG_corr_bench.json        the same for `--workload corr` (configs[3])
#   TAG_corr_mfma.txt / .json      MFMA counters of the correlation kernels (SQ_VALU_MFMA_BUSY_CYCLES ... ) + mfma_util
#   TAG_sgbm_kernel_stats.csv, TAG_sgbm_bench.json                               `--workload sgbm` (configs[1])
# Every counter pass is its own rocprofv3 run with --pmc only (no trace domains), as the MI355X guide asks.
TAG=${1:-r02_x}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_$TAG; rm -rf $O; mkdir -p $O
FULL="bench.py --steps 5 --no-cpu-baseline --no-e2e --workload full"
timeout -k 10 600 python3 bench.py > $O/${TAG}_bench.json 2> $O/bench.err || exit 1
echo "bench done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt -o run --output-format csv -- python3 $FULL > $O/kt.log 2>&1 || exit 1
cp $O/kt/run_kernel_stats.csv $O/${TAG}_bench_kernel_stats.csv
echo "kernel stats done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $O/pf -o run --output-format csv -- python3 bench.py --steps 3 --no-cpu-baseline --no-e2e --workload full > $O/pf.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $O/pw -o run --output-format csv -- python3 bench.py --steps 3 --no-cpu-baseline --no-e2e --workload full > $O/pw.log 2>&1 || exit 1
FPL=$(python3 -c "import json,sys; print(json.load(open('$O/${TAG}_bench.json'))['config']['frames_per_step_per_gpu'])")
python3 tools/make_traffic.py $O/pf/run_counter_collection.csv $O/pw/run_counter_collection.csv $O $TAG $FPL
echo "traffic done"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU \
    -d $O/sq1 -o run --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --workload full > $O/sq1.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVES \
    -d $O/sq2 -o run --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --workload full > $O/sq2.log 2>&1 || exit 1
python3 tools/pmc_sq_summary.py $O/sq1/run_counter_collection.csv $O/sq2/run_counter_collection.csv > $O/${TAG}_sq_full.txt
echo "sq done"
timeout -k 10 300 python3 bench.py --workload corr > $O/${TAG}_corr_bench.json 2>> $O/bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/ktc -o run --output-format csv -- python3 bench.py --workload corr > $O/ktc.log 2>&1 || exit 1
cp $O/ktc/run_kernel_stats.csv $O/${TAG}_corr_kernel_stats.csv
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $O/pfc -o run --output-format csv -- python3 bench.py --workload corr > $O/pfc.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $O/pwc -o run --output-format csv -- python3 bench.py --workload corr > $O/pwc.log 2>&1 || exit 1
python3 tools/pmc_sq_summary.py $O/pfc/run_counter_collection.csv $O/pwc/run_counter_collection.csv > $O/${TAG}_corr_traffic.txt
# MFMA utilisation of the correlation path (north_star: "rocprof HBM GB/s and MFMA utilisation"): its own --pmc pass
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d $O/pmc_mfma -o run --output-format csv -- python3 bench.py --workload corr > $O/pmfma.log 2>&1 || exit 1
python3 tools/corr_mfma.py $O/pmc_mfma/run_counter_collection.csv $O/pfc/run_counter_collection.csv $O/pwc/run_counter_collection.csv $O $TAG
rm -rf $O/pmc_mfma
echo "corr done"
timeout -k 10 300 python3 bench.py --workload sgbm --no-cpu-baseline > $O/${TAG}_sgbm_bench.json 2>> $O/bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kts -o run --output-format csv -- python3 bench.py --steps 5 --no-cpu-baseline --workload sgbm > $O/kts.log 2>&1 || exit 1
cp $O/kts/run_kernel_stats.csv $O/${TAG}_sgbm_kernel_stats.csv
rm -rf $O/kt $O/pf $O/pw $O/sq1 $O/sq2 $O/ktc $O/pfc $O/pwc $O/kts
echo "all done"
