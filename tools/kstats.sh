#!/bin/bash
# per-kernel time of PROG (default tools/gf_one.py) -> gpurun_out/kstats.txt
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/kst
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/kst -o run --output-format csv -- python3 ${PROG:-tools/gf_one.py} > gpurun_out/kst.log 2>&1 || exit 1
python3 - <<'PY' > gpurun_out/kstats.txt
import csv
rows = list(csv.DictReader(open("gpurun_out/kst/run_kernel_stats.csv")))
for r in rows[:12]:
    print(f"{r['Name'][:60]:60s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:10.1f} pct={r['Percentage']}")
PY
