#!/bin/bash
# Per-kernel times of the revised iteration (run on the GPU box from the repo root):
#   tools/prof_revised.sh <tag> [iters]
set -e
tag=${1:-rev}; iters=${2:-128}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python tools/bench_revised.py --iters 256 --gemm-reps 1 | tee gpurun_out/${tag}_bench.json
rm -rf gpurun_out/${tag}_kt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_kt -o kt -- python3 tools/bench_revised.py --iters $iters --gemm-reps 1 > gpurun_out/${tag}_kt.log 2>&1
f=$(find gpurun_out/${tag}_kt -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print("%-70s calls %5s avg %9.1f ns  %5s %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]), r["Percentage"]))
PY
cp "$f" gpurun_out/${tag}_kernel_stats.csv
