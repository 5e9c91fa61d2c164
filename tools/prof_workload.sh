#!/bin/bash
# Profile set of one bench.py workload (run on the GPU box from the repo root):
#   tools/prof_workload.sh <tag> <bench.py arguments...>
# -> gpurun_out/<tag>_bench.json, <tag>_kernel_stats.csv, <tag>_pmc_summary.json
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python bench.py "$@" > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || { tail -5 gpurun_out/${tag}_bench.err; exit 1; }
cat gpurun_out/${tag}_bench.json
rm -rf gpurun_out/${tag}_kt gpurun_out/${tag}_fetch gpurun_out/${tag}_write
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_kt -o kt -- python3 bench.py "$@" --cpu-pivots 0 > gpurun_out/${tag}_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_fetch -o f -- python3 bench.py "$@" --cpu-pivots 0 > gpurun_out/${tag}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_write -o w -- python3 bench.py "$@" --cpu-pivots 0 > gpurun_out/${tag}_write.log 2>&1
cp "$(find gpurun_out/${tag}_kt -name '*kernel_stats.csv' | head -1)" gpurun_out/${tag}_kernel_stats.csv
cd tools && python3 pmc_kernels.py ../gpurun_out/${tag}_fetch ../gpurun_out/${tag}_write > ../gpurun_out/${tag}_pmc_summary.json && cd ..
head -12 gpurun_out/${tag}_kernel_stats.csv
rm -rf gpurun_out/${tag}_kt gpurun_out/${tag}_fetch gpurun_out/${tag}_write
