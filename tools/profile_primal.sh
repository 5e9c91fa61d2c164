#!/bin/bash
# headline profile set (run on the GPU box from the repo root):  tools/profile_primal.sh <tag>
# -> gpurun_out/<tag>_bench_primal.json (+ _driver_steps20), <tag>_bench_kernel_stats.csv (rocprofv3
#    --kernel-trace --stats of the same command), <tag>_bench_pmc_summary.json (two --pmc passes)
set -e
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python bench.py > gpurun_out/${tag}_bench_primal.json 2> gpurun_out/${tag}_bench_primal.err
cat gpurun_out/${tag}_bench_primal.json
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_bench_primal_driver_steps20.json 2>> gpurun_out/${tag}_bench_primal.err
rm -rf gpurun_out/prof_kt gpurun_out/prof_fetch gpurun_out/prof_write
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt -o kt -- python3 bench.py --cpu-pivots 0 > gpurun_out/prof_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -o f -- python3 bench.py --steps 24 --warmup 4 --cpu-pivots 0 > gpurun_out/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_write -o w -- python3 bench.py --steps 24 --warmup 4 --cpu-pivots 0 > gpurun_out/prof_write.log 2>&1
python tools/pmc_summary.py gpurun_out/prof_fetch gpurun_out/prof_write k_ov2_sweep 16 4096 8192 > gpurun_out/${tag}_bench_pmc_summary.json
cp "$(find gpurun_out/prof_kt -name '*kernel_stats.csv' | head -1)" gpurun_out/${tag}_bench_kernel_stats.csv
head -6 gpurun_out/${tag}_bench_kernel_stats.csv
rm -rf gpurun_out/prof_kt gpurun_out/prof_fetch gpurun_out/prof_write
