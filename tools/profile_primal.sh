# headline profile set (run on the GPU box from the repo root): bench line, kernel trace, PMC passes
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python bench.py > gpurun_out/bench_primal.json 2> gpurun_out/bench_primal.err
cat gpurun_out/bench_primal.json
rm -rf gpurun_out/prof_kt gpurun_out/prof_fetch gpurun_out/prof_write
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt -o kt -- python3 bench.py --steps 512 --warmup 64 --cpu-pivots 0 > gpurun_out/prof_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -o f -- python3 bench.py --steps 96 --warmup 16 --cpu-pivots 0 > gpurun_out/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_write -o w -- python3 bench.py --steps 96 --warmup 16 --cpu-pivots 0 > gpurun_out/prof_write.log 2>&1
python tools/pmc_summary.py gpurun_out/prof_fetch gpurun_out/prof_write k_ov2_sweep 16 4096 8192 > gpurun_out/pmc_summary.json
find gpurun_out/prof_kt -name "*stats*" | head
