# usage: bash tools/r2_run.sh <tag> <step> [<step> ...]   steps: tests-focus | tests-all | probe | bench | profile
# A step that times out (124/137) or dies on a signal stops the script: no GPU step after a hang.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
tag=$1; shift
run() {  # name, timeout, command...
    local name=$1 to=$2; shift 2
    echo "=== $name ($(date +%T))"
    timeout -k 10 "$to" "$@" > "gpurun_out/${tag}_${name}.log" 2>&1
    local rc=$?
    tail -n 15 "gpurun_out/${tag}_${name}.log"
    echo "=== $name rc=$rc"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -ge 128 ]; then echo "STOP: $name hung or was killed"; exit $rc; fi
    return 0
}
for step in "$@"; do
  case $step in
    tests-focus) run tests_focus 900 python -m pytest tests/test_block_gpu.py tests/test_primal_gpu.py tests/test_cut_gpu.py -m gpu -x -q ;;
    tests-all) run tests_all 1100 python -m pytest tests -m gpu -x -q ;;
    tests-rest) run tests_rest 1100 python -m pytest tests -m gpu -x -q --deselect tests/test_block_gpu.py --deselect tests/test_primal_gpu.py --deselect tests/test_cut_gpu.py ;;
    tests-rev) run tests_rev 900 python -m pytest tests/test_revised_gpu.py tests/test_program_gpu.py tests/test_configs_gpu.py -m gpu -x -q ;;
    bench-m512) run bench_m512 300 python bench.py --m 512 --n 1024 --steps 96 --warmup 8 ;;
    tests-bb) run tests_bb 900 python -m pytest tests/test_bb_gpu.py tests/test_configs_gpu.py -m gpu -x -q ;;
    bench-bb) run bench_bb 600 python bench.py --workload bb ;;
    bench-revised) run bench_revised 600 python bench.py --workload revised --steps 200 --warmup 16 ;;
    bench-configs)
      run bench_m512 300 python bench.py --m 512 --n 1024 --steps 96 --warmup 8
      run bench_m2048 300 python bench.py --m 2048 --n 2048 --steps 128 --warmup 16
      run bench_block1 300 python bench.py --block 1 --steps 256 --warmup 32
      run bench_sens 600 python bench.py --workload sens --steps 32 --warmup 2 ;;
    profile-bb)
      rm -rf gpurun_out/${tag}_bbkt
      run prof_bb 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_bbkt -o kt -- python3 bench.py --workload bb --cpu-pivots 0
      find gpurun_out/${tag}_bbkt -name "*kernel_stats*" -exec cp {} gpurun_out/${tag}_bb_kernel_stats.csv \;
      find gpurun_out/${tag}_bbkt -name "*kernel_trace*" -delete ;;
    rehearse-n2)  # the N > 1 code path of bench.py with 2 ranks on this box's one GPU (gloo)
      export LPR_BENCH_SHARED_GPU=1
      run rehearse_primal 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5
      run rehearse_bb 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 2 --workload bb
      run rehearse_revised 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --workload revised --steps 32 --warmup 4
      unset LPR_BENCH_SHARED_GPU ;;
    trace-gaps)
      rm -rf gpurun_out/${tag}_gaps
      run prof_gaps 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${tag}_gaps -o kt -- python3 bench.py --steps 64 --warmup 8 --cpu-pivots 0
      python tools/hop_from_trace.py $(find gpurun_out/${tag}_gaps -name "*kernel_trace.csv" | head -n 1) > gpurun_out/${tag}_step_gaps.json
      cat gpurun_out/${tag}_step_gaps.json
      rm -rf gpurun_out/${tag}_gaps ;;
    profile-revised)
      rm -rf gpurun_out/${tag}_revkt
      run prof_rev 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_revkt -o kt -- python3 bench.py --workload revised --steps 200 --warmup 16 --cpu-pivots 0
      find gpurun_out/${tag}_revkt -name "*kernel_stats*" -exec cp {} gpurun_out/${tag}_revised_kernel_stats.csv \;
      find gpurun_out/${tag}_revkt -name "*kernel_trace*" -delete ;;
    probe) run probe 600 python tools/r2_probe.py ;;
    probe-quick) run probe 300 python tools/r2_probe.py --quick ;;
    bench) run bench 600 python bench.py ; run bench_driver 600 python bench.py --gpus 1 --steps 20 --warmup 5 ;;
    profile)
      rm -rf gpurun_out/${tag}_kt gpurun_out/${tag}_fetch gpurun_out/${tag}_write
      run prof_kt 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_kt -o kt -- python3 bench.py --steps 64 --warmup 8 --cpu-pivots 0
      run prof_fetch 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_fetch -o f -- python3 bench.py --steps 24 --warmup 4 --cpu-pivots 0
      run prof_write 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_write -o w -- python3 bench.py --steps 24 --warmup 4 --cpu-pivots 0
      python tools/pmc_summary.py gpurun_out/${tag}_fetch gpurun_out/${tag}_write k_ov2_sweep 16 4096 8192 > gpurun_out/${tag}_pmc_summary.json
      find gpurun_out/${tag}_kt -name "*kernel_stats*" -exec cp {} gpurun_out/${tag}_kernel_stats.csv \;
      # the raw traces are large: keep the summaries only
      rm -rf gpurun_out/${tag}_fetch gpurun_out/${tag}_write
      find gpurun_out/${tag}_kt -name "*kernel_trace*" -delete
      ;;
    *) echo "unknown step $step"; exit 2 ;;
  esac
done
echo "=== all steps done"
