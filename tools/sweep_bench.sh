# Builds (here, cross-compiled) or runs (on the GPU box) the sweep micro-benchmark variants.
#   bash tools/sweep_bench.sh build      -> tools/_bin/sweep_bench_d<bits>
#   bash tools/sweep_bench.sh run <tag>  -> gpurun_out/<tag>_sweep_bench.jsonl
# RUNS: lines of "<diag> <args of sweep_bench>"
set -u
case ${1:-} in
  build)
    mkdir -p tools/_bin
    for d in ${DIAGS:-0 1 3}; do
      /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -DLPR_BUILD \
        -DLPR_OV_KERNELS_ONLY -DLPR_OV_DIAG=$d ${EXTRA:-} -I lpr_381_group_v22_amd/csrc -I include -Wno-unused-function \
        -Wno-unused-value tools/sweep_bench.hip -o tools/_bin/sweep_bench_d$d${SUFFIX:-} &
    done
    wait ;;
  run)
    tag=$2; out=gpurun_out/${tag}_sweep_bench.jsonl; mkdir -p gpurun_out; : > $out
    while read -r d args; do
      [ -z "$d" ] && continue
      timeout -k 5 60 tools/_bin/sweep_bench_d$d $args >> $out 2>&1 || { echo "STOP d$d $args"; cat $out; exit 1; }
    done < tools/_bin/runs.txt
    cat $out ;;
esac
