# A/B on ONE box: wave-granular hand-offs (the library) vs workgroup-granular (_lib_alt), alternating.
#   bash tools/ab_handoff.sh build   (here: builds lpr_381_group_v22_amd/_lib_alt with -DLPR_OV_WAVE_HANDOFF=0)
#   gpurun -- 'bash tools/ab_handoff.sh'   (on the box; writes gpurun_out/ab_handoff.jsonl)
set -u
if [ "${1:-}" = build ]; then
  cd lpr_381_group_v22_amd/csrc && make && mkdir -p _obj_alt ../_lib_alt &&
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall \
    -Wno-unused-function -Wno-unused-value -DLPR_BUILD -DLPR_OV_WAVE_HANDOFF=0 -c overlap_kernels.hip \
    -o _obj_alt/overlap_kernels.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../_lib_alt/liblpr_engine.so \
    $(ls _obj/*.o | grep -v overlap_kernels.o) _obj_alt/overlap_kernels.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
  exit $?
fi
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out; out=gpurun_out/ab_handoff.jsonl; : > $out
cp lpr_381_group_v22_amd/_lib/liblpr_engine.so /tmp/lib_wave.so
cp lpr_381_group_v22_amd/_lib_alt/liblpr_engine.so /tmp/lib_wg.so
# whatever happens below, the product library is the one left installed
trap 'cp /tmp/lib_wave.so lpr_381_group_v22_amd/_lib/liblpr_engine.so' EXIT
for rep in 1 2 3; do
  for which in wave wg; do
    cp /tmp/lib_$which.so lpr_381_group_v22_amd/_lib/liblpr_engine.so
    for args in "--steps 64 --warmup 8" "--steps 64 --warmup 8 --no-kernel-timing" "--m 2048 --n 4096 --steps 96 --warmup 16 --no-kernel-timing" "--m 1024 --n 2048 --steps 96 --warmup 16 --no-kernel-timing"; do
      timeout -k 10 120 python bench.py $args --cpu-pivots 0 > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "FAILED $which $args"; tail -2 gpurun_out/ab.err; exit 1; }
      python - "$which" "$args" >> $out <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab.json"))
print(json.dumps({"lib": sys.argv[1], "args": sys.argv[2], "pivots_per_s": d["value"]}))
PY
    done
  done
done
cp /tmp/lib_wave.so lpr_381_group_v22_amd/_lib/liblpr_engine.so
cat $out
