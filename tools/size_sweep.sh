#!/bin/bash
# Which primal path is fastest at which tableau size (run on the GPU box from the repo root).
#   tools/size_sweep.sh [steps] [warmup] -- "<m> <n>" ... -- <variant> ...
# e.g. tools/size_sweep.sh 256 32 -- "256 512" "512 1024" "1000 1000" -- 0 0x2000 0x4008 0x3008
# writes gpurun_out/size_sweep.jsonl (one line per size x variant; 0 = the engine's own choice).
# (This one script replaces the four near-copies of round 2; their size / variant lists were:
#  small: 256x512 .. 640x1280 with 0x4008, 0x7ffe; mid: 768x1536 .. 3072x6144 with 0x4008, 0x6008,
#  0x7ffe, block 1; large: 768x1536 .. 4096x8192 with 0x4008 vs 0x3008.)
steps=${1:-256}; warm=${2:-32}; shift 2 2>/dev/null
[ "${1:-}" = "--" ] && shift
sizes=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do sizes+=("$1"); shift; done
[ "${1:-}" = "--" ] && shift
variants=("$@")
[ ${#sizes[@]} -eq 0 ] && sizes=("256 512" "512 1024" "768 1536" "1024 2048" "2048 2048" "2048 4096" "4096 8192")
[ ${#variants[@]} -eq 0 ] && variants=(0 0x2000 0x4008 0x3008)
mkdir -p gpurun_out; out=gpurun_out/size_sweep.jsonl; : > $out
for mn in "${sizes[@]}"; do
  set -- $mn
  for v in "${variants[@]}"; do
    vd=$((v))
    timeout -k 10 120 python bench.py --m $1 --n $2 --variant $vd --steps $steps --warmup $warm --cpu-pivots 0 --no-kernel-timing > gpurun_out/sz.json 2> gpurun_out/sz.err || { echo "{\"m\": $1, \"n\": $2, \"variant\": \"$v\", \"failed\": \"$(tail -1 gpurun_out/sz.err | cut -c1-120)\"}" >> $out; continue; }
    python3 - "$1" "$2" "$v" >> $out <<'PY'
import json, sys
m, n, v = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
d = json.load(open("gpurun_out/sz.json"))
print(json.dumps({"m": m, "n": n, "variant": v, "pivots_per_s": d["value"],
                  "us_per_pivot": round(1e6 / d["value"], 2), "mb": round((m + 1) * (m + n + 1) * 8 / 1e6, 1),
                  "pivots_per_step": d.get("pivots_per_step")}))
PY
  done
done
cat $out
