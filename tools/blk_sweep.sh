# tuning sweep of the K-pivots-per-sweep path on the headline tableau (run on the GPU box)
set -e
mkdir -p gpurun_out
run() {  # name, args...
  name=$1; shift
  timeout -k 10 200 python bench.py "$@" --steps 512 --warmup 64 --cpu-pivots 0 > gpurun_out/blk_$name.json 2> gpurun_out/blk_$name.err
  python - <<PY
import json
d=json.load(open("gpurun_out/blk_$name.json"))
r=d["roofline"] or {}
print("$name", "value", d["value"], "ms/step", d["ms_per_step"], "launch_ms", r.get("avg_launch_ms"), "frac", r.get("frac"))
PY
}
for spec in "$@"; do
  blk=${spec%%:*}; tr=${spec##*:}
  run b${blk}_tr${tr} --block $blk --variant $((0x6000 + tr))
done
