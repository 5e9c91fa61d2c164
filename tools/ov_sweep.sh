# tuning sweep of the K-pivot paths: "base:K:tr" triples, base = 4 (fused heads, in-place sweep),
# 5 (overlapped), 6 (one launch per head)   (run on the GPU box)
mkdir -p gpurun_out
for spec in "$@"; do
  IFS=: read base blk tr <<< "$spec"
  LPR_OV_DIAG=1 timeout -k 10 200 python bench.py --block $blk --variant $((base * 0x1000 + tr)) --steps $((blk*96)) --warmup $((blk*8)) --cpu-pivots 0 > gpurun_out/ov_${base}_${blk}_${tr}.json 2> gpurun_out/ov_${base}_${blk}_${tr}.err || { echo "FAILED $spec"; tail -3 gpurun_out/ov_${base}_${blk}_${tr}.err; continue; }
  python - <<PY
import json
d=json.load(open("gpurun_out/ov_${base}_${blk}_${tr}.json")); r=d["roofline"] or {}
print("base=$base K=$blk tr=$tr", "pivots/s", d["value"], "us/pivot", round(d["ms_per_step"]*1e3,2), "sweep_us", r.get("avg_launch_ms") and round(r["avg_launch_ms"]*1e3,1))
PY
  grep "ov diag x" gpurun_out/ov_${base}_${blk}_${tr}.err | grep -v "pivots 0" | tail -1 | cut -c1-200
done
