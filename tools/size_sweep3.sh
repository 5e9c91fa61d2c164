# Round 2: which K-pivot form is faster at which tableau size, now that a loop head costs 8-10 us?
# (heads, then in-place sweep 0x4008  vs  two-stream overlap 0x3008); writes gpurun_out/size_sweep3.jsonl
mkdir -p gpurun_out; out=gpurun_out/size_sweep3.jsonl; : > $out
for mn in "768 1536" "1024 2048" "1536 3072" "2048 2048" "2048 4096" "3072 4096" "3072 6144" "4096 8192"; do
  set -- $mn
  for v in $((0x4008)) $((0x3008)); do
    timeout -k 10 120 python bench.py --m $1 --n $2 --variant $v --steps 96 --warmup 16 --cpu-pivots 0 --no-kernel-timing > gpurun_out/sz.json 2> gpurun_out/sz.err || { echo "{\"m\": $1, \"n\": $2, \"variant\": $v, \"failed\": \"$(tail -1 gpurun_out/sz.err | cut -c1-120)\"}" >> $out; continue; }
    python - >> $out <<PY
import json
d=json.load(open("gpurun_out/sz.json"))
print(json.dumps({"m": $1, "n": $2, "variant": hex($v), "pivots_per_s": d["value"], "us_per_pivot": round(1e6/d["value"],2), "mb": round(($1+1)*($1+$2+1)*8/1e6,1)}))
PY
  done
done
cat $out
