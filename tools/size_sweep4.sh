# After the wave-granular hand-offs: default path vs forced K-pivot forms at small and mid sizes
mkdir -p gpurun_out; out=gpurun_out/size_sweep4.jsonl; : > $out
for mn in "256 512" "512 1024" "768 1536" "1024 2048" "1536 3072" "2048 2048" "2048 4096"; do
  set -- $mn
  for v in 0 $((0x4008)) $((0x3008)); do
    timeout -k 10 120 python bench.py --m $1 --n $2 --variant $v --steps 256 --warmup 32 --cpu-pivots 0 --no-kernel-timing > gpurun_out/sz.json 2> gpurun_out/sz.err || { echo "{\"m\": $1, \"n\": $2, \"variant\": $v, \"failed\": \"$(tail -1 gpurun_out/sz.err | cut -c1-120)\"}" >> $out; continue; }
    python - >> $out <<PY
import json
d=json.load(open("gpurun_out/sz.json"))
print(json.dumps({"m": $1, "n": $2, "variant": hex($v), "pivots_per_s": d["value"], "us_per_pivot": round(1e6/d["value"],2), "mb": round(($1+1)*($1+$2+1)*8/1e6,1), "pivots_per_step": d.get("pivots_per_step")}))
PY
  done
done
cat $out
