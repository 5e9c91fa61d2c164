mkdir -p gpurun_out
for mn in "256 512" "384 768" "512 1024" "640 1280"; do
  set -- $mn
  for cfg in "16 $((0x4008))" "8 $((0x4008))" "0 $((0x7ffe))"; do
    set -- $mn $cfg
    timeout -k 10 120 python bench.py --m $1 --n $2 --block $3 --variant $4 --steps 256 --warmup 32 --cpu-pivots 0 --no-kernel-timing > gpurun_out/sz.json 2> gpurun_out/sz.err || { echo "m=$1 n=$2 block=$3 variant=$4 FAILED: $(tail -1 gpurun_out/sz.err | cut -c1-120)"; continue; }
    python - <<PY
import json
d=json.load(open("gpurun_out/sz.json"))
print("m=$1 n=$2 block=$3 variant=$4", "pivots/s", d["value"], "us/pivot", round(d["ms_per_step"]*1e3,2))
PY
  done
done
