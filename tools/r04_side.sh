#!/bin/bash
# usage: tools/r04_side.sh <outdir>   (GPU box, repo root): side-stream priority / encoder-products placement A/B of the config-3/5 step
out=$1; mkdir -p $out
for dt in bf16 f32; do
  for pr in normal low; do
    for es in 0 1; do
      FOV_SIDE_PRIORITY=$pr FOV_WGRAD_ENC_SIDE=$es timeout -k 10 120 python3 bench.py --mode train_mixing --dtype $dt --no-cpu-baseline > $out/${dt}_${pr}_${es}.json 2> $out/${dt}_${pr}_${es}.err || exit 1
      python3 - <<PY
import json
d = json.loads(open("$out/${dt}_${pr}_${es}.json").read().strip().splitlines()[-1])
print("$dt side=$pr enc_side=$es  ms_per_step %.4f" % d["ms_per_step"])
PY
    done
  done
done
