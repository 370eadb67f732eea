#!/bin/bash
# usage (GPU box, repo root): tools/final_round.sh <outdir>   - the measurements a round's profiles/ are refreshed from
out=$1; mkdir -p $out
python3 bench.py --steps 50 --warmup 5 > $out/bench.json 2> $out/bench.err
for m in train infer_mixing train_mixing; do python3 bench.py --mode $m --steps 30 --warmup 5 > $out/$m.json 2> $out/$m.err; done
for m in infer_mixing train_mixing; do python3 bench.py --mode $m --dtype bf16 --steps 30 --warmup 5 > $out/${m}_bf16.json 2> $out/${m}_bf16.err; done
python3 bench.py --mode config1 > $out/config1.json 2> $out/config1.err
python3 bench.py --mode a10 > $out/a10.json 2> $out/a10.err
python3 bench.py --mode convlstm > $out/convlstm.json 2> $out/convlstm.err
bash tools/pmc_run.sh $out bench -- --steps 20 --warmup 3 --no-cpu-baseline
bash tools/prof_train_mixing.sh $out f32
bash tools/prof_train_mixing.sh $out bf16
ls -la $out
