#!/bin/bash
# usage (GPU box, repo root): tools/final_round.sh <outdir> [bench|pmc|all]   - the measurements a round's profiles/ are
# refreshed from; the two halves fit one 20-minute gpurun call each
out=$1; part=${2:-all}; mkdir -p $out
# progress goes to $out/progress.log (gpurun takes a command that writes nothing for 7 minutes to be hung)
exec >> $out/progress.log 2>&1
if [ $part != pmc ]; then
python3 bench.py > $out/bench.json 2> $out/bench.err
echo "[final_round] bench done"
for m in train infer_mixing train_mixing; do python3 bench.py --mode $m --steps 200 --warmup 10 --cpu-budget 8 > $out/$m.json 2> $out/$m.err; done
for m in infer_mixing train_mixing; do python3 bench.py --mode $m --dtype bf16 --steps 200 --warmup 10 --cpu-budget 8 > $out/${m}_bf16.json 2> $out/${m}_bf16.err; done
# the mixing model at the metric's horizon (config.py:20-23: running_length / predict_step = 30)
for m in infer_mixing train_mixing; do
  python3 bench.py --mode $m --t-in 30 --t-out 30 --steps 100 --warmup 5 --cpu-budget 6 > $out/${m}_t30.json 2> $out/${m}_t30.err
  python3 bench.py --mode $m --dtype bf16 --t-in 30 --t-out 30 --steps 100 --warmup 5 --cpu-budget 6 > $out/${m}_bf16_t30.json 2> $out/${m}_bf16_t30.err
done
echo "[final_round] mixing modes done"
python3 bench.py --mode config1 --train > $out/config1.json 2> $out/config1.err
python3 bench.py --mode a10 > $out/a10.json 2> $out/a10.err
echo "[final_round] config1, a10 done"
python3 bench.py --mode convlstm > $out/convlstm.json 2> $out/convlstm.err
echo "[final_round] convlstm done"
fi
if [ $part != bench ]; then
bash tools/pmc_run.sh $out bench -- --steps 20 --warmup 3 --no-cpu-baseline
bash tools/pmc_sq.sh $out bench -- --steps 20 --warmup 3 --no-cpu-baseline
echo "[final_round] headline PMC done"
bash tools/prof_train_mixing.sh $out f32
bash tools/prof_train_mixing.sh $out bf16
# lstm.py's training step at its native shape: per-launch timeline of the last step
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_a10 -o tm -- python3 $GRAFT_REPO_ROOT/tools/a10_train_step.py --steps 20 > $GRAFT_REPO_ROOT/$out/a10_train_profiled.txt 2> $GRAFT_REPO_ROOT/$out/a10_train.err)
t=$(find $out/prof_a10 -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python3 tools/step_timeline.py $t rmsprop > $out/a10_train_step_timeline.txt
rm -rf $out/prof_a10
python3 tools/a10_train_step.py --steps 300 > $out/a10_train_step.txt 2>> $out/a10_train.err
echo "[final_round] a10 training timeline done"
# HBM-side bytes per step of the multi-launch modes (FETCH_SIZE / WRITE_SIZE in separate passes)
root=$GRAFT_REPO_ROOT
for spec in "train_mixing f32 adam_kernel" "train_mixing bf16 adam_kernel" "infer_mixing f32 mix_decoder_kernel" "infer_mixing bf16 mix_decoder_bf16_kernel" "train f32 adam_kernel"; do
  set -- $spec
  for ctr in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do
    (cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc $ctr --output-format csv -d $root/$out/pmcstep_$1_$2_$ctr -o p -- python3 $root/bench.py --mode $1 --dtype $2 --steps 12 --warmup 3 --no-cpu-baseline > /dev/null 2>> $root/$out/pmcstep.err)
    python3 tools/pmc_step_total.py $out/pmcstep_$1_$2_$ctr $3 fov >> $out/pmcstep_$1_$2.txt
    rm -rf $out/pmcstep_$1_$2_$ctr
  done
  echo "[final_round] pmc $1 $2 done"
done
fi
ls -la $out
