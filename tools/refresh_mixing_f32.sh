out=gpurun_out/final11; root=$GRAFT_REPO_ROOT
for m in infer_mixing train_mixing; do python3 bench.py --mode $m --steps 200 --warmup 10 --cpu-budget 8 > $out/$m.json 2> $out/$m.err; done
bash tools/prof_train_mixing.sh $out f32
for spec in "train_mixing f32 adam_kernel" "infer_mixing f32 mix_decoder_kernel"; do
  set -- $spec
  : > $out/pmcstep_$1_$2.txt
  for ctr in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do
    (cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc $ctr --output-format csv -d $root/$out/pmcstep_$1_$2_$ctr -o p -- python3 $root/bench.py --mode $1 --dtype $2 --steps 12 --warmup 3 --no-cpu-baseline > /dev/null 2>> $root/$out/pmcstep.err)
    python3 tools/pmc_step_total.py $out/pmcstep_$1_$2_$ctr $3 fov >> $out/pmcstep_$1_$2.txt
    rm -rf $out/pmcstep_$1_$2_$ctr
  done
done
for m in infer_mixing train_mixing; do python3 -c "
import json; d=json.loads(open('$out/$m.json').read().strip().splitlines()[-1]); print('$m', d['ms_per_step'], d['value'])"; done
grep -E "^(FETCH|WRITE)" $out/pmcstep_train_mixing_f32.txt $out/pmcstep_infer_mixing_f32.txt
