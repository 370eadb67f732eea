out=gpurun_out/final10; mkdir -p $out; root=$GRAFT_REPO_ROOT
python3 bench.py --mode a10 > $out/a10_v3.json 2> $out/a10_v3.err
: > $out/pmcstep_a10.txt
for ctr in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc $ctr --output-format csv -d $root/$out/pmcs_a10_$ctr -o p -- python3 $root/tools/pmc_simple_steps.py a10 20 > /dev/null 2>> $root/$out/pmc_simple.err)
  python3 tools/pmc_run_total.py $out/pmcs_a10_$ctr 20 >> $out/pmcstep_a10.txt
  rm -rf $out/pmcs_a10_$ctr
done
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $root/$out/prof_a10 -o tm -- python3 $root/tools/a10_train_step.py --steps 20 > $root/$out/a10_train_profiled.txt 2> $root/$out/a10_train.err)
t=$(find $out/prof_a10 -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python3 tools/step_timeline.py $t rmsprop > $out/a10_train_step_timeline.txt
rm -rf $out/prof_a10
python3 tools/a10_train_step.py --steps 300 > $out/a10_train_step.txt
cat $out/a10_train_step.txt; grep -E "^(FETCH|WRITE)" $out/pmcstep_a10.txt; cat $out/a10_train_step_timeline.txt
