out=gpurun_out/r04_conv; mkdir -p $out; root=$GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_gpu_convlstm.py -q -x > $out/test.log 2>&1; tail -2 $out/test.log
timeout -k 10 300 python bench.py --mode convlstm --no-cpu-baseline > $out/convlstm.json 2> $out/convlstm.err; python -c "
import json; d=json.load(open('$out/convlstm.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], {k:v for k,v in d.items() if k in ('cell_only','training','whole_model')})"
: > $out/pmcstep_convlstm.txt
for ctr in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d $root/$out/pmcs_$ctr -o p -- python3 $root/tools/pmc_simple_steps.py convlstm 2 > /dev/null 2>> $root/$out/pmc.err)
  python3 tools/pmc_run_total.py $out/pmcs_$ctr 2 >> $out/pmcstep_convlstm.txt
  rm -rf $out/pmcs_$ctr
done
cat $out/pmcstep_convlstm.txt
