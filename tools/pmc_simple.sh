#!/bin/bash
# usage (GPU box, repo root): tools/pmc_simple.sh <outdir>   - FETCH_SIZE / WRITE_SIZE per step of the config1, a10 and convlstm workloads
out=$1; mkdir -p $out; root=$GRAFT_REPO_ROOT
for what in config1 a10 convlstm; do
  steps=$( [ $what = convlstm ] && echo 2 || echo 20 )
  : > $out/pmcstep_$what.txt
  for ctr in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc $ctr --output-format csv -d $root/$out/pmcs_${what}_$ctr -o p -- python3 $root/tools/pmc_simple_steps.py $what $steps > /dev/null 2>> $root/$out/pmc_simple.err)
    python3 tools/pmc_run_total.py $out/pmcs_${what}_$ctr $steps >> $out/pmcstep_$what.txt
    rm -rf $out/pmcs_${what}_$ctr
    echo "[pmc_simple] $what $ctr done" >> $out/progress.log
  done
done
