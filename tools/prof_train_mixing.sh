#!/bin/bash
# usage: tools/prof_train_mixing.sh <outdir> <dtype>   (on the GPU box, from the repo root)
# rocprofv3 kernel trace of bench.py --mode train_mixing; writes <outdir>/<dtype>_kernel_stats.csv and the per-launch
# timeline of the last optimizer step.
out=$1; dt=$2; mkdir -p $GRAFT_REPO_ROOT/$out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_$dt -o tm -- python3 $GRAFT_REPO_ROOT/bench.py --mode train_mixing --dtype $dt --steps 10 --warmup 3 > $GRAFT_REPO_ROOT/$out/prof_${dt}.json 2> $GRAFT_REPO_ROOT/$out/prof_${dt}.err
cd $GRAFT_REPO_ROOT
f=$(find $out/prof_$dt -name "*kernel_stats.csv" | head -1)
t=$(find $out/prof_$dt -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && cp $f $out/${dt}_kernel_stats.csv
[ -n "$t" ] && python3 tools/step_timeline.py $t adam > $out/${dt}_step_timeline.txt
rm -rf $out/prof_$dt
