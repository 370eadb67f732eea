#!/bin/bash
# usage: [STEP_HAS=<kernel substring>] tools/prof_mode.sh <outdir> <tag> <bench.py arguments...>   : rocprofv3 kernel stats + last-step timeline of one bench mode
out=$1; tag=$2; shift 2
mkdir -p $GRAFT_REPO_ROOT/$out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_$tag -o tm -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > $GRAFT_REPO_ROOT/$out/${tag}_profiled.json 2> $GRAFT_REPO_ROOT/$out/${tag}.err
cd $GRAFT_REPO_ROOT
f=$(find $out/prof_$tag -name "*kernel_stats.csv" | head -1)
t=$(find $out/prof_$tag -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && cp $f $out/${tag}_kernel_stats.csv
[ -n "$t" ] && python3 tools/step_timeline.py $t adam $STEP_HAS > $out/${tag}_step_timeline.txt
rm -rf $out/prof_$tag
