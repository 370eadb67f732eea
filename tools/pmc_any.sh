#!/bin/bash
# usage: tools/pmc_any.sh <outdir> <tag> "<counters>" <script.py> [args]   : one rocprofv3 --pmc pass over an arbitrary script, per-kernel means
out=$1; tag=$2; ctrs=$3; shift 3
root=$GRAFT_REPO_ROOT
mkdir -p $root/$out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctrs --output-format csv -d $root/$out/any_$tag -o p -- python3 $root/"$@" > /dev/null 2>> $root/$out/${tag}_any.err
cd $root
python3 tools/pmc_summary.py $out/any_$tag fov > $out/${tag}_any_summary.txt
rm -rf $out/any_$tag
