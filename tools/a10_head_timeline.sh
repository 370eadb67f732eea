# kernel timeline of lstm.py's training step under each head (rocprofv3 --kernel-trace, last optimizer step)
# usage: bash tools/a10_head_timeline.sh <outdir under gpurun_out> [heads...]
out=gpurun_out/${1:-a10_heads}; shift; mkdir -p $out; root=$GRAFT_REPO_ROOT
for h in ${@:-gmm raw meanvar}; do
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $root/$out/prof_$h -o tm -- python3 $root/tools/a10_train_step.py --steps 20 --head $h > $root/$out/train_profiled_$h.txt 2> $root/$out/train_$h.err)
  t=$(find $out/prof_$h -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python3 tools/step_timeline.py $t rmsprop > $out/a10_${h}_train_step_timeline.txt
  rm -rf $out/prof_$h
  python3 tools/a10_train_step.py --steps 300 --head $h >> $out/a10_train_step.txt
done
cat $out/a10_train_step.txt
