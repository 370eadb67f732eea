# per-kernel totals of the ConvLSTM training step (rocprofv3 --kernel-trace --stats over tools/convlstm_train_step.py: 3 steps)
out=gpurun_out/r05_conv; mkdir -p $out; root=$GRAFT_REPO_ROOT
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/kt -o p -- python3 $root/tools/convlstm_train_step.py 2 > $root/$out/train_step.txt 2>> $root/$out/kt.err)
cat $out/train_step.txt
f=$(find $out/kt -name "*kernel_stats.csv" | head -1); cp $f $out/convlstm_train_kernel_stats.csv; rm -rf $out/kt
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r05_conv/convlstm_train_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:16]:
    print("%-78s calls %5s total %9.2f ms  avg %9.1f us  %5.1f%%" % (r['Name'][:78], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot))
print("sum of kernels: %.1f ms over 3 steps" % (tot/1e6))
PY
