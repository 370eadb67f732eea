out=gpurun_out/r04_conv; mkdir -p $out; root=$GRAFT_REPO_ROOT
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/kt -o p -- python3 $root/tools/pmc_simple_steps.py convlstm 2 > /dev/null 2>> $root/$out/kt.err)
f=$(find $out/kt -name "*kernel_stats.csv" | head -1); cp $f $out/convlstm_predict_kernel_stats.csv; rm -rf $out/kt
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r04_conv/convlstm_predict_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:14]:
    print("%-70s calls %5s total %9.2f ms  avg %9.1f us  %5.1f%%" % (r['Name'][:70], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot))
print("sum of kernels: %.1f ms over 2 predicts" % (tot/1e6))
PY
