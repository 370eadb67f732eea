cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_wr
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_wr -o tm -- python3 $GRAFT_REPO_ROOT/tools/wgrad_rows_probe.py > /tmp/wr_out.txt 2>&1 || { tail -5 /tmp/wr_out.txt; exit 1; }
grep variant /tmp/wr_out.txt
python3 - <<PY
import csv,glob,collections
t=glob.glob('/tmp/prof_wr/**/*kernel_trace.csv',recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(t)):
    if 'wgrad_rows' in r['Kernel_Name']:
        d[(r['Kernel_Name'][:60], r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size'))].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for k,v in d.items():
    v.sort(); print(k, 'n', len(v), 'median %.1f us min %.1f' % (v[len(v)//2]/1e3, v[0]/1e3))
PY
