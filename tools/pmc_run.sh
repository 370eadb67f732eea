#!/bin/bash
# usage: tools/pmc_run.sh <outdir> <tag> -- <bench.py arguments>       (on the GPU box, from the repo root)
# Three separate rocprofv3 --pmc passes (SQ busy cycles / FETCH_SIZE / WRITE_SIZE: the TCC counters do not fit one pass,
# MI355X_MICROARCH.md "rocprofv3 PMC slots") plus a --kernel-trace --stats pass of the SAME command; per-kernel means
# go to <outdir>/<tag>_pmc_summary.txt and <outdir>/<tag>_kernel_stats.csv.
out=$1; tag=$2; shift 3
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for ctr in SQ_VALU_MFMA_BUSY_CYCLES FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --output-format csv -d $root/$out/pmc_${tag}_$ctr -o p -- python3 $root/bench.py "$@" > /dev/null 2>> $root/$out/${tag}_pmc.err
done
rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/pmc_${tag}_trace -o p -- python3 $root/bench.py "$@" > $root/$out/${tag}_profiled.json 2>> $root/$out/${tag}_pmc.err
cd $root
{
  echo "# rocprofv3 --pmc (one counter per pass) over: python3 bench.py $*"
  echo "# mean per dispatch; SQ_VALU_MFMA_BUSY_CYCLES is summed over all SIMDs; FETCH_SIZE / WRITE_SIZE in KiB (raw)"
  for ctr in SQ_VALU_MFMA_BUSY_CYCLES FETCH_SIZE WRITE_SIZE; do python3 tools/pmc_summary.py $out/pmc_${tag}_$ctr fov; done
  echo "# mean duration per kernel (us), rocprofv3 --kernel-trace of the same command"
  t=$(find $out/pmc_${tag}_trace -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python3 tools/kernel_means.py $t fov
} > $out/${tag}_pmc_summary.txt
f=$(find $out/pmc_${tag}_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/${tag}_kernel_stats.csv
rm -rf $out/pmc_${tag}_*
