#!/bin/bash
# usage: tools/pmc_sq.sh <outdir> <tag> -- <bench.py arguments>   : SQ wave-state counters of the bench kernels (two passes)
out=$1; tag=$2; shift 3
root=$GRAFT_REPO_ROOT
mkdir -p $root/$out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $root/$out/sq_${tag}_$i -o p -- python3 $root/bench.py "$@" > /dev/null 2>> $root/$out/${tag}_sq.err
done
cd $root
for j in 1 2 3 4; do python3 tools/pmc_summary.py $out/sq_${tag}_$j fov; done > $out/${tag}_sq_summary.txt
rm -rf $out/sq_${tag}_*
