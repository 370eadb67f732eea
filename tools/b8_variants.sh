#!/bin/bash
# Build timing variants of the bf16 BPTT kernel (each leaves one piece of the step out: WRONG results, never loaded by the
# package) into build/dbg/libfov_B8_<V>.so.   usage (container): tools/b8_variants.sh
set -e
cd $(dirname $0)/../longterm360fov_amd/csrc
mkdir -p ../../build/dbg ../../build/exp
for v in NOTAPE NODZ NOGATHER NOPUB; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DFOV_DBG_B8_$v -c -o ../../build/exp/lstm_bwd8_$v.o lstm_bwd8.hip &
done
wait
for v in NOTAPE NODZ NOGATHER NOPUB; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o ../../build/dbg/libfov_B8_$v.so ../../build/exp/lstm_bwd8_$v.o $(ls ../../build/obj/*.o | grep -v stamps | grep -v lstm_bwd8.o)
done
