#!/bin/bash
# Build timing variants of the three-role width-512 launch (each leaves one wait or product out: WRONG results, never loaded by
# the package) into build/dbg/libfov_W16_<V>.so.   usage (container): tools/w16_variants.sh ; on the GPU box:
#   FOV_LIB_PATH=build/dbg/libfov_W16_NOZWAIT.so python tools/a10_prologue_probe.py
set -e
cd $(dirname $0)/../longterm360fov_amd/csrc
mkdir -p ../../build/dbg ../../build/exp
for v in NOZWAIT NOPRODMM NOPRODGATHER NOL2GATHER NOL1GATHER; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DFOV_DBG_W16_$v -c -o ../../build/exp/lstm_wide16_$v.o lstm_wide16.hip &
done
wait
for v in NOZWAIT NOPRODMM NOPRODGATHER NOL2GATHER NOL1GATHER; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o ../../build/dbg/libfov_W16_$v.so ../../build/exp/lstm_wide16_$v.o $(ls ../../build/obj/*.o | grep -v stamps | grep -v lstm_wide16.o)
done
