#!/bin/bash
# A/B of two builds of the library in ONE gpurun call (boxes differ by several per cent): tools/ab_lib.sh <a.so> <b.so> [bench args]
# (file names inside the package directory); runs the bench alternately with each build in place of liblpvmpc.so.
A=$1; B=$2; shift 2
PKG=autonomous-racing-lpv-mpp-mpc_amd
cp $PKG/liblpvmpc.so /tmp/liblpvmpc_keep.so
for rep in 1 2 3; do
  for L in $A $B; do
    cp $PKG/$L $PKG/liblpvmpc.so
    python bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$L: value %.4f M  main kernel avg %.4f ms' % (d['value']/1e6, d['roofline']['kernel_avg_ms']))"
  done
done
cp /tmp/liblpvmpc_keep.so $PKG/liblpvmpc.so
