#!/bin/bash
# A/B of the tail kernel's iteration / check cost for two builds of the library in one GPU call: tools/tail_ab.sh <a.so> <b.so>
# (file names inside the package directory; tools/tail_timing.py on each, longest and shortest path through the certificates)
P=autonomous-racing-lpv-mpp-mpc_amd
cp $P/liblpvmpc.so /tmp/liblpvmpc_keep.so
for L in "$@"; do
  cp $P/$L $P/liblpvmpc.so
  echo "== $L (longest path)"; timeout -k 10 200 python tools/tail_timing.py 2>/dev/null | grep "tail=1"
  echo "== $L (shortest path)"; EPS_INF=1e30 timeout -k 10 200 python tools/tail_timing.py 2>/dev/null | grep "tail=1"
done
cp /tmp/liblpvmpc_keep.so $P/liblpvmpc.so
