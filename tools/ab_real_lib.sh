#!/bin/bash
# the loop on the reference's Epoch_002 / 012 / 016 -> Epoch_001 under library build variants: ab_real_lib.sh name [name ...]   ("base")
for v in "$@"; do
  L=""; [ "$v" != "base" ] && L=$GRAFT_REPO_ROOT/piecewise-icp_amd/variants/libpwicp_$v.so
  echo "== $v"
  for E in 2 12 16; do env PWICP_LIB=$L python tools/real_pair_loop.py $E 40 | tail -1; done
done
