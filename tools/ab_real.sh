#!/bin/bash
# the loop on the reference's Epoch_002 / Epoch_012 -> Epoch_001 under run-time switches: each argument one "ENV=VAL ..." setting ("-": defaults)
for v in "$@"; do
  [ "$v" == "-" ] && v="PWICP_NOP=1"
  echo "== $v"
  for E in 2 12; do env $v python tools/real_pair_loop.py $E 30 | tail -1; done
done
