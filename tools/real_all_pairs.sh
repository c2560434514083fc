#!/bin/bash
# the registration loop on the reference's 19 pairs (Epoch_002 .. Epoch_020 -> Epoch_001, fixtures): median loop wall per pair and their sum
# usage: real_all_pairs.sh [runs]      (PWICP_LIB=... for a build variant)
RUNS=${1:-15}
S=0
for E in $(seq 2 20); do
  L=$(python tools/real_pair_loop.py $E $RUNS | tail -1)
  M=$(echo "$L" | sed 's/.*median \([0-9.]*\) ms.*/\1/')
  printf "%s:%s " $E $M
  S=$(python -c "print($S + $M)")
done
echo; echo "sum of the 19 medians: $S ms"
