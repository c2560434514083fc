#!/bin/bash
# Knock-out profiling of k_fus_run: builds that end a centre's run early (-DPW_FUS_KO=1: after the prologue, 2: after the search has
# taken in the centre's own list, 3: before the outcome is compared / written, 0: the whole run) time the kernel of the FIRST sweep of
# round 0 with events (the same 10^6 centres in every build, every centre against the untouched state of the round's start) and then
# hand the cloud to the host passes - what each part of a run costs.
# build: for k in 0 1 2 3; do tools/build_variant.sh ko$k "-DPW_FUS_KO=$k" frontend; done ; run on the GPU box: bash tools/fus_knockout.sh
for v in ko1 ko2 ko3 ko0; do
  L=$GRAFT_REPO_ROOT/piecewise-icp_amd/variants/libpwicp_$v.so
  echo "== $v"; PWICP_LIB=$L timeout 300 python bench.py --workload frontend --steps 1 --no-cpu-baseline 2>&1 | grep -E 'knock-out' | tail -2
done
