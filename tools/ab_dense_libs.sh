#!/bin/bash
# the dense 1-NN launch under library build variants (tools/build_variant.sh): ab_dense_libs.sh name [name ...]   ("base" = the regular build)
# DV_POINTS=4000000 for the 4 M-point pair; prints dense_variants' line per variant
for v in "$@"; do
  L=""; [ "$v" != "base" ] && L=$GRAFT_REPO_ROOT/piecewise-icp_amd/variants/libpwicp_$v.so
  printf "%-14s" $v; env PWICP_LIB=$L python tools/dense_variants.py "-" 2>&1 | tail -1
done
