#!/bin/bash
# A/B of compile-time variants: rebuilds the named translation units with extra flags and links them with the objects of the
# regular build into piecewise-icp_amd/variants/libpwicp_<NAME>.so (git-ignored, travels with gpurun); run with PWICP_LIB=<that file>.
# usage: tools/build_variant.sh NAME "FLAGS" unit [unit ...]      e.g.  tools/build_variant.sh deep8 "-DPW_SCAN_DEEP=8" grid
set -e
NAME=$1; FLAGS=$2; shift 2
P=$(cd $(dirname $0)/../piecewise-icp_amd && pwd)
mkdir -p $P/variants/$NAME
OBJS=""
for o in $P/build/*.o; do
  b=$(basename $o .o); use=$o
  for u in "$@"; do
    if [ "$b" == "$u" ]; then
      hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -I$P/../include -I$P/csrc -Wall -Wno-unused-function $FLAGS -c $P/csrc/$u.hip -o $P/variants/$NAME/$u.o
      use=$P/variants/$NAME/$u.o
    fi
  done
  OBJS="$OBJS $use"
done
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -pthread -ldl -o $P/variants/libpwicp_$NAME.so
echo built $P/variants/libpwicp_$NAME.so
