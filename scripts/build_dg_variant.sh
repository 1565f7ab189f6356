#!/bin/bash
# Development aid: libexahype_hip.so with only the 3-D Euler ADER-DG unit rebuilt with extra flags -> exahype_amd/lib/var_<tag>/ (select with
# EXA_LIB=...; NO_ILP=1: without the max-ilp scheduling strategy of the product build).  usage: scripts/build_dg_variant.sh <tag> <flags...>      (the main library must be built: python -m exahype_amd.build)
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
B=exahype_amd/_build; O=$B/var_$tag; L=exahype_amd/lib/var_$tag
mkdir -p $O $L
hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-pass-failed -I exahype_amd/csrc -DEXA_DIM=3 -DEXA_PDE_ID=1 -DEXA_UNIT_A ${NO_ILP:+-DEXA_NO_ILP_MARK} $( [ -z "$NO_ILP" ] && echo -mllvm -amdgpu-sched-strategy=max-ilp ) "$@" \
      -c -x hip exahype_amd/csrc/dg_inst.hip -o $O/dg_3_1.o
objs=$(ls $B/*.o | grep -v dg_3_1.o)
hipcc -shared -fPIC --offload-arch=gfx950 -o $L/libexahype_hip.so $objs $O/dg_3_1.o
echo $L/libexahype_hip.so
