#!/bin/bash
# Development aid: libexahype_hip.so with only the FV unit rebuilt with extra flags -> exahype_amd/lib/var_<tag>/ (select with EXA_LIB=...).
# usage: scripts/build_fv_variant.sh <tag> <flags...>      (the main library must be built: python -m exahype_amd.build)
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
B=exahype_amd/_build; O=$B/var_$tag; L=exahype_amd/lib/var_$tag
mkdir -p $O $L
hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-pass-failed -I exahype_amd/csrc -ffp-contract=off "$@" \
      -c exahype_amd/csrc/fv_rusanov.hip -o $O/fv_rusanov.o
objs=$(ls $B/*.o | grep -v fv_rusanov.o)
hipcc -shared -fPIC --offload-arch=gfx950 -o $L/libexahype_hip.so $objs $O/fv_rusanov.o
echo $L/libexahype_hip.so
