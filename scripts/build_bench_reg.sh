#!/bin/bash
# Development aid: build scripts/bin/bench_reg[_TAG] (and its stamp build) with extra -D flags.  usage: scripts/build_bench_reg.sh [TAG [flags...]]
TAG=${1:+_$1}; shift
F="-O3 -std=c++17 --offload-arch=gfx950 -Wno-pass-failed -Wno-unused-variable -I exahype_amd/csrc $@"
mkdir -p scripts/bin
hipcc $F scripts/bench_reg.hip -o scripts/bin/bench_reg$TAG 2>&1 | grep -E "error|Error" 
hipcc $F -DEXA_STAMPS scripts/bench_reg.hip -o scripts/bin/bench_reg${TAG}_stamps 2>&1 | grep -E "error|Error"
ls -la scripts/bin/bench_reg$TAG scripts/bin/bench_reg${TAG}_stamps | awk '{print $5, $9}'
