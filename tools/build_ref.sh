#!/bin/bash
# bash tools/build_ref.sh <git-ref> <out.so> [extra hipcc flags]: build libgpcore.so of another commit (A/B baselines)
set -e
REF=$1; OUT=$2; shift 2
T=$(mktemp -d); git archive $REF gpyreg_amd/csrc include | tar -x -C $T
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared "$@" -o $OUT $T/gpyreg_amd/csrc/gpcore.hip
rm -rf $T; ls -la $OUT
