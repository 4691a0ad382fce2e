#!/bin/bash
# build a named variant of the library for A/B timing: tools/build_variant.sh <name> [extra hipcc flags]
name=$1; shift
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Iinclude -Imultigrid_amd/csrc \
  "$@" -o tools/libmgrid_$name.so multigrid_amd/csrc/mgrid.hip
