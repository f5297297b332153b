#!/bin/bash
# usage: tools/build_variant.sh <name> [extra hipcc flags...]   -> tools/_build/libokenv_<name>.so (development aid)
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p tools/_build
/opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt --offload-arch=gfx950 -fPIC -shared \
  -fvisibility=hidden -Wno-unused-function "$@" -I include -I openkitchen_amd/csrc \
  -o tools/_build/libokenv_$NAME.so openkitchen_amd/csrc/okenv_capi.hip openkitchen_amd/csrc/facade/*.cpp
echo built tools/_build/libokenv_$NAME.so
