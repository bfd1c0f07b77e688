#!/bin/bash
# Build another copy of the library with extra compiler flags (measurement variants, A/B runs): tools/_ab/libsml_hip_<name>.so
# usage: tools/build_variant.sh <name> [flags...]       e.g. tools/build_variant.sh noreplay -DSML_DBG_NOREPLAY
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/tools/_ab; TMP=$(mktemp -d)
mkdir -p $OUT
for s in transfer_net mf_kernels index_prep capi; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c $ROOT/sml_amd/csrc/$s.hip -o $TMP/$s.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libsml_hip_$NAME.so $TMP/transfer_net.o $TMP/mf_kernels.o $TMP/index_prep.o $TMP/capi.o
rm -rf $TMP
echo $OUT/libsml_hip_$NAME.so
