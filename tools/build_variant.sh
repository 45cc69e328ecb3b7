#!/bin/bash
# Build an A/B variant of the library: tools/build_variant.sh <tag> <source.hip> [-DNAME=VALUE ...]
# compiles ONE source with extra flags into hirest_amd/lib/<source>.<tag>.o and links it with the other (default) objects into
# hirest_amd/lib/libhirest_hip.<tag>.so; select it with HIREST_LIB_VARIANT=<tag> (hirest_amd/_lib.py).  The variants travel to the GPU box
# with the snapshot (built files are git-ignored, not gpurun-ignored), so several builds are timed inside ONE gpurun call — the boxes of the pool
# differ by +-3 %, more than most of what is being measured.
set -e
tag=$1; src=$2; shift 2
cd "$(dirname "$0")/.."
L=hirest_amd/lib
python -m hirest_amd.build > /dev/null
extra=""
case $src in attention.hip) extra="-ffinite-math-only";; preprocess.hip|eval.hip) extra="-ffp-contract=off";; esac
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast $extra "$@" -c hirest_amd/csrc/$src -o $L/${src%.hip}.$tag.o
objs=""
for o in $L/*.o; do
  b=$(basename $o .o)
  case $b in *.*) continue;; esac
  if [ "$b" = "${src%.hip}" ]; then objs="$objs $L/${src%.hip}.$tag.o"; else objs="$objs $o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libhirest_hip.$tag.so $objs
echo $L/libhirest_hip.$tag.so
