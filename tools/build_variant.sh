#!/bin/bash
# Build an A/B variant of liblkhip.so: tools/build_variant.sh <name> <file.hip> "<extra hipcc flags>"  ->  build/ab/<name>.so
# (<file.hip> is recompiled with the extra flags, every other object comes from the current in-tree build; select a variant with
# LK_LIB_PATH=build/ab/<name>.so — tools/ab_*.sh interleave several on one box)
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; extra=$3
C=lightkurve_amd/csrc
make -s -C $C
mkdir -p build/ab build/obj
base=$(basename $src .hip)
fp=""; case $base in bls|pgsmooth) fp="-ffp-contract=off";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result $fp $extra -c $C/$base.hip -o build/obj/$name.$base.o
objs=""; for o in capi ls lsfast bls regress flatten pld pgsmooth fold ingest device; do
  if [ $o = $base ]; then objs="$objs build/obj/$name.$base.o"; else objs="$objs $C/$o.o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/ab/$name.so $objs
echo build/ab/$name.so
