#!/bin/bash
# Builds the library with extra compiler flags into raytracer-rust_amd/csrc_ab/<name>/librtk.so (for tools/ab_many.sh):
#   tools/build_variant.sh <name> [flags...]        e.g.  tools/build_variant.sh phases1 -DRTK_PHASES=1
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d /tmp/rtk_variant_XXXX)
mkdir -p $TMP/raytracer-rust_amd $TMP/include
cp -rp $ROOT/raytracer-rust_amd/csrc $TMP/raytracer-rust_amd/csrc
cp $ROOT/include/rtk.h $TMP/include/
# ONLY=<file stem> rebuilds that translation unit alone (the flags only matter to it), else everything
if [ -n "${ONLY:-}" ]; then rm -f $TMP/raytracer-rust_amd/csrc/$ONLY.o $TMP/raytracer-rust_amd/csrc/librtk.so; else rm -f $TMP/raytracer-rust_amd/csrc/*.o $TMP/raytracer-rust_amd/csrc/*.so; fi
make -s -C $TMP/raytracer-rust_amd/csrc HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $*" librtk.so > /dev/null
mkdir -p $ROOT/raytracer-rust_amd/csrc_ab/$NAME
cp $TMP/raytracer-rust_amd/csrc/librtk.so $ROOT/raytracer-rust_amd/csrc_ab/$NAME/librtk.so
cp $TMP/raytracer-rust_amd/csrc/*_resources.txt $ROOT/raytracer-rust_amd/csrc_ab/$NAME/ 2>/dev/null || true
rm -rf $TMP
echo built $NAME
