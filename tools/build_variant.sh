#!/bin/bash
# build/variants/libkpdi_<tag>.so = the library with ONE translation unit rebuilt with extra flags:
#   tools/build_variant.sh <tag> <file.hip> [-DMACRO ...]      (run tools with KPDI_LIB_PATH=build/variants/libkpdi_<tag>.so)
set -e
cd "$(dirname "$0")/../kikuchipy_amd/csrc"
tag=$1; src=$2; shift 2
mkdir -p ../../build/variants
obj=../../build/variants/${src%.hip}_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -c $src -o $obj
objs=""
for f in api plan sweep exact64 finalize extras group match match16 tailgemm prep merge preproc project refine osm h5ebsd rescore; do  # (the Makefile's OBJS)
  if [ "$f.hip" == "$src" ]; then objs="$objs $obj"; else objs="$objs $f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o ../../build/variants/libkpdi_$tag.so -ldl
echo build/variants/libkpdi_$tag.so
