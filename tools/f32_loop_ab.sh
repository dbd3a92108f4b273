#!/bin/bash
# Where the f32 wide kernel's tile time goes: timing-only ablation builds (tools/build_variant.sh <tag> match16.hip -D...)
# through tools/tile_ramp_probe.py wide (match ms against whole tiles per workgroup: slope = ms per tile).
#   bash tools/f32_loop_ab.sh <out.txt> shipped noepi nodma_noepi ...     (run on the GPU box)
cd "$(dirname "$0")/.." && R=$PWD
out=$1; shift
: > $out
for tag in "$@"; do
  lib=$R/build/variants/libkpdi_$tag.so
  [ "$tag" == shipped ] && lib=$R/kikuchipy_amd/csrc/libkpdi.so
  echo "== $tag" >> $out
  KPDI_LIB_PATH=$lib timeout 300 python tools/tile_ramp_probe.py wide 2>&1 | tail -3 >> $out
done
cat $out
