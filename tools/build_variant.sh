#!/bin/bash
# builds ab/libpopcorn_<tag>.so = the library with extra compiler flags on every source (the per-file flags of the Makefile kept):
#   [SKIP="a.hip b.hip"] tools/build_variant.sh <tag> <extra flags...>        then: gpurun -- 'bash tools/ab_lib.sh ab/libpopcorn_<tag>.so'
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
B=/tmp/variant_$TAG; mkdir -p $B $ROOT/ab
cd $ROOT/popcorn_amd/csrc || exit 1
for f in *.hip; do
  extra=""
  case $f in conv3x3.hip|conv3x3_bwd.hip|up_bwd.hip|convt2x2.hip) extra="-mllvm -amdgpu-mfma-vgpr-form=1";; esac
  more=("$@")
  case " $SKIP " in *" $f "*) more=();; esac          # SKIP="head.hip level2.hip": those files keep the Makefile's flags
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$ROOT/include $extra "${more[@]}" -c $f -o $B/${f%.hip}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/ab/libpopcorn_$TAG.so $B/*.o && ls -la $ROOT/ab/libpopcorn_$TAG.so
