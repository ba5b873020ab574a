#!/bin/bash
# Builds build/variants/libavm_hip_<name>.so: the shipped objects with the three window_solve builds recompiled with extra flags
# (compiler-flag / macro experiments; scripts/dev_variants.sh times them on the GPU box).
#   scripts/build_variant.sh ipra "-mllvm -enable-ipra -fno-optimize-sibling-calls" [tp|x|base ...]
set -e
cd "$(dirname "$0")/../anticipated-vins-mono_amd/csrc"
name=$1; extra=$2; shift 2
which=${@:-base x tp}
out=../../build/variants; mkdir -p $out/$name
make -s -j4 >/dev/null
FLAGS="-O3 -std=c++17 -fconstexpr-steps=16000000 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable -Wno-pass-failed -mllvm -sink-insts-to-avoid-spills"
IPRA=${IPRA--mllvm -enable-ipra -fno-optimize-sibling-calls}   # (as csrc/Makefile: base and tp builds; IPRA= switches it off)
cp window_solve.o window_solve_x.o window_solve_tp.o $out/$name/
for w in $which; do
  case $w in
    base) /opt/rocm/bin/hipcc $FLAGS $IPRA $extra -c window_solve.hip -o $out/$name/window_solve.o & ;;
    x) /opt/rocm/bin/hipcc $FLAGS -mllvm -amdgpu-prealloc-sgpr-spill-vgprs $extra -DAVM_X=1 -c window_solve.hip -o $out/$name/window_solve_x.o & ;;
    tp) /opt/rocm/bin/hipcc $FLAGS $IPRA $extra -DAVM_TP=1 -c window_solve.hip -o $out/$name/window_solve_tp.o & ;;
  esac
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libavm_hip_$name.so avm_api.o preint.o prior_eig.o fsel.o triangulate.o adapters.o \
  $out/$name/window_solve.o $out/$name/window_solve_x.o $out/$name/window_solve_tp.o -ldl
echo built $out/libavm_hip_$name.so
