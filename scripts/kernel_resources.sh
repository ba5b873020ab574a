#!/bin/bash
# Registers, spills, private memory and LDS of every kernel of one build of a csrc/*.hip file (cross-compiles, no GPU needed).
#   scripts/kernel_resources.sh window_solve.hip -DAVM_TP=1        (IPRA= scripts/kernel_resources.sh ... : without the round-5 flags)
set -e
cd "$(dirname "$0")/../anticipated-vins-mono_amd/csrc"
src=$1; shift
tmp=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fconstexpr-steps=16000000 --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -Wno-pass-failed \
  -mllvm -sink-insts-to-avoid-spills ${IPRA--mllvm -enable-ipra -fno-optimize-sibling-calls} "$@" --cuda-device-only -c "$src" -o "$tmp/k.bundle" 2> "$tmp/log" || { cat "$tmp/log"; exit 1; }
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$tmp/k.bundle" --output="$tmp/k.co"
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$tmp/k.co" | grep -E "\.name:|\.vgpr_count|\.agpr_count|\.sgpr_count|spill_count|private_segment_fixed|group_segment_fixed" | sed 's/^ *//'
[ -n "$KEEP" ] && cp "$tmp/k.co" "$KEEP"
rm -rf "$tmp"
