#!/bin/bash
# csrc/dma.h (dma_stage) leaves M0 pointing into LDS instead of saving / restoring it around the LDS-DMA block.  That is safe as
# long as nothing the compiler emits in those translation units reads M0 on its own: this script compiles them to ISA and lists
# every instruction touching M0 that is not one of ours (expected output: nothing).
set -e
cd "$(dirname "$0")/../millieye_amd/csrc"
T=$(mktemp -d)
for f in conv.hip conv_h16.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -I. -S --cuda-device-only -o $T/$f.s $f
  grep -n "m0" $T/$f.s | grep -v "s_mov_b32 m0, s[0-9]*$\|s_add_u32 m0, m0, 0x[0-9a-f]*$\|s_mov_b32 s[0-9]*, m0$" | sed "s/^/$f: /" || true
done
rm -rf $T
