#!/bin/bash
# VGPRs / SGPRs / scratch / LDS of the kernels in a host object or shared library (code-object metadata).
# usage: scripts/kernel_resources.sh <file.o|file.so> [name filter (grep -E)]
f=$1; pat=${2:-.}
tmp=$(mktemp -d /tmp/kres.XXXX)
B=/opt/rocm/lib/llvm/bin
$B/llvm-objcopy --dump-section .hip_fatbin=$tmp/fat.bin $f /dev/null 2>/dev/null || $B/llvm-objcopy --dump-section .hip_fatbin=$tmp/fat.bin $f $tmp/copy
python3 - $tmp/fat.bin $tmp <<'PY'
import sys
# a .so holds one fatbin per translation unit, each padded to 4 KB: split at the bundle magic
data = open(sys.argv[1], 'rb').read()
magic = b'__CLANG_OFFLOAD_BUNDLE__'
pos = [i for i in range(len(data)) if data.startswith(magic, i)] if len(data) < (1 << 28) else []
for k, p in enumerate(pos):
    open('%s/fat%d.bin' % (sys.argv[2], k), 'wb').write(data[p:pos[k + 1] if k + 1 < len(pos) else len(data)])
PY
for fb in $tmp/fat[0-9]*.bin; do
  $B/clang-offload-bundler --unbundle --type=o --input=$fb --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$tmp/co 2>/dev/null || continue
  $B/llvm-readelf --notes $tmp/co | awk '
    /\.name:/ {name=$2} /\.vgpr_count:/ {v=$2} /\.sgpr_count:/ {s=$2} /\.private_segment_fixed_size:/ {p=$2} /\.group_segment_fixed_size:/ {g=$2}
    /\.vgpr_spill_count:/ {sp=$2} /\.wavefront_size:/ {print name, "vgpr", v, "sgpr", s, "scratch", p, "spill", sp, "lds", g}' | while read n rest; do echo "$(echo $n | c++filt | sed -e "s/(.*//") $rest"; done
done | grep -E "$pat" || true
rm -rf $tmp
