#!/bin/bash
# Disassembly of one kernel (first whose demangled name matches the grep -E pattern) from a host object / shared library.
# usage: scripts/kernel_isa.sh <file.o|file.so> <pattern> > out.s
f=$1; pat=$2
tmp=$(mktemp -d /tmp/kisa.XXXX)
B=/opt/rocm/lib/llvm/bin
$B/llvm-objcopy --dump-section .hip_fatbin=$tmp/fat.bin $f /dev/null 2>/dev/null || $B/llvm-objcopy --dump-section .hip_fatbin=$tmp/fat.bin $f $tmp/copy
python3 - $tmp/fat.bin $tmp <<'PY'
import sys
data = open(sys.argv[1], 'rb').read()
magic = b'__CLANG_OFFLOAD_BUNDLE__'
pos, i = [], data.find(magic)
while i >= 0:
    pos.append(i); i = data.find(magic, i + 1)
for k, p in enumerate(pos):
    open('%s/fat%d.bin' % (sys.argv[2], k), 'wb').write(data[p:pos[k + 1] if k + 1 < len(pos) else len(data)])
PY
for fb in $tmp/fat[0-9]*.bin; do
  $B/clang-offload-bundler --unbundle --type=o --input=$fb --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$tmp/co 2>/dev/null || continue
  sym=$(nm $tmp/co | awk '$2 ~ /[Tt]/ {print $3}' | while read s; do echo "$s $(echo $s | c++filt)"; done | grep -E "$pat" | grep -v "\.kd" | head -1 | cut -d' ' -f1)
  [ -n "$sym" ] && $B/llvm-objdump -d --disassemble-symbols=$sym $tmp/co && break
done
rm -rf $tmp
