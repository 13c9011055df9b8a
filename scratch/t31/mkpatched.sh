#!/bin/bash
# usage: mkpatched.sh NAME DEVICE_ASM.s  -> scratch/t31/var/libNAME.so : the old tree's library with conv_dma2's device code
# replaced by the given (hand-patched) assembly.  Mirrors hipcc's own sub-commands (hipcc -###).
set -e
NAME=$1; ASM=$2
T=/root/repo/scratch/t31; L=/opt/rocm/lib/llvm/bin; W=/tmp/t31/w_$NAME; mkdir -p $W $T/var
$L/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $ASM -o $W/dev.o
$L/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $W/dev.hsaco $W/dev.o
$L/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$W/dev.hsaco -output=$W/dev.hipfb
$L/clang++ -x hip --cuda-host-only -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Xclang -fcuda-include-gpubinary -Xclang $W/dev.hipfb -c $T/tree/pemp_amd/csrc/conv_dma2.hip -o $W/conv_dma2.o
objs=$(ls $T/tree/pemp_amd/_obj/*.o | grep -v conv_dma2.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $T/var/lib$NAME.so $W/conv_dma2.o $objs
echo built $T/var/lib$NAME.so
