# usage: tools/disasm.sh obj.o out.s   - gfx950 ISA of a hipcc object (the steps of fab_torch_amd/_isa_check.py)
L=/opt/rocm/lib/llvm/bin; T=$(mktemp -d)
$L/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin $1 && $L/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.elf && $L/llvm-objdump -d --no-show-raw-insn $T/dev.elf > $2
rm -rf $T
