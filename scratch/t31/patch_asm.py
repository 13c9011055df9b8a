"""patch_asm.py IN.s OUT.s SPEC...   SPEC = first:last:old=new[,old=new...]   (line numbers relative to the kernel's label line = 1)
Renames whole-token scalar registers inside a line range of the tile-31 kernel and raises its .amdhsa_next_free_sgpr."""
import re, sys
SYM = "_ZN4pemp16conv_dma2_kernelILi128ELi128ELi2ELi4ELb1ELi0ELb1ELb0EEEvNS_8ConvArgsE"
src = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(src) if l.startswith(SYM + ":"))
for spec in sys.argv[3:]:
    a, b, ren = spec.split(":")
    a, b = int(a), int(b)
    pairs = [p.split("=") for p in ren.split(",")]
    n = 0
    for i in range(start + a - 1, start + b):
        l = src[i]
        for old, new in pairs:
            l2 = re.sub(r"(?<![\w\[:])" + re.escape(old) + r"(?![\w:\]])", "@@" + new, l)
            if l2 != l:
                n += 1
            l = l2
        src[i] = l.replace("@@", "")
    print(spec, "->", n, "replacements")
# descriptor
k = next(i for i, l in enumerate(src) if l.strip() == ".amdhsa_kernel " + SYM)
for i in range(k, k + 60):
    if ".amdhsa_next_free_sgpr" in src[i]:
        src[i] = "\t\t.amdhsa_next_free_sgpr 74"
        break
open(sys.argv[2], "w").write("\n".join(src))
