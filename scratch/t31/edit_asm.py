"""edit_asm.py IN.s OUT.s [--dump v72=..] SPEC...   SPEC = LINE:+text|text  (insert AFTER kernel-relative line; '|' separates lines)
or LINE:-text (insert BEFORE).  --dump REG,REG,...: wave 0 of block 0 stores these (s or v registers, up to 6) when the main loop ends
(behind the activation tensor, as instrument_asm.py)."""
import sys
SYM = "_ZN4pemp16conv_dma2_kernelILi128ELi128ELi2ELi4ELb1ELi0ELb1ELb0EEEvNS_8ConvArgsE"
src = open(sys.argv[1]).read().split("\n")
s0 = next(i for i, l in enumerate(src) if l.startswith(SYM + ":"))
L = lambda n: s0 + n - 1
before, after = {}, {}
args = sys.argv[3:]
dump = None
if args and args[0] == "--dump":
    dump = args[1].split(",")
    args = args[2:]
for spec in args:
    n, rest = spec.split(":", 1)
    tgt = after if rest[0] == "+" else before
    tgt.setdefault(L(int(n)), []).extend("\t" + t for t in rest[1:].split("|"))
if dump is not None:
    assert "s_cbranch_scc1 .LBB46_52" in src[L(1378)]
    before.setdefault(L(3), []).extend(["\ts_mov_b64 s[90:91], s[0:1]", "\ts_mov_b32 s92, s2"])
    code = ["\ts_cmp_lg_u32 s92, 0", "\ts_cbranch_scc1 .Ldbg_skip", "\ts_cmp_lg_u32 s58, 0", "\ts_cbranch_scc1 .Ldbg_skip",
            "\ts_load_dwordx2 s[94:95], s[90:91], 0x0", "\ts_waitcnt lgkmcnt(0)",
            "\tv_mbcnt_lo_u32_b32 v78, -1, 0", "\tv_mbcnt_hi_u32_b32 v78, -1, v78", "\tv_lshlrev_b32_e32 v78, 2, v78",
            f"\tv_add_u32_e32 v78, {(5202 + 1) * 1024}, v78"]
    for k, r in enumerate(dump):
        code += [f"\tv_mov_b32_e32 v80, {r}", f"\tglobal_store_dword v78, v80, s[94:95] offset:{256 * k}"]
    code += ["\ts_waitcnt vmcnt(0)", ".Ldbg_skip:"]
    before.setdefault(L(1379), []).extend(code)
out = []
for i, l in enumerate(src):
    out.extend(before.get(i, []))
    out.append(l)
    out.extend(after.get(i, []))
txt = "\n".join(out)
k = txt.index(".amdhsa_kernel " + SYM)
j = txt.index(".amdhsa_next_free_sgpr", k)
e = txt.index("\n", j)
txt = txt[:j] + ".amdhsa_next_free_sgpr 96" + txt[e:]
open(sys.argv[2], "w").write(txt)
