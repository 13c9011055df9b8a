"""instrument_asm.py IN.s OUT.s DBG_BLOCK : ISA-level trace of the tile-31 kernel's scalar K-step state (no re-compilation, so the
register allocation and schedule of the failing build stay exactly as they are): every main-loop iteration records
tap / kw / kh / cb / sa_ / sb_ (s28, s56, s37, s18, s59, s41) into lane <loop counter> of v72..v77; wave 0 of block DBG_BLOCK stores the six
registers behind the activation tensor (x + (5202 + 1) * 1024 bytes) when the loop ends."""
import sys
SYM = "_ZN4pemp16conv_dma2_kernelILi128ELi128ELi2ELi4ELb1ELi0ELb1ELb0EEEvNS_8ConvArgsE"
src = open(sys.argv[1]).read().split("\n")
dbg = int(sys.argv[3])
s0 = next(i for i, l in enumerate(src) if l.startswith(SYM + ":"))
L = lambda n: s0 + n - 1          # kernel-relative line (1 = label) -> index
assert "buffer_load_dwordx4 v56, s[0:3], s59 offen lds" in src[L(1314)]
assert "buffer_load_dwordx4 v13, s[4:7], s41 offen lds" in src[L(1348)]
assert "s_cbranch_scc1 .LBB46_52" in src[L(1378)]
ins = {}
ins[L(3)] = ["\ts_mov_b64 s[74:75], s[0:1]", "\ts_mov_b32 s78, s2", "\tv_mbcnt_lo_u32_b32 v79, -1, 0", "\tv_mbcnt_hi_u32_b32 v79, -1, v79"] + \
    [f"\tv_mov_b32_e32 v{r}, -1" for r in range(72, 78)]
def rec(k, r):          # lane <s22> of v(72 + k) := scalar register r   (v_writelane cannot take two SGPRs)
    return [f"\tv_mov_b32_e32 v80, {r}", f"\tv_cndmask_b32_e64 v{72 + k}, v{72 + k}, v80, s[72:73]"]
ins[L(1314)] = ["\tv_cmp_eq_u32_e64 s[72:73], s22, v79"] + sum((rec(k, r) for k, r in enumerate(("s28", "s56", "s37", "s18", "s59"))), [])
ins[L(1348)] = ["\tv_cmp_eq_u32_e64 s[72:73], s22, v79"] + rec(5, "s41")
ins[L(1379)] = [
    f"\ts_cmp_lg_u32 s78, {dbg}", "\ts_cbranch_scc1 .Ldbg_skip", "\ts_cmp_lg_u32 s58, 0", "\ts_cbranch_scc1 .Ldbg_skip",
    "\ts_load_dwordx2 s[76:77], s[74:75], 0x0", "\ts_waitcnt lgkmcnt(0)",
    "\tv_mbcnt_lo_u32_b32 v78, -1, 0", "\tv_mbcnt_hi_u32_b32 v78, -1, v78", "\tv_lshlrev_b32_e32 v78, 2, v78",
    f"\tv_add_u32_e32 v78, {(5202 + 1) * 1024}, v78"] + \
    [f"\tglobal_store_dword v78, v{72 + k}, s[76:77] offset:{256 * k}" for k in range(6)] + ["\ts_waitcnt vmcnt(0)", ".Ldbg_skip:"]
out = []
for i, l in enumerate(src):
    if i in ins:
        out.extend(ins[i])
    out.append(l)
txt = "\n".join(out)
k = txt.index(".amdhsa_kernel " + SYM)
j = txt.index(".amdhsa_next_free_sgpr", k)
e = txt.index("\n", j)
txt = txt[:j] + ".amdhsa_next_free_sgpr 80" + txt[e:]
open(sys.argv[2], "w").write(txt)
