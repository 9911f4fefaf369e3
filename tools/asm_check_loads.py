#!/usr/bin/env python3
"""Static check of the device assembly of the resident sweep kernel (hg_resident.hip.h).  The columns that refill the window land in
vector registers v224 .. v255 (and the lanes' per-set values in v216 .. v220), named by hand in inline assembly and loaded by
instructions the compiler does not see as loads; the kernel carries a register cap (amdgpu_num_vgpr of HALF the limit: the attribute counts the unified VGPR + AGPR file), so the compiler must never allocate them.  This script verifies exactly that on the emitted
code: in every k_sweep_resident kernel, no instruction outside an inline-assembly block names a register >= v216, and inside
inline assembly only the expected instructions do (global_load_dword[x2], v_and_b32, v_mov_b32, v_readlane_b32).
Callees are compiled without that cap, so the kernel body may call nothing but the walkers (res_walker, res_walker2: the walker's
workgroup has no column loads in flight and everything the walkers call is reached from there only): every function symbol the body
materialises must be one of those and every s_swappc_b64 must have such a symbol, so a helper of the streaming path that stopped
being inlined fails this check instead of silently overwriting the landing registers.
usage: asm_check_loads.py file.s   (exit 1 on a violation)"""
import re
import sys

LIMITS = {"k_sweep_resident": 216, "k_sweep_limb": 200, "k_sweep_limb4": 208}  # hg_resident.hip.h: RS_VGPR_LIMIT; hg_streamer2.hip.h: RL_VGPR_LIMIT (the same discipline, more named registers)
text = open(sys.argv[1]).read().splitlines()


def vregs(s):
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", s):
        out |= set(range(int(a), int(b) + 1))
    out |= {int(x) for x in re.findall(r"\bv(\d+)\b", s)}
    return out


funcs = {m.group(1) for t in text for m in [re.match(r"^\s*\.type\s+(\S+),@function", t)] if m}
bad = checked = 0
i, n = 0, len(text)
while i < n:
    m = re.match(r"^(\S*?(k_sweep_resident|k_sweep_limb4|k_sweep_limb)I\S*):\s", text[i])
    if m and not text[i].startswith("."):
        LIMIT = LIMITS[m.group(2)]
        name, inasm, loads, reads = m.group(1), False, 0, 0
        calls, targets = 0, []
        j = i + 1
        while j < n and not text[j].startswith(".Lfunc_end"):  # (the body may hold several s_endpgm: early returns)
            t = text[j].strip()
            if "#ASMSTART" in t:
                inasm = True
            elif "#ASMEND" in t:
                inasm = False
            elif t and not t.startswith((";", ".")) and not t.endswith(":"):
                t = t.split(";")[0]
                if t.split()[0] in ("s_swappc_b64", "s_setpc_b64", "s_call_b64"):
                    calls += 1
                for sym in re.findall(r"(\S+)@rel32@lo", t):
                    if sym in funcs:
                        targets.append(sym)
                        if "res_walker" not in sym:
                            print("%s line %d: the kernel body calls %s (only the walkers may be called: callees do not honour the register cap)" % (name[:48], j + 1, sym[:60]))
                            bad += 1
                hi = {r for r in vregs(t) if r >= LIMIT}
                if hi and not inasm:
                    print("%s line %d: %s   (names v%s outside inline assembly)" % (name[:48], j + 1, t.strip()[:80], sorted(hi)[:4]))
                    bad += 1
                elif hi:
                    op = t.split()[0]
                    if op.startswith("global_load_dword"):
                        loads += 1
                    elif op in ("v_and_b32", "v_mov_b32", "v_readlane_b32"):
                        reads += 1
                    else:
                        print("%s line %d: %s   (unexpected inline-assembly use of the landing registers)" % (name[:48], j + 1, t.strip()[:80]))
                        bad += 1
            j += 1
        print("%s: %d loads into / %d reads of the landing registers, all inside inline assembly" % (name, loads, reads))
        print("  calls in the body: %d, function symbols materialised: %s" % (calls, sorted({re.sub(r"^_ZN2hg\d+", "", x)[:12] for x in targets})))
        if calls > len(targets):
            print("  ... a call without a walker symbol of its own: its target cannot be checked")
            bad += 1
        if loads == 0 or reads == 0:
            print("  ... but none were found: the check does not see what it is meant to check")
            bad += 1
        checked += 1
        i = j
    i += 1
print("kernels checked: %d, violations: %d" % (checked, bad))
sys.exit(1 if bad or not checked else 0)
