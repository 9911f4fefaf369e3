#!/usr/bin/env python3
"""Static check of the device assembly behind the inline-assembly LDS reads of the streaming loops (hg_kernels.h:
lds_read128 / lds_wait / pin_after_wait): between a `ds_read_b128` issued from inline assembly and the first
`s_waitcnt lgkmcnt(0)` behind it, no instruction may mention the registers it writes (the compiler does not know they are
still in flight: a copy placed there would read stale data).  usage: asm_check.py file.s   (exit 1 on a violation)"""
import re
import sys

lines = open(sys.argv[1]).read().splitlines()
bad = 0
checked = 0
i = 0
while i < len(lines):
    if "#ASMSTART" in lines[i] and i + 1 < len(lines) and "ds_read_b128" in lines[i + 1]:
        m = re.search(r"ds_read_b128\s+v\[(\d+):(\d+)\]", lines[i + 1])
        regs = set(range(int(m.group(1)), int(m.group(2)) + 1))
        checked += 1
        j = i + 2
        while j < len(lines) and "s_waitcnt lgkmcnt(0)" not in lines[j] and "s_endpgm" not in lines[j]:
            t = lines[j].strip()
            if t and not t.startswith((";", ".", "ds_read_b128")) and "ds_read_b128" not in t:
                used = set()
                for a, b in re.findall(r"v\[(\d+):(\d+)\]", t):
                    used |= set(range(int(a), int(b) + 1))
                used |= {int(x) for x in re.findall(r"\bv(\d+)\b", t)}
                if used & regs and not t.startswith(("v_lshlrev", "v_and", "v_add", "v_bfe", "v_lshl_add")):  # address arithmetic may reuse nothing of these
                    print("line %d: %s   (registers of the read at line %d in flight)" % (j + 1, t, i + 2))
                    bad += 1
                elif used & regs:
                    print("line %d: %s   (touches in-flight registers of the read at line %d)" % (j + 1, t, i + 2))
                    bad += 1
            j += 1
    i += 1
print("inline-assembly LDS reads checked: %d, violations: %d" % (checked, bad))
sys.exit(1 if bad else 0)
