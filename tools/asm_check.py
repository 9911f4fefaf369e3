#!/usr/bin/env python3
"""Static check of the device assembly behind the inline-assembly LDS reads of the streaming loops (hg_kernels.h:
lds_read128 / lds_wait / pin_after_wait): between a `ds_read_b128` issued from inline assembly and the first
`s_waitcnt lgkmcnt(0)` behind it, no instruction may mention the registers it writes (the compiler does not know they are
still in flight: a copy placed there would read stale data).  Second: the issue order of LDS-DMA loads and plain loads that the
hand-placed `s_waitcnt vmcnt(N > 0)` of the same loops rely on (see below).  usage: asm_check.py file.s   (exit 1 on a violation)"""
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

# Second check: the streaming loops wait with a hand-placed `s_waitcnt vmcnt(N)`, N > 0 = the plain column loads of the most recent
# tile, for everything older -- among it the LDS-DMA loads (global_load_lds_*) of the eps tile the inline ds_read is about to read.
# Loads return in issue order, so that holds iff in the block that issues a tile's loads every global_load_lds_* comes BEFORE the
# plain loads (the source issues dma_eps() and then the tile's N column loads; blocks that issue fewer are followed by a full
# drain in the source).  Nothing in the source orders a DMA against a plain load (the compiler sees no alias between them): this
# is where a reordering by the machine scheduler would show.
VMEM = re.compile(r"^\s*(global_load|global_store|global_atomic|buffer_load|buffer_store|buffer_atomic|flat_load|flat_store|flat_atomic|scratch_)")
func_start = [k for k, l in enumerate(lines) if re.match(r"^[A-Za-z_][\w.$]*:\s*(;.*)?$", l) and not l.startswith(".L")]
func_start.append(len(lines))
blocks = 0
for a, b in zip(func_start[:-1], func_start[1:]):
    counts = set()
    for k in range(a, b - 1):
        if "#ASMSTART" in lines[k]:
            m = re.match(r"\s*s_waitcnt vmcnt\((\d+)\)", lines[k + 1])
            if m and int(m.group(1)) > 0:
                counts.add(int(m.group(1)))
    if not counts:
        continue
    blk = []  # VMEM instructions of the current basic block: (line, is_dma)
    def close_block():
        global bad, blocks
        dma = [q for q, (_, d) in enumerate(blk) if d]
        if dma and len(dma) < len(blk):
            blocks += 1
            if any(not d for _, d in blk[:dma[-1]]):
                print("line %d: an LDS-DMA load behind a plain load of the same block (vmcnt(N) no longer covers it)" % (blk[dma[-1]][0] + 1))
                bad += 1
    for k in range(a, b):
        t = lines[k]
        if re.match(r"^\.LBB\w+:", t) or re.match(r"^\s*s_(c?branch|endpgm|setpc)", t):
            close_block()
            blk = []
        elif VMEM.match(t):
            blk.append((k, "global_load_lds" in t))
    close_block()
print("inline-assembly LDS reads checked: %d, load-issue blocks with LDS-DMA and plain loads checked: %d, violations: %d" % (checked, blocks, bad))
sys.exit(1 if bad else 0)
