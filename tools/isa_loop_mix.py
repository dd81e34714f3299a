# -*- coding: utf-8 -*-
"""Instruction mix of a kernel's basic blocks from hipcc's --save-temps assembly (gfx950): per block the instruction
count by class (fp64 arithmetic, DPP moves, v_readlane, selects, plain moves, LDS, global, SALU, waits, branches).
Usage: isa_loop_mix.py <file.s> <mangled-name-substring> [min_block_size]"""
import collections
import re
import sys

path, key = sys.argv[1], sys.argv[2]
minsize = int(sys.argv[3]) if len(sys.argv) > 3 else 40
src = open(path).read()
m = re.search(r"^(\S*%s\S*):" % re.escape(key), src, re.M)
name = m.group(1)
i = src.index(name + ":")
j = src.index(".Lfunc_end", i)
blocks, cur = [], None
for ln in src[i:j].split("\n"):
    mm = re.match(r"^(\.LBB\d+_\d+):", ln)
    if mm:
        cur = mm.group(1)
        blocks.append([cur, []])
        continue
    if cur and ln.startswith("\t") and not ln.strip().startswith((".", ";")):
        blocks[-1][1].append(ln.strip())


def cls(x):
    op = x.split()[0]
    if "dpp" in x and op.startswith("v_mov"):
        return "dpp mov"
    if op.startswith(("v_fma_f64", "v_fmac_f64", "v_mul_f64", "v_add_f64")):
        return "fp64 fma/mul/add"
    if "f64" in op:
        return "fp64 other"
    if op.startswith("v_readlane") or op.startswith("v_readfirstlane"):
        return "v_readlane"
    if op.startswith("v_cndmask"):
        return "v_cndmask"
    if op.startswith(("v_mov", "v_accvgpr")):
        return "v_mov/accvgpr"
    if op.startswith("v_cmp"):
        return "v_cmp"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_"):
        return "valu other (int / 32-bit)"
    return "other"


print("kernel", name)
names = [b[0] for b in blocks]
for k, (nm, ins) in enumerate(blocks):
    tg = [x.split()[-1] for x in ins if x.startswith(("s_cbranch", "s_branch"))]
    back = [t for t in tg if t in names and names.index(t) <= k]
    if len(ins) < minsize and not back:
        continue
    c = collections.Counter(cls(x) for x in ins)
    print("%s: %d instructions%s" % (nm, len(ins), ("  (loops back to %s)" % back) if back else ""))
    print("   " + ", ".join("%s %d" % kv for kv in c.most_common()))
