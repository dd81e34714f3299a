# -*- coding: utf-8 -*-
"""Registers / LDS / scratch of every gfx950 kernel in the built objects, from the code-object
metadata (what the hardware allocates: .vgpr_count is the UNIFIED count, arch + accumulation)."""
import glob, os, re, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
rows = []
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with tempfile.TemporaryDirectory() as tmp:
    so = os.path.join(tmp, "lib.so")
    subprocess.check_call(["cp", os.path.join(ROOT, "celerite_amd", "libcelerite_hip.so"), so])
    subprocess.run([LLVM + "/llvm-objdump", "--offloading", so], capture_output=True, cwd=tmp)
    for co in sorted(glob.glob(os.path.join(tmp, "lib.so.*gfx950*"))):
        notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
        for blk in notes.split("- .agpr_count:")[1:]:
            get = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
            name = re.search(r"\.name:\s+(\S+)", blk).group(1)
            name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            agpr = int(blk.split()[0])
            v = get("vgpr_count")
            alloc = (v + 7) // 8 * 8
            rows.append((name.replace("clr::", "").replace("(BatchParams)", "")[:70], v, agpr, get("sgpr_count"),
                         get("group_segment_fixed_size"), get("private_segment_fixed_size"),
                         512 // max(alloc, 1) if alloc else 8))
want = sys.argv[1:] or [""]
print("%-70s %6s %5s %5s %7s %8s %s" % ("kernel", "vgpr*", "agpr", "sgpr", "lds_B", "scratch", "waves/SIMD by registers"))
for r in rows:
    if any(w in r[0] for w in want):
        print("%-70s %6d %5d %5d %7d %8d %d" % (r[0], r[1], r[2], r[3], r[4], r[5], min(r[6], 8)))
print("# vgpr* = unified allocation (architectural + accumulation registers), as in the kernel descriptor")
