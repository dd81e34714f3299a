# -*- coding: utf-8 -*-
"""Copy the summaries of a tools/prof_run_r05.sh run (gpurun_out/prof_r05) into profiles/ under a round tag:
<tag>_kernel_trace_stats.txt / <tag>_pmc_counters.txt (headline), <tag>_wide_* (config 4), <tag>_wide64_*.
Usage: python tools/collect_prof_r05.py gpurun_out/prof_r05 r05f ; then tools/make_pmc_latest.py for pmc_latest.json."""
import os, shutil, sys
src, tag = sys.argv[1], sys.argv[2]
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
for name, out in (("headline", tag), ("wide", tag + "_wide"), ("wide64", tag + "_wide64"), ("small", tag + "_small"), ("rows", tag + "_rows")):
    tr = os.path.join(src, name + "_kernel_trace_stats.txt")
    if os.path.exists(tr):
        shutil.copy(tr, os.path.join(root, out + "_kernel_trace_stats.txt"))
    lines = ["# rocprofv3 --pmc passes (tools/prof_run_r05.sh %s), per dispatch averages; FETCH_SIZE / WRITE_SIZE in KiB (FETCH_SIZE x2 on "
             "gfx950); SQ_* per counter instance (32 instances)\n" % name]
    for kind in ("fetch", "write", "sq"):
        p = os.path.join(src, "%s_%s_summary.txt" % (name, kind))
        if os.path.exists(p):
            lines += ["%s: %s" % (kind, ln) for ln in open(p) if ln.strip()]
    if len(lines) > 1:
        open(os.path.join(root, out + "_pmc_counters.txt"), "w").writelines(lines)
        print("wrote", out + "_pmc_counters.txt", len(lines) - 1, "lines")
