# -*- coding: utf-8 -*-
"""Turns a rocprofv3 `--kernel-trace --stats` SQLite database (rocpd format, the
default output of ROCm 7.2's rocprofv3) into the plain-text per-kernel summary
that is committed under profiles/."""
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    rows = c.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), "
        "max(grid_x), max(grid_y), max(workgroup_x) from kernels group by name order by sum(duration) desc"
    ).fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["%-100s %6s %12s %12s %12s %12s %6s %5s %5s %5s %7s %8s  %s" % (
        "kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "vgpr", "agpr", "sgpr",
        "lds_B", "scratch", "grid x wg")]
    for r in rows:
        lines.append("%-100s %6d %12.1f %12.1f %12.1f %12.1f %6.2f %5d %5d %5d %7d %8d  (%d,%d) x %d" % (
            r[0][:100], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / total,
            r[6], r[7], r[8], r[9], r[10], r[11], r[12], r[13]))
    lines.append("# vgpr / agpr: rocprofv3's architectural and accumulation register counts as recorded in the trace (for "
                 "kernels that spill state into AGPRs the trace shows the architectural part only); the unified allocation "
                 "that sets the waves per SIMD is in the code-object metadata: tools/kernel_resources.py")
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main(*sys.argv[1:3])
