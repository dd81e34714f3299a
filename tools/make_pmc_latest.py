# -*- coding: utf-8 -*-
"""profiles/pmc_latest.json from the summaries of a tools/prof_run_r05.sh run (gpurun_out/prof_r05/*_summary.txt):
HBM bytes per launch of the kernels bench.py reports `traffic` for = FETCH_SIZE x 2 + WRITE_SIZE (KiB per dispatch;
the x2 on gfx950 per /opt/skills/guides/MI355X_MICROARCH.md, HBM section), stamped with the fingerprint of the kernel
sources the counters were taken from (bench.csrc_fingerprint) -- bench.py nulls `traffic` when the kernels have changed.
Usage: python tools/make_pmc_latest.py gpurun_out/prof_r05 [round tag]"""
import json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

src = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else "r06"


def table(name):
    out = {}
    path = os.path.join(src, name + "_summary.txt")
    if not os.path.exists(path):
        return out
    for ln in open(path):
        m = re.match(r"(.{120})\| (\S+)\s+n=(\d+)\s+avg=(\S+)\s+avg_dur_us=(\S+)", ln)
        if m:
            out[m.group(1).strip()] = (float(m.group(4)), int(m.group(3)), float(m.group(5)))
    return out


def traffic(fetch, write, key):
    f = [v for k, v in fetch.items() if key in k]
    w = [v for k, v in write.items() if key in k]
    if not f or not w:
        return None
    return int((2.0 * f[0][0] + w[0][0]) * 1024)


hf, hw = table("headline_fetch"), table("headline_write")
wf, ww = table("wide_fetch"), table("wide_write")
old = json.load(open(os.path.join(bench.ROOT, "profiles", "pmc_latest.json")))
rec = {
    "source": "profiles/%s_pmc_counters.txt (headline), %s_wide_pmc_counters.txt (config 4): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, "
              "separate passes (tools/prof_run_r05.sh), KiB per dispatch, FETCH_SIZE x2 on gfx950 per MI355X_MICROARCH.md; "
              "the accuracy-family and gradient entries are round 3's (kernels unchanged since: clr_batch_kernels.h warm_kernel, "
              "clr_grad_kernels.h)" % (tag, tag),
    "csrc_fingerprint": bench.csrc_fingerprint(),
    "config": {"batch": 1024, "N": 100000, "J_real": 2, "J_comp": 3, "chunks": 64},
    "traffic_bytes_per_launch": {
        "summarize": traffic(hf, hw, "summarize_split_kernel<2, 3, true, true"),
        "summarize (single wave)": traffic(hf, hw, "summarize_kernel<2, 3, true, true"),
        "prefix": None,
        "correct": traffic(hf, hw, "correct_kernel<8>"),
        "replay (materialising)": traffic(hf, hw, "replay_kernel<2, 3, 2, true, false, false>"),
        "replay (materialising, lean)": traffic(hf, hw, "replay_kernel<2, 3, 3, true, false, false>"),
        "chunk-head fix-up of a materialising run (round 6)": traffic(hf, hw, "replay_kernel<2, 3, 2, true, false, true>"),
    },
    "other_shapes": dict(old.get("other_shapes", {})),
}
gc, sa = traffic(hf, hw, "group_compose_kernel<8>"), [v for k, v in hf.items() if "seg_advance_kernel<8>" in k]
if gc is not None and sa:
    saw = [v for k, v in hw.items() if "seg_advance_kernel<8>" in k]
    rec["traffic_bytes_per_launch"]["prefix"] = int(gc + 2 * (2.0 * sa[0][0] + saw[0][0]) * 1024)
w = traffic(wf, ww, "wide_scan_kernel<32, true, 1, true")
if w is not None:
    rec["other_shapes"] = {k: v for k, v in rec["other_shapes"].items() if not k.startswith("config4")}
    rec["other_shapes"]["config4 wide summarize (B=256, N=1e5, width 32, 8 chunks)"] = w
json.dump(rec, open(os.path.join(bench.ROOT, "profiles", "pmc_latest.json"), "w"), indent=1)
print(json.dumps(rec, indent=1))
