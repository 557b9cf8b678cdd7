#!/usr/bin/env python3
"""Fold the rocprofv3 --pmc passes of scripts/profile_bench.sh into one JSON (per-kernel mean per launch).

    python scripts/pmc_summary.py gpurun_out/<tag> profiles/r02/pmc_counters_<workload>_<tag>.json \
        --workload nms10_osd2 --frames 131072

Reads every <tag>_<PASS>/**/*_counter_collection.csv (one row per dispatch and counter) and the bench line of
<tag>_stats.log.  FETCH_SIZE / WRITE_SIZE are in KiB as rocprofv3 reports them (FETCH_SIZE under-counts wide
coalesced reads by 2x on gfx950, MI355X_MICROARCH.md HBM -- bench.py doubles it); SQ_* cycle counters are
quad-cycles summed over the chip; GRBM_GUI_ACTIVE is summed over the 8 XCDs.  Adds per kernel
    valu_issue_frac = SQ_ACTIVE_INST_VALU / (GRBM_GUI_ACTIVE / 8 / 4 * 1024 SIMDs)
-- a RATIO, not a utilisation: SQ_ACTIVE_INST_VALU counts ONE quad-cycle issue slot per vector instruction and wave (it
equals SQ_INSTS_VALU in every pass taken here), the denominator is the quad-cycles the 1024 SIMDs had during the launch
(GRBM_GUI_ACTIVE / 8 is the launch's duration in shader clocks: 2.3 M / 8 = 287 k cycles for the 115.7 us NMS launch =
2.48 GHz).  Instructions of the 2-cycle class (v_add/mul/fma/xor: profiles/r01/ubench_valu_issue*.txt) take less than
their slot, so a kernel whose vector ALUs never idle reads ABOVE 1 (NMS: 1.12; ADVICE r02 asked what the 12 % are).
For kernels that run fewer waves than the chip holds, SQ_WAIT_ANY / SQ_WAVE_CYCLES (share of a wave's life parked on
s_waitcnt) and SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES are printed too.
"""
from __future__ import annotations

import argparse
import collections
import csv
import glob
import json
import os
import re


def short(name):
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0].strip()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("prefix", help="gpurun_out/<tag> (the passes are <prefix>_FETCH_SIZE, _WRITE_SIZE, _SQ, _SQ2)")
    ap.add_argument("out")
    ap.add_argument("--workload", default="nms10_osd2")
    ap.add_argument("--frames", type=int, default=131072)
    ap.add_argument("--skip-launches", type=int, default=2, help="warm-up launches of every kernel left out of the mean")
    args = ap.parse_args()
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in [args.prefix + "_" + tag for tag in ("FETCH_SIZE", "WRITE_SIZE", "SQ", "SQ2")]:   # (exact names: another workload's tag may extend this one)
        if not os.path.isdir(d):
            continue
        for path in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            with open(path, newline="") as f:
                for row in csv.DictReader(f):
                    acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    per = {}
    for kern, counters in acc.items():
        if not kern.startswith("ldpc::"):
            continue
        per[kern] = {c: (sum(v[args.skip_launches:]) / max(1, len(v[args.skip_launches:])) if len(v) > args.skip_launches else sum(v) / len(v))
                     for c, v in counters.items()}
        per[kern]["launches_counted"] = max(len(v) for v in counters.values())
        g, a = per[kern].get("GRBM_GUI_ACTIVE"), per[kern].get("SQ_ACTIVE_INST_VALU")
        if g and a is not None:
            per[kern]["valu_issue_frac"] = a / (g / 8.0 / 4.0 * 1024.0)
        wc = per[kern].get("SQ_WAVE_CYCLES")
        if wc:
            for name, key in (("SQ_WAIT_ANY", "wait_any_share"), ("SQ_WAIT_INST_ANY", "wait_inst_share")):
                if name in per[kern]:
                    per[kern][key] = per[kern][name] / wc
    bench_line = None
    log = args.prefix + "_stats.log"
    if os.path.exists(log):
        for line in open(log):
            if line.startswith("{") and '"metric"' in line:
                bench_line = json.loads(line)
    out = dict(bench_workload=args.workload, frames_per_launch=args.frames,
               units="FETCH_SIZE/WRITE_SIZE in KiB as rocprofv3 reports them; on gfx950 FETCH_SIZE under-counts wide coalesced reads "
                     "by 2x (MI355X_MICROARCH.md HBM); SQ_* cycle counters in quad-cycles summed over the chip; GRBM_GUI_ACTIVE summed "
                     "over the 8 XCDs; valu_issue_frac = SQ_ACTIVE_INST_VALU / (GRBM_GUI_ACTIVE / 8 / 4 * 1024)",
               per_launch_mean=per, bench_line_of_the_stats_pass=bench_line)
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    for kern, c in sorted(per.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
        print(f"{kern:40s} FETCH {c.get('FETCH_SIZE', 0):10.1f} KiB  WRITE {c.get('WRITE_SIZE', 0):10.1f} KiB  valu_issue {c.get('valu_issue_frac', float('nan')):.2f}")


if __name__ == "__main__":
    main()
