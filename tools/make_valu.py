#!/usr/bin/env python3
"""profiles/valu.json: the issue-side roofs of the window kernels from rocprofv3 --pmc passes of the bench command.

usage: tools/make_valu.py <insts.csv> <active.csv> <ta.csv> <out.json> [<merge-into.json>] [--suffix @variant] [--head SHA]
  insts.csv   pass with SQ_INSTS_VALU (per dispatch, summed over the chip)
  active.csv  pass with SQ_ACTIVE_INST_VALU and GRBM_GUI_ACTIVE
  ta.csv      pass with TA_TA_BUSY_sum, TD_TD_BUSY_sum and GRBM_GUI_ACTIVE
Per kernel class (per launch, rocprofv3 serialises the launches: every kernel is ALONE on the chip):
  valu_issue_frac = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x kernel cycles)   (a wave64 vector instruction holds its
                    16-lane SIMD for 4 cycles; kernel cycles = GRBM_GUI_ACTIVE / 8 XCD instances)
  ta_busy_frac / td_busy_frac = TA_TA_BUSY_sum / TD_TD_BUSY_sum over (256 CUs x kernel cycles)"""
import collections
import csv
import json
import re
import sys

sys.path.insert(0, __import__("os").path.dirname(__file__))
from make_traffic import CLASSES  # noqa: E402


def per_launch(path):
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        for cls, pat in CLASSES:
            if re.search(pat, r["Kernel_Name"]):
                tot[cls][r["Counter_Name"]] += float(r["Counter_Value"])
                disp[cls].add(r["Dispatch_Id"])
    return {c: {k: v / len(disp[c]) for k, v in tot[c].items()} for c in tot}, {c: len(d) for c, d in disp.items()}


def main():
    argv = list(sys.argv)
    suffix = head = ""
    for flag in ("--suffix", "--head"):
        if flag in argv:
            i = argv.index(flag)
            val = argv[i + 1]
            del argv[i:i + 2]
            if flag == "--suffix":
                suffix = val
            else:
                head = val
    sys.argv = argv
    insts, n_i = per_launch(sys.argv[1])
    act, _ = per_launch(sys.argv[2])
    ta, _ = per_launch(sys.argv[3])
    out = {"source": "rocprofv3 --pmc passes (separate runs, no tracing domains) of python3 bench.py --steps 2 --warmup 1 "
                     "--no-cpu-baseline --no-side-legs --host-pairs 0 --no-profile [--mode planes]; per-launch averages; "
                     "rocprofv3 serialises the launches, so every kernel is alone on the chip here",
           "kernels": {}}
    if len(sys.argv) > 5:
        try:
            old = json.load(open(sys.argv[5]))
            out["kernels"] = old["kernels"]
            if "head" in old and not head:
                head = old["head"]
        except Exception:
            pass
    if head:
        out["head"] = head
    for cls, _ in CLASSES:
        if cls not in act:
            continue
        cyc = act[cls]["GRBM_GUI_ACTIVE"] / 8.0
        e = {"launches_sampled": n_i.get(cls), "kernel_cycles": cyc,
             "insts_valu_per_launch": insts.get(cls, {}).get("SQ_INSTS_VALU"),
             "valu_issue_frac": act[cls]["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * cyc)}
        if cls in ta:
            c2 = ta[cls]["GRBM_GUI_ACTIVE"] / 8.0
            e["ta_busy_frac"] = ta[cls]["TA_TA_BUSY_sum"] / (256.0 * c2)
            e["td_busy_frac"] = ta[cls]["TD_TD_BUSY_sum"] / (256.0 * c2)
        out["kernels"][cls + suffix] = e
    json.dump(out, open(sys.argv[4], "w"), indent=1)
    print(json.dumps(out["kernels"], indent=1))


if __name__ == "__main__":
    main()
