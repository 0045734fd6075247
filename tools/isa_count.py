#!/usr/bin/env python3
"""Static instruction mix of the engine's kernels (no GPU needed).

usage: tools/isa_count.py [substring-of-mangled-kernel-name ...]
Compiles the kernel-holding units to gfx950 assembly and prints, per kernel whose mangled name contains one of
the substrings (default: the 11x11 RUNBLK2 kernels), VGPR/SGPR use and the count of VALU / SALU /
VMEM / LDS instructions of the whole body and of its largest loop."""
import os, re, subprocess, sys, collections

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = "/tmp/isa/all_units.s"


def build():
    os.makedirs("/tmp/isa", exist_ok=True)
    parts = []
    for unit in ("pm_launch", "pm_sweeps", "pm_seed", "pm_planes_host", "pm_imaging", "pm_tiled"):  # the units that hold kernels
        out = "/tmp/isa/%s.s" % unit
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off",
               "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wno-pass-failed", "-Wno-unused-command-line-argument",
               "-I" + ROOT + "/include", "-I" + ROOT + "/ocean-perception_amd/csrc",
               "-I" + ROOT + "/ocean-perception_amd/host", "-S", "--cuda-device-only", "-o", out,
               ROOT + "/ocean-perception_amd/csrc/%s.hip" % unit]
        subprocess.run(cmd, check=True)
        parts.append(open(out).read())
    open(OUT, "w").write("\n".join(parts))


def kinds(lines):
    c = collections.Counter()
    for l in lines:
        m = re.match(r"\s+([a-z_0-9]+)", l)
        if not m:
            continue
        op = m.group(1)
        if op.startswith("v_"):
            c["valu"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
            if op == "s_nop":
                c["s_nop"] += 1
            if op == "s_waitcnt":
                c["s_waitcnt"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            c["vmem"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
    return dict(c)


def main():
    pats = sys.argv[1:] or ["k_runblk2ILi0ELi32ELi0ELi11ELi11E", "k_runblk2ILi0ELi32ELi1ELi11ELi11E"]
    build()
    txt = open(OUT).read().split("\n")
    starts = [i for i, l in enumerate(txt) if re.match(r"^_Z\w+:", l)]
    for i in starts:
        name = txt[i].split(":")[0]
        if not any(p in name for p in pats):
            continue
        j = next(k for k in range(i, len(txt)) if txt[k].startswith(".Lfunc_end"))
        body = txt[i:j]
        # loops: from a label marked "Loop Header" to the last branch back to it
        labels = {m.group(1): k for k, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
        best = (0, 0, "")
        for lab, k in labels.items():
            if "Loop Header" not in body[k] and "Inner Loop Header" not in body[k]:
                continue
            back = [q for q in range(k, len(body)) if re.search(r"s_c?branch\w*\s+" + re.escape(lab) + r"\b", body[q])]
            if back and back[-1] - k > best[1] - best[0]:
                best = (k, back[-1], lab)
        vg = [l for l in txt[j:j + 60] if "num_vgpr" in l or "numbered_sgpr" in l]
        print(name)
        print("  regs:", " ".join(l.split(".")[-1].strip() for l in vg))
        print("  whole body:", kinds(body))
        if best[2]:
            print("  largest loop %s (%d lines):" % (best[2], best[1] - best[0]), kinds(body[best[0]:best[1] + 1]))


if __name__ == "__main__":
    main()
