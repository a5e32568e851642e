#!/usr/bin/env python3
"""Dev tool: the rocprofv3 --pmc passes of tools/ubench/dram_calib.hip -> a per-dispatch table of every memory-side TCC counter
against the bytes the kernel is known to move.      pmc_dram_calib.py <dir with one sub-directory per pass> [kernel_trace.csv]
Answers (profiles/r06_dram_calib.md): which unit each counter counts in on gfx950 (32 / 64 / 128 B), and whether
TCC_EA0_RDREQ_DRAM* excludes Infinity-Cache hits (k_calib_reread: launches 2..8 of a 64 MiB buffer)."""
import collections, csv, glob, os, sys

root = sys.argv[1]
KNOWN = {  # kernel prefix -> (bytes read, bytes written) per launch
    "k_calib_read": (1 << 30, 0), "k_calib_write": (0, 1 << 30), "k_calib_copy": (1 << 30, 1 << 30), "k_calib_reread": (64 << 20, 0),
    "k_calib_l2read": (32 * (2 << 20), 0), "void k_calib_gather": ((4 << 20) * 400, 0)}
val = collections.defaultdict(lambda: collections.defaultdict(float))     # (kernel, dispatch order) -> counter -> value
order = {}
for f in sorted(glob.glob(os.path.join(root, "*", "*", "*counter_collection.csv")) + glob.glob(os.path.join(root, "*", "*", "*", "*counter_collection.csv"))):
    seq = collections.Counter()
    seen = {}
    for r in csv.DictReader(open(f)):
        k, d = r["Kernel_Name"], r["Dispatch_Id"]
        if "k_calib" not in k:
            continue
        if (k, d) not in seen:
            seen[(k, d)] = seq[k]
            seq[k] += 1
        val[(k, seen[(k, d)])][r["Counter_Name"]] += float(r["Counter_Value"])
dur = collections.defaultdict(list)
if len(sys.argv) > 2:
    for r in csv.DictReader(open(sys.argv[2])):
        if "k_calib" in r["Kernel_Name"]:
            dur[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
counters = sorted({c for v in val.values() for c in v})
print("| kernel | launch | known read MB | known write MB | us | " + " | ".join(counters) + " |")
print("|---|---|---|---|---|" + "---|" * len(counters))
for (k, i) in sorted(val, key=lambda ki: (list(KNOWN).index(next(p for p in KNOWN if ki[0].startswith(p))), ki[0], ki[1])):
    kr, kw = KNOWN[next(p for p in KNOWN if k.startswith(p))]
    d = sorted(dur.get(k, []))
    us = f"{d[i][1]:.1f}" if i < len(d) else "-"
    cells = []
    for c in counters:
        v = val[(k, i)].get(c)
        if v is None:
            cells.append("-")
        elif c in ("FETCH_SIZE", "WRITE_SIZE"):
            cells.append(f"{v * 1024 / 1e6:.1f} MB")
        else:
            known = kw if "WR" in c else kr
            cells.append(f"{v:.4g}" + (f" ({known / v:.1f} B/req)" if v > 0 and known else ""))
    print(f"| `{k[:40]}` | {i + 1} | {kr / 1e6:.1f} | {kw / 1e6:.1f} | {us} | " + " | ".join(cells) + " |")
