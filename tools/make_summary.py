#!/usr/bin/env python3
"""Dev tool: gpurun_out/<tag>/<shape>_<prec>/ (tools/collect_profiles.sh) -> profiles/<tag>_<shape>_<prec>_kernel_stats.csv,
profiles/<tag>_<shape>_<prec>_counters.md, and the PMC traffic figures bench.py reads, merged into profiles/r03_traffic.json
under the key "<shape>:<prec>".      make_summary.py <tag> <shape> <prec> "<source note>"
Units per MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts wide coalesced reads at half
their bytes (doubled in the traffic column); SQ_* cycle counters are quad-cycles summed over waves."""
import csv, json, os, shutil, sys
tag, shape, prec, note = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else ""
src, dst = f"gpurun_out/{tag}/{shape}_{prec}", "profiles"
base = f"{dst}/{tag}_{shape}_{prec}"
load = lambda n: json.load(open(f"{src}/pmc_{n}.json")) if os.path.exists(f"{src}/pmc_{n}.json") else {}
fe, wr, l2, lds, occ, mix = (load(n) for n in ("fetch", "write", "l2", "lds", "occ", "mix"))
dr = load("dram")        # round 6: TCC_EA0_RDREQ_DRAM_32B_sum / TCC_EA0_WRREQ_WRITE_DRAM_32B_sum, x 32 B, byte-exact (profiles/r06_dram_calib.md)
if dr:                   # expressed in the units the table below expects: "FETCH_SIZE" KiB such that 2 x it = the read bytes
    for k, v in dr.items():
        n = v.get("launches", 1)
        fe[k] = {"launches": n, "FETCH_SIZE": 32.0 * v.get("TCC_EA0_RDREQ_DRAM_32B_sum", 0.0) / 2.0 / 1024.0}
        wr[k] = {"launches": n, "WRITE_SIZE": 32.0 * v.get("TCC_EA0_WRREQ_WRITE_DRAM_32B_sum", 0.0) / 1024.0}
out = [f"# {tag} / {shape} / {prec}: rocprofv3 counters per kernel ({note})", "",
       f"Command of every pass: `rocprofv3 --pmc <counters> -- python3 bench.py --shape {shape} --precision {prec} --no-cpu --no-f64 --no-netflix --no-rows --no-profile [...]` "
       "(one pass per counter group; `--kernel-trace --stats` in a pass of its own -> the kernel_stats.csv beside this file). "
       "PMC passes serialise the kernels: concurrent U-step classes are seen alone.", ""]
if os.path.exists(f"{src}/kernel_stats.csv"):
    shutil.copy(f"{src}/kernel_stats.csv", base + "_kernel_stats.csv")
    out += ["| kernel (rocprofv3 --kernel-trace --stats) | calls | total us | avg us | % |", "|---|---|---|---|---|"]
    for r in list(csv.DictReader(open(f"{src}/kernel_stats.csv")))[:24]:
        out.append(f"| `{r['Name'][:78]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e3:.1f} | {float(r['AverageNs'])/1e3:.1f} | {float(r['Percentage']):.2f} |")
    out.append("")
names = sorted(set(fe) | set(wr) | set(l2) | set(lds), key=lambda k: -(fe.get(k, {}).get("FETCH_SIZE", 0) + wr.get(k, {}).get("WRITE_SIZE", 0)))
traffic = {}
if dr:
    out += ["Traffic columns: the byte-exact 32-byte-unit counters (`TCC_EA0_RDREQ_DRAM_32B_sum`, `TCC_EA0_WRREQ_WRITE_DRAM_32B_sum` x 32 B, one pass); "
            "'FETCH raw' = read bytes / 2 (what FETCH_SIZE would have shown).  These are the L2s' requests to local memory: Infinity-Cache hits INCLUDED; "
            "the HBM side of the same workload is the `hbm` block of the bench line (memory-controller activity).", ""]
out += ["| kernel | launches | FETCH raw MB/launch | WRITE MB/launch | traffic = 2 x FETCH + WRITE MB/launch | L2 hit rate (TCC_HIT / (HIT + MISS)) | LDS bank-conflict cycles / LDS active cycles | LDS instructions per wave-kcycle |",
        "|---|---|---|---|---|---|---|---|"]
for k in names:
    if k.startswith("__amd") or k.startswith("k_spin") or k.startswith("k_nop"):
        continue
    f, w = fe.get(k, {}), wr.get(k, {})
    n = max(f.get("launches", 0), w.get("launches", 0), 1)
    fb = 1024 * f.get("FETCH_SIZE", 0) / max(f.get("launches", 1), 1); wb = 1024 * w.get("WRITE_SIZE", 0) / max(w.get("launches", 1), 1)
    if f or w:
        traffic[k] = {"launches": int(n), "fetch_bytes_per_launch_raw": fb, "write_bytes_per_launch": wb}
    h = l2.get(k, {}); hit = h.get("TCC_HIT_sum", 0); miss = h.get("TCC_MISS_sum", 0)
    ld = lds.get(k, {}); bc = ld.get("SQ_LDS_BANK_CONFLICT", 0); act = ld.get("SQ_LDS_IDX_ACTIVE", 0); wc = ld.get("SQ_WAVE_CYCLES", 0)
    out.append(f"| `{k[:78]}` | {int(n)} | {fb/1e6:.2f} | {wb/1e6:.2f} | {(2*fb+wb)/1e6:.2f} | "
               + (f"{100*hit/(hit+miss):.1f} %" if hit + miss else "-") + " | "
               + (f"{100*bc/act:.1f} %" if act else "-") + " | " + (f"{1000*ld.get('SQ_INSTS_LDS', 0)/(4*wc):.1f}" if wc else "-") + " |")
open(base + "_counters.md", "w").write("\n".join(out) + "\n")
tpath = f"{dst}/{tag[:3]}_traffic.json"
tj = json.load(open(tpath)) if os.path.exists(tpath) else {"workloads": {}}
tj["workloads"][f"{shape}:{prec}"] = {"source": f"profiles/{tag[:3]}_traffic.json [{shape}:{prec}] ({tag}, {note})", "kernels": traffic}
json.dump(tj, open(tpath, "w"), indent=0)
for tbl, title in (("pmc_occupancy_table.txt", "Achieved occupancy (SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE, its own pass)"),
                   ("pmc_mix_table.txt", "Where the wave cycles go (SQ_WAVE_CYCLES SQ_ACTIVE_INST_* SQ_WAIT_* SQ_INSTS_*, its own pass)")):
    if os.path.exists(f"{src}/{tbl}"):
        with open(base + "_counters.md", "a") as fo:
            fo.write(f"\n{title}:\n\n" + open(f"{src}/{tbl}").read().rstrip() + "\n")
print("\n".join(out[:40]))
