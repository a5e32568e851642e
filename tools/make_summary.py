#!/usr/bin/env python3
"""Dev tool: profiles/<tag>_* from gpurun_out/<tag>/ (tools/collect_profiles.sh): copies the small artefacts and writes
profiles/<tag>_summary.md.   make_summary.py <tag> "<title line>" """
import csv, json, os, shutil, sys
tag, title = sys.argv[1], sys.argv[2]
src, dst = f"gpurun_out/{tag}", "profiles"
for a, b in [("kernel_stats.csv", f"{tag}_kernel_stats.csv"), ("pmc_traffic_table.txt", f"{tag}_pmc_traffic_table.txt"),
             ("bench.json", f"{tag}_bench.json"), ("bench_noprofile.json", f"{tag}_bench_noprofile.json"), ("traffic.json", "r02_traffic.json")]:
    shutil.copy(os.path.join(src, a), os.path.join(dst, b))
b = json.loads(open(f"{src}/bench.json").read().strip().split("\n")[-1])
bn = json.loads(open(f"{src}/bench_noprofile.json").read().strip().split("\n")[-1])
cpu = b.get("cpu_baseline") or {}
out = [f"# {title}", "",
       "Commands (GPU box, 1x MI355X, ml1m-shaped synthetic, PrimalCR++ k=100 lambda=5000, fp32 storage / fp64 accumulation; `tools/collect_profiles.sh`):", "",
       f"* `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu` -> `{tag}_kernel_stats.csv`",
       f"* `rocprofv3 --pmc FETCH_SIZE -- python3 bench.py --no-cpu --no-profile --steps 10` and the same with `--pmc WRITE_SIZE` (separate passes) -> `r02_traffic.json`, `{tag}_pmc_traffic_table.txt` (tools/pmc_traffic.py; counters in KiB, FETCH_SIZE also given x2 per MI355X_MICROARCH.md)",
       f"* `python3 bench.py --verbose` -> `{tag}_bench.json`; `python3 bench.py --no-profile --no-cpu` -> `{tag}_bench_noprofile.json`", "",
       f"bench: {b['ms_per_step']:.2f} ms per outer iteration with sampled event timing ({bn['ms_per_step']:.2f} ms without = {bn['value']:.3e} pairs/s), "
       f"test NDCG@10 {b['ndcg10_test']:.4f}, pairwise error {b['pairwise_error_test']:.4f}; "
       + (f"reference OpenMP -n {cpu.get('cores')}: {cpu.get('s_per_iter', 0):.2f} s per iteration ({b.get('speedup_vs_cpu_baseline', 0):.0f}x)." if cpu else ""), "",
       (f"fp64 storage (the reference's arithmetic type), same run: {b['f64']['ms_per_step']:.2f} ms per outer iteration, test NDCG@10 {b['f64']['ndcg10_test']:.4f}. " if b.get("f64") else "")
       + (f"reference -n 1: {cpu['single_thread']['s_per_iter']:.2f} s per iteration." if cpu.get("single_thread") else ""), "",
       "roofline (dominant kernel = most GPU time): " + json.dumps(b["roofline"]), "",
       "roofline_phase: " + json.dumps(b.get("roofline_phase")), "",
       "| kernel (rocprofv3 --kernel-trace --stats) | calls | total us | avg us | % |", "|---|---|---|---|---|"]
rows = list(csv.DictReader(open(f"{src}/kernel_stats.csv")))
for r in rows[:28]:
    out.append(f"| `{r['Name'][:64]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e3:.1f} | {float(r['AverageNs'])/1e3:.1f} | {float(r['Percentage']):.2f} |")
out += ["", "HIP-event averages from the same build inside bench.py (sampled launches timed, see the roofline note; share = share of the summed GPU time, concurrent length classes not discounted):", "",
        "| slot | avg us | launches | GPU-time share | algorithmic MB | algorithmic GB/s | frac of 8 TB/s | PMC bytes/launch (2 x FETCH + WRITE) |", "|---|---|---|---|---|---|---|---|"]
for k, v in sorted(b["kernels"].items(), key=lambda kv: -kv[1]["gpu_time_share"]):
    out.append(f"| {k} | {v['avg_us']} | {v['launches']} | {v['gpu_time_share']} | {v['algorithmic_bytes']/1e6:.2f} | {v['achieved_GBs']} | {v['frac_hbm_peak']} | {v['traffic_bytes']} |")
occ = os.path.join(src, "pmc_occupancy_table.txt")
if os.path.exists(occ):
    shutil.copy(occ, os.path.join(dst, f"{tag}_pmc_occupancy_table.txt"))
    out += ["", "Achieved occupancy (`rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE`, its own pass; PMC runs "
            "serialise the kernels, so the concurrent U-step classes are shown alone):", "", open(occ).read().rstrip()]
mix = os.path.join(src, "pmc_mix_table.txt")
if os.path.exists(mix):
    shutil.copy(mix, os.path.join(dst, f"{tag}_pmc_mix_table.txt"))
    out += ["", "Where the wave cycles go (`rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY "
            "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD`, its own pass): every kernel waits on memory for 43-87 % of its wave cycles",
            "", open(mix).read().rstrip()]
extra = f"{dst}/{tag}_extra.md"          # hand-written notes on runs outside bench.py (kept across regenerations)
if os.path.exists(extra):
    out += ["", open(extra).read().rstrip()]
open(f"{dst}/{tag}_summary.md", "w").write("\n".join(out) + "\n")
print("\n".join(out[:12]))
