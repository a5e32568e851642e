#!/bin/bash
# Dev tool (GPU box): everything profiles/<tag>_* is built from, in one gpurun call.
#   bash tools/collect_profiles.sh <tag> [note on the commit]       e.g. r02_a "HEAD abc1234"
# Writes gpurun_out/<tag>/: kernel-trace stats CSVs, two PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, no trace
# domains beside --kernel-trace), the traffic table, and the bench JSON lines with and without event timing.
set -o pipefail
TAG=${1:-r02_a}
HEAD_NOTE=${2:-"working tree"}
OUT=$PWD/gpurun_out/$TAG
REPO=$PWD
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
echo "[1/5] kernel trace"; rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/bench.py" --no-cpu --no-f64 > "$OUT/bench_under_trace.json" 2> "$OUT/trace.err" || exit 1
echo "[2/5] pmc FETCH_SIZE"; rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$REPO/bench.py" --no-cpu --no-f64 --no-profile --steps 10 > /dev/null 2> "$OUT/pmc_fetch.err" || exit 1
echo "[3/5] pmc WRITE_SIZE"; rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$REPO/bench.py" --no-cpu --no-f64 --no-profile --steps 10 > /dev/null 2> "$OUT/pmc_write.err" || exit 1
echo "[3b] pmc occupancy"; rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_occ" -- python3 "$REPO/bench.py" --no-cpu --no-f64 --no-profile --steps 10 > /dev/null 2> "$OUT/pmc_occ.err" || exit 1
echo "[3c] pmc instruction mix"; rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d "$OUT/pmc_mix" -- python3 "$REPO/bench.py" --no-cpu --no-f64 --no-profile --steps 10 > /dev/null 2> "$OUT/pmc_mix.err" || exit 1
cd "$REPO"
python3 tools/pmc_mix.py "$OUT/pmc_mix" > "$OUT/pmc_mix_table.txt" || exit 1
python3 tools/pmc_occupancy.py "$OUT/pmc_occ" "$OUT/pmc_occupancy_table.txt" > /dev/null || exit 1
python3 tools/pmc_traffic.py "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/traffic.json" "profiles/r02_traffic.json ($TAG, $HEAD_NOTE)" > "$OUT/pmc_traffic_table.txt" || exit 1
cp "$OUT/traffic.json" profiles/r02_traffic.json      # bench.py reads the PMC figures from here
echo "[4/5] bench (event timing, cpu baseline)"; python3 bench.py --verbose > "$OUT/bench.json" 2> "$OUT/bench.err" || exit 1
echo "[5/5] bench (no event timing)"; python3 bench.py --no-cpu --no-f64 --no-profile > "$OUT/bench_noprofile.json" 2>/dev/null || exit 1
cp $(ls "$OUT"/trace/*/*kernel_stats.csv | head -1) "$OUT/kernel_stats.csv"
rm -rf "$OUT"/trace/*/*kernel_trace.csv      # large
ls -la "$OUT"
tail -c 1500 "$OUT/bench.json"
