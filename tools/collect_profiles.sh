#!/bin/bash
# Dev tool (GPU box): everything profiles/<tag>_<shape>_* is built from, in one gpurun call.
#   bash tools/collect_profiles.sh <tag> <shape> <precision> [passes] [extra bench.py args...]
#       e.g. r06_e ml1m f32 "trace dram l2 lds occ mix"   (dram = the byte-exact 32-byte-unit counters of round 6; fetch / write = FETCH_SIZE / WRITE_SIZE)        r03_a netflix f32 "trace fetch write l2 lds" --steps 3 --warmup 1
# Writes gpurun_out/<tag>/<shape>_<precision>/: kernel-trace stats CSV and one rocprofv3 --pmc pass per counter group (separate runs,
# no trace domain beside --kernel-trace; the program directly behind "--").
set -o pipefail
TAG=${1:-r03_a}; SHAPE=${2:-ml1m}; PREC=${3:-f32}; PASSES=${4:-"trace fetch write l2 lds occ mix"}
shift 4 2>/dev/null
EXTRA="$@"
OUT=$PWD/gpurun_out/$TAG/${SHAPE}_$PREC
REPO=$PWD
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
B="python3 $REPO/bench.py --no-live-traffic --no-hbm --full-line --full-record /tmp/bench_full_prof.json --shape $SHAPE --precision $PREC --no-cpu --no-cli --no-f64 --no-netflix --no-rows $EXTRA"
declare -A PMC=( [dram]="TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum" [fetch]="FETCH_SIZE" [write]="WRITE_SIZE" [l2]="TCC_HIT_sum TCC_MISS_sum" [lds]="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES"
                 [occ]="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
                 [mix]="SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" )
for P in $PASSES; do
    echo "[$SHAPE/$PREC] pass $P"
    if [ "$P" = trace ]; then
        rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $B > "$OUT/bench_under_trace.json" 2> "$OUT/trace.err" || { tail -5 "$OUT/trace.err"; exit 1; }
        cp $(ls "$OUT"/trace/*/*kernel_stats.csv | head -1) "$OUT/kernel_stats.csv"
        rm -rf "$OUT"/trace/*/*kernel_trace.csv "$OUT"/trace/*/*agent_info.csv      # large
    else
        rocprofv3 --pmc ${PMC[$P]} --output-format csv -d "$OUT/pmc_$P" -- $B --no-profile > /dev/null 2> "$OUT/pmc_$P.err" || { tail -5 "$OUT/pmc_$P.err"; exit 1; }
        python3 "$REPO/tools/pmc_table.py" "$OUT/pmc_$P" > "$OUT/pmc_$P.json" || exit 1
        [ "$P" = occ ] && { python3 "$REPO/tools/pmc_occupancy.py" "$OUT/pmc_occ" "$OUT/pmc_occupancy_table.txt" > /dev/null || exit 1; }
        [ "$P" = mix ] && { python3 "$REPO/tools/pmc_mix.py" "$OUT/pmc_mix" > "$OUT/pmc_mix_table.txt" || exit 1; }
        rm -rf "$OUT/pmc_$P"                                                             # the per-dispatch CSV is large
    fi
done
ls -la "$OUT"
