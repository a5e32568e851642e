#!/bin/bash
# round 6, first GPU call: what the memory-side TCC counters count (tools/ubench/dram_calib.hip), the -m gpu suite, the driver's bench
set -o pipefail
REPO=$PWD
OUT=$REPO/gpurun_out/r06_calib
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > "$REPO/gpurun_out/counters_list.txt" 2>&1 || true
for f in cwsr_enable sched_policy hws_max_conc_proc num_kcq max_num_of_queues_per_device mes queue_preemption_timeout_ms; do
    echo "$f = $(cat /sys/module/amdgpu/parameters/$f 2>&1)"; done > "$REPO/gpurun_out/r06_kfd_params.txt"
rocminfo > "$REPO/gpurun_out/r06_rocminfo.txt" 2>&1 || true
i=0
for C in "TCC_EA0_RDREQ_DRAM_sum" "TCC_EA0_WRREQ_DRAM_sum" "TCC_EA0_RDREQ_DRAM_32B_sum" "TCC_EA0_WRREQ_WRITE_DRAM_32B_sum" \
         "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_BUBBLE_sum" \
         "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
         "TCC_EA0_RDREQ_GMI_32B_sum TCC_EA0_RDREQ_IO_32B_sum"; do
    i=$((i + 1))
    rocprofv3 --pmc $C --output-format csv -d "$OUT/p$i" -- "$REPO/tools/ubench/_build/dram_calib" > "$OUT/p$i.out" 2> "$OUT/p$i.err" \
        || { echo "pass $i ($C) failed"; tail -3 "$OUT/p$i.err"; }
done
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- "$REPO/tools/ubench/_build/dram_calib" > "$OUT/trace.out" 2> "$OUT/trace.err"
python3 "$REPO/tools/pmc_dram_calib.py" "$OUT" $(ls "$OUT"/trace/*/*kernel_trace.csv | head -1) > "$REPO/gpurun_out/r06_dram_calib.md" || exit 1
cat "$REPO/gpurun_out/r06_dram_calib.md"
cd "$REPO"
python -m pytest tests -m gpu -x -q > gpurun_out/r06_a_gputests.$(date +%s).log 2>&1
rc=$?; tail -4 gpurun_out/r06_a_gputests.*.log; [ $rc -eq 0 ] || exit $rc
python bench.py --full-record gpurun_out/r06_a_bench_full.json > gpurun_out/r06_a_line.json 2> gpurun_out/r06_a_bench.err || { tail -20 gpurun_out/r06_a_bench.err; exit 1; }
tail -3 gpurun_out/r06_a_bench.err; wc -c gpurun_out/r06_a_line.json
