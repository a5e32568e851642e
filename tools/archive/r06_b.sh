#!/bin/bash
# round 6, second GPU call: (1) kernels that wait for each other across processes against the number of hardware queues the
# processes hold (tools/ubench/queue_budget_probe.hip), (2) the memory controllers' activity (mem_busy_percent) on known patterns
# (tools/ubench/dram_calib loop ..., tools/umc_sample.py), (3) both 32-byte-unit DRAM counters in ONE rocprofv3 pass, (4) the new tests
set -o pipefail
REPO=$PWD
mkdir -p gpurun_out/r06_b
export TMPDIR=/tmp
ls /sys/class/drm/ > gpurun_out/r06_b/drm_cards.txt 2>&1
rocm-smi --showmemuse --showuse > gpurun_out/r06_b/rocm_smi.txt 2>&1 || true
timeout -k 10 240 tools/ubench/_build/queue_budget_probe 4 1,2,4,6,8,12 5 > gpurun_out/r06_b/queue_budget_4proc.txt 2>&1; echo "probe rc $?"
cat gpurun_out/r06_b/queue_budget_4proc.txt
timeout -k 10 120 tools/ubench/_build/queue_budget_probe 2 1,4,8,12 5 > gpurun_out/r06_b/queue_budget_2proc.txt 2>&1; echo "probe rc $?"
cat gpurun_out/r06_b/queue_budget_2proc.txt
for P in idle read write copy reread reread192 gather1 gather7 gather109 gather1024; do
    python3 tools/umc_sample.py --out gpurun_out/r06_b/umc_calib.jsonl -- tools/ubench/_build/dram_calib loop $P 3 || exit 1
done
cd /tmp
rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum --output-format csv -d "$REPO/gpurun_out/r06_b/both" -- "$REPO/tools/ubench/_build/dram_calib" > /dev/null 2> "$REPO/gpurun_out/r06_b/both.err" \
    && python3 "$REPO/tools/pmc_dram_calib.py" "$REPO/gpurun_out/r06_b" > "$REPO/gpurun_out/r06_b/both_in_one_pass.md"
head -8 "$REPO/gpurun_out/r06_b/both_in_one_pass.md"
rm -rf "$REPO/gpurun_out/r06_b/both"
cd "$REPO"
python -m pytest tests/test_gpu_parity.py tests/test_cli.py -x -q -m gpu -k "pair_rated_twice or side_file or blocked_user" > gpurun_out/r06_b_tests.$(date +%s).log 2>&1
rc=$?; tail -4 gpurun_out/r06_b_tests.*.log; exit $rc
