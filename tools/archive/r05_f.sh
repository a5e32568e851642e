#!/bin/bash
# developer check, with the check lines shown: every solver the in-process GPU tests create builds the SpMM plan twice
# (device: shipped path; host: rounds 1-4's builder) and compares every array
set -o pipefail
mkdir -p gpurun_out
PCR_SANITIZED_DIR=$PWD/build_next/check python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_system.py -m gpu -x -q -s > gpurun_out/r05_f_plancheck.log 2>&1
rc=$?
echo "solvers checked: $(grep -c 'device plan == host plan' gpurun_out/r05_f_plancheck.log); mismatches: $(grep -c 'MISMATCH' gpurun_out/r05_f_plancheck.log)"
grep "plan check" gpurun_out/r05_f_plancheck.log | sed 's/.*\[plan check\]/[plan check]/' | sort | uniq -c | sort -rn | head -40
tail -3 gpurun_out/r05_f_plancheck.log
exit $rc
