#!/bin/bash
# developer check: the device-built SpMM plan against the host-built one (-DPCR_PLAN_CHECK library) on every solver the in-process
# GPU tests create, then the normal suite with the shipped library and the set-up phases on the Netflix shape
set -o pipefail
mkdir -p gpurun_out
PCR_SANITIZED_DIR=$PWD/build_next/check python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_system.py -m gpu -x -q > gpurun_out/r05_e_plancheck.log 2>&1
rc=$?
grep -c "device plan == host plan" gpurun_out/r05_e_plancheck.log; grep "MISMATCH\|differs" gpurun_out/r05_e_plancheck.log | head; tail -5 gpurun_out/r05_e_plancheck.log
[ $rc -eq 0 ] || exit $rc
python -m pytest tests -m gpu -x -q > gpurun_out/r05_e_gputests.log 2>&1
rc=$?
tail -5 gpurun_out/r05_e_gputests.log
[ $rc -eq 0 ] || exit $rc
bash tools/r05_c.sh
