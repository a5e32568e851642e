#!/bin/bash
# round 6: the RCCL -> peer-to-peer fall-back on a layout RCCL refuses (two ranks on one device), and the N > 1 tests once more
set -o pipefail
python -m pytest tests/test_bench_entry.py -x -q -m gpu > gpurun_out/r06_m_tests.$(date +%s).log 2>&1
rc=$?; tail -12 gpurun_out/r06_m_tests.*.log | cut -c1-300; exit $rc
