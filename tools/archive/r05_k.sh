#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r05_k_gputests.log 2>&1
rc=$?
tail -5 gpurun_out/r05_k_gputests.log
exit $rc
