#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r05_d_gputests.log 2>&1
rc=$?
tail -5 gpurun_out/r05_d_gputests.log
[ $rc -eq 0 ] || exit $rc
bash tools/r05_c.sh
