#!/bin/bash
# round 5 profile set: kernel-trace stats + PMC passes on the three records of the driver's line
set -o pipefail
bash tools/collect_profiles.sh r05_a ml1m f32 "trace fetch write l2 lds" > gpurun_out/r05_collect_f32.log 2>&1 || { tail -5 gpurun_out/r05_collect_f32.log; exit 1; }
bash tools/collect_profiles.sh r05_a ml1m f64 "trace fetch write l2" > gpurun_out/r05_collect_f64.log 2>&1 || { tail -5 gpurun_out/r05_collect_f64.log; exit 1; }
bash tools/collect_profiles.sh r05_a netflix f32 "trace fetch write l2" --steps 3 --warmup 1 > gpurun_out/r05_collect_nf.log 2>&1 || { tail -5 gpurun_out/r05_collect_nf.log; exit 1; }
tail -3 gpurun_out/r05_collect_nf.log
