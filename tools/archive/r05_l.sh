#!/bin/bash
# (1) the driver's launch mode (torch.distributed.run) + the five-rank launcher test; (2) rocprofv3 kernel stats of the drop-in CLI on the
# Netflix-shaped directory, -t 10: the set-up kernels (k_plan_*, rocPRIM's sort, k_gain_from_levels, k_mat_in) beside the iteration's
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_bench_entry.py -m gpu -x -q -k "drivers_launcher or five_ranks" > gpurun_out/r05_l_tests.log 2>&1
rc=$?; tail -4 gpurun_out/r05_l_tests.log; [ $rc -eq 0 ] || exit $rc
python - <<'PY'
import sys
sys.path.insert(0, ".")
from primalcr_amd import synth
synth.write_dir(synth.generate_fast("netflix"), "/tmp/pcr_nf")
PY
export TMPDIR=/tmp
R=$PWD
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05_l_trace -- $R/primalcr_amd/bin/omp-pmf-train -k 100 -t 10 -n 16 --timing /tmp/pcr_nf /tmp/nf.model > $R/gpurun_out/r05_l_cli.out 2> $R/gpurun_out/r05_l_cli.err
rc=$?
grep timing $R/gpurun_out/r05_l_cli.err
cp $(ls $R/gpurun_out/r05_l_trace/*/*kernel_stats.csv | head -1) $R/gpurun_out/r05_l_cli_kernel_stats.csv
rm -rf $R/gpurun_out/r05_l_trace
head -30 $R/gpurun_out/r05_l_cli_kernel_stats.csv | cut -c1-150
exit $rc
