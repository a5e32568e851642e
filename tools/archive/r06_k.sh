#!/bin/bash
# round 6: the whole -m gpu suite on the final build, the driver's bench command, and the CLI's start-up with / without the optional
# kernel families in the code object (build_next/noopt: -DPCR_NO_OPTIONAL_KERNELS, a measurement build)
set -o pipefail
mkdir -p gpurun_out/r06_k
R=$PWD
python -m pytest tests -x -q -m gpu > gpurun_out/r06_k_gputests.$(date +%s).log 2>&1
rc=$?; tail -6 gpurun_out/r06_k_gputests.*.log | cut -c1-300; [ $rc -eq 0 ] || exit $rc
python bench.py --full-record gpurun_out/r06_k_bench_full.json > gpurun_out/r06_k_line.json 2> gpurun_out/r06_k_bench.err || { tail -20 gpurun_out/r06_k_bench.err; exit 1; }
wc -c gpurun_out/r06_k_line.json; python -c "
import json; l=json.load(open('gpurun_out/r06_k_line.json')); print(l['ms_per_step'], l['value'], l['roofline']['frac'], l['hbm'], l['f64']['ms_per_step'], l['netflix']['ms_per_step'], l['cli'])"
d=$(mktemp -d /tmp/cli_XXXX); python -c "
import sys; sys.path.insert(0, '$R')
from primalcr_amd import synth
synth.write_dir(synth.generate('ml1m', seed=synth.SEED), '$d/data')"
for i in 1 2 3 4 5; do
  (cd $d && /usr/bin/time -f "full code object: wall %e s" $R/primalcr_amd/bin/omp-pmf-train -k 100 -l 5000 -t 10 -n 16 --timing data m.model 2>&1 | grep -E "timing-create|wall" | cut -c1-120)
  (cd $d && LD_PRELOAD=$R/build_next/noopt/libprimalcr.so /usr/bin/time -f "without the optional kernels: wall %e s" $R/primalcr_amd/bin/omp-pmf-train -k 100 -l 5000 -t 10 -n 16 --timing data m.model 2>&1 | grep -E "timing-create|wall" | cut -c1-120)
done | tee gpurun_out/r06_k/cli_code_object_ab.txt
