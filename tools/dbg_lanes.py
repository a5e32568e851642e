"""Dev tool (GPU box): what pick_lanes measures (queue sharing, command-processor pipe sharing) with N foreign streams created by the
host application (PyTorch) before the solver.  usage: dbg_lanes.py <N>   (torch first: one HIP runtime per process)"""
import sys; sys.path.insert(0, '.')
import torch
import primalcr_amd as pcr
from primalcr_amd import synth
R = synth.generate("small", seed=2)
ds = pcr.Dataset.from_ratings(R)
extra = [torch.cuda.Stream() for _ in range(int(sys.argv[1]))]
for e in extra:
    with torch.cuda.stream(e): torch.zeros(4, device="cuda").add_(1)
torch.cuda.synchronize()
with pcr.tuned(debug=1):
    s = pcr.Solver(ds, pcr.Parameter(k=16))
s.close()
