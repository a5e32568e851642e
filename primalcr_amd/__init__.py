"""primalcr_amd -- MI355X-native PrimalCR / PrimalCR++ collaborative-ranking solver.

The product is ``primalcr_amd/lib/libprimalcr.so`` (hand-written HIP for gfx950 behind the C
ABI of ``include/primalcr.h``) plus the drop-in CLIs ``primalcr_amd/bin/omp-pmf-train`` and
``omp-pmf-predict``.  This package is the thin host-side mirror of the reference's solver
interface (``pmf.h``) over that C ABI via ctypes -- used by tests and ``bench.py``.

There is no CPU fallback: importing works without a GPU (host-side helpers such as the loader
and ``initial`` are usable), but every training entry point raises ``PcrError`` when the HIP
library or a GPU is missing.
"""
from .api import (PCR_F32, PCR_F64, PCR_SOLVER_PCR, PCR_SOLVER_PCRPP, Dataset, Parameter, PcrError, Solver,
                  comm_unique_id, initial, initial_rows, lib, lib_path, use_library, model_load, model_save, partition_users, predict, tune, tuned)

__all__ = ["PCR_F32", "PCR_F64", "PCR_SOLVER_PCR", "PCR_SOLVER_PCRPP", "Dataset", "Parameter", "PcrError", "Solver",
           "comm_unique_id", "initial", "initial_rows", "lib", "lib_path", "use_library", "model_load", "model_save", "partition_users", "predict", "tune", "tuned"]
