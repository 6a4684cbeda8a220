"""ramses_amd -- MI355X-native Godunov hydro sweep, multigrid Poisson smoother
and virtual-boundary halo exchange for RAMSES (reference: tatary/ramses).

Layers:
  csrc/      hand-written HIP kernels for gfx950 + the C ABI (include/ramses_amd.h)
  lib/       libramses_amd.so (built in-tree by ramses_amd.build / __graft_entry__.build)
  patch/     the RAMSES PATCH= directory: Fortran 90 shims (ISO_C_BINDING) that keep
             godunov_fine()/set_unew()/set_uold()/... and call the C ABI
  hydro.py, poisson.py, parallel.py, poisson_parallel.py, amr.py
             Python host mirror of the same interface (tests, bench)
"""
from ._capi import RamsesAmdError, make_params, lib, LIB_PATH  # noqa: F401
