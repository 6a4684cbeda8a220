"""Three V-cycles at 512^3 (for counter passes of the fused smoother): scripts/mgtune_probe9.py [TUNE]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ramses_amd._capi import lib, check  # noqa: E402
from ramses_amd.poisson import PoissonLevel  # noqa: E402

tune = int(sys.argv[1]) if len(sys.argv) > 1 else 1
check(lib().ramses_amd_mg_tune(tune))
level = 9
n = 2 ** level
lev = PoissonLevel(level, boxlen=1.0, epsilon=1e-3)
g = torch.Generator(device="cuda").manual_seed(3)
lev.rho.copy_(1.0 + torch.rand((n, n, n), generator=g, device="cuda", dtype=torch.float64))
it, err = lev.multigrid_fine(float(lev.rho.mean().item()))
torch.cuda.synchronize()
print("mgtune_probe9: tune", tune, "iterations", it, "error", err)
