import sys; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch, time
from ramses_amd._capi import lib, check
from ramses_amd.poisson import PoissonLevel
def solve(level, tune, eps=1e-6, reps=1):
    check(lib().ramses_amd_mg_tune(tune))
    n = 2 ** level
    lev = PoissonLevel(level, boxlen=1.0, epsilon=eps)
    g = torch.Generator(device="cuda").manual_seed(3)
    lev.rho.copy_(1.0 + torch.rand((n, n, n), generator=g, device="cuda", dtype=torch.float64))
    a, b = int(0.3 * n), int(0.55 * n)
    lev.rho[a:b, a:b, a:b] += 9.0
    rt = float(lev.rho.mean().item())
    it, err = lev.multigrid_fine(rt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        it, err = lev.multigrid_fine(rt)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / reps
    return it, err, lev.phi.clone(), t
for level in (7, 8):
    base = solve(level, 1)
    for tune in (12, 16):
        r = solve(level, tune)
        print("level", level, "tune", tune, "iters", r[0], base[0], "phi equal", bool(torch.equal(r[2], base[2])))
for tune in (1, 12, 16):
    it, err, _, t = solve(9, tune, eps=1e-30, reps=2)
    print("level 9 tune", tune, "ms/vcycle %.3f" % (t / it * 1e3))
check(lib().ramses_amd_mg_tune(1))
