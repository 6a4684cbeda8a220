"""The MHD sweep alone (for rocprofv3 / A-B runs on the GPU box): scripts/mhd_probe.py LEVEL [STEPS] [RIEMANN] [RIEMANN2D]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ramses_amd.mhd import MhdLevel, make_mhd_params  # noqa: E402


def main():
    level = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    r1 = sys.argv[3] if len(sys.argv) > 3 else "hlld"
    r2 = sys.argv[4] if len(sys.argv) > 4 else "hlld"
    n = 2 ** level
    lev = MhdLevel(n, n, n, 1.0 / n, params=make_mhd_params(gamma=5.0 / 3.0, slope_type=2, riemann=r1, riemann2d=r2))
    x = (torch.arange(n, dtype=torch.float64, device="cuda") + 0.5) / n
    Z, Y, X = torch.meshgrid(x, x, x, indexing="ij")
    r2_ = (X - 0.5) ** 2 + (Y - 0.5) ** 2 + (Z - 0.5) ** 2
    b = (1.0, 0.5, -0.3)
    lev.uold[0].fill_(1.0)
    for c in range(3):
        lev.uold[5 + c].fill_(b[c])
        lev.uold[8 + c].fill_(b[c])
    lev.uold[4] = (0.1 + 10.0 * torch.exp(-r2_ / (2 * 0.05 ** 2))) / (5.0 / 3.0 - 1.0) + 0.5 * sum(v * v for v in b)
    del X, Y, Z, r2_
    dt = 0.2 / n / 5.0
    for _ in range(2):
        lev.step(dt)
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        lev.step(dt)
    e.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(e) / steps
    cs = float(lev.uold.double().sum().item())
    print("mhd_probe level %d %s/%s %s: %.4f ms per sweep, %.3f G cells/s, %.4f of 8 TB/s at 176 B/cell, sum %.17g"
          % (level, r1, r2, os.environ.get("RAMSES_AMD_MHD_VARIANT", "default"), ms, n ** 3 / ms / 1e6, n ** 3 * 176 / (ms * 1e-3) / 8e12, cs))


if __name__ == "__main__":
    main()
