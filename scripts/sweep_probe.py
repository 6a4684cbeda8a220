"""The dense sweep of a uniform periodic level with any solver / slope pair (bench.py times LLF + minmod only):
scripts/sweep_probe.py N RIEMANN SLOPE_TYPE [STEPS] -> ms per sweep of the fast and the strict build (A/B of build variants
with RAMSES_AMD_LIB=...)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import ramses_amd  # noqa: E402
from ramses_amd import ic  # noqa: E402
from ramses_amd.hydro import HydroLevel  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    riemann = sys.argv[2] if len(sys.argv) > 2 else "hllc"
    st = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
    out = []
    for fast in (True, False):
        p = ramses_amd.make_params(courant_factor=0.8, fast_math=fast, riemann=riemann, slope_type=st)
        lev = HydroLevel(n, n, n, 0.5 / n, params=p, ng=0)
        corner, back, dx = ic.sedov3d_corner_and_background(n)
        for v in range(5):
            lev.uold[v].fill_(float(back[v]))
            lev.uold[v, 0, 0, 0] = float(corner[v])
        dt = lev.courant_fine()[0]
        for _ in range(30):
            lev.step(dt)
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(steps):
            lev.step(dt)
        e.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(e) / steps
        out.append("%s %.3f ms (%.1f%%)" % ("fast" if fast else "strict", ms, 100 * n ** 3 * 80 / (ms * 1e-3) / 8e12))
        del lev
    print("sweep_probe %d^3 %s slope %d: %s" % (n, riemann, st, "  ".join(out)))


if __name__ == "__main__":
    main()
