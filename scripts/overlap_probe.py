#!/usr/bin/env python
"""What the shell / interior split of the dense sweep costs on one GPU: an n^3 brick with one ghost
oct per side (the MPI-resident layout), fast and strict build: ms of the whole sweep, of the shell launch, of the interior
launch, of the periodic self-fill standing in for the exchange, and of the overlapped schedule (shell -> {fill on a
second stream || interior}).   python scripts/overlap_probe.py [n]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import ramses_amd  # noqa: E402
from ramses_amd.hydro import HydroLevel  # noqa: E402


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    for fast in (1, 0):
        p = ramses_amd.make_params(courant_factor=0.8, fast_math=bool(fast))
        lev = HydroLevel(n, n, n, 0.5 / n, params=p, ng=2)
        lev.uold[0].fill_(1.0)
        lev.uold[4].fill_(2.5e-5)
        lev.uold[4, n // 2, n // 2, n // 2] = 1e3
        lev.make_virtual_fine_dp()
        dt = lev.courant_fine()[0]
        for _ in range(40):          # clock ramp
            lev.godunov_fine(dt)
        side = torch.cuda.Stream()

        def overlapped():
            comp = torch.cuda.current_stream()
            lev.godunov_fine_shell(dt)
            e1 = torch.cuda.Event(); e1.record(comp)
            with torch.cuda.stream(side):
                side.wait_event(e1)
                lev.uold, lev.unew = lev.unew, lev.uold
                lev.make_virtual_fine_dp()
                lev.uold, lev.unew = lev.unew, lev.uold
                e2 = torch.cuda.Event(); e2.record(side)
            lev.godunov_fine_interior(dt)
            comp.wait_event(e2)

        def serial():
            lev.godunov_fine(dt)
            lev.uold, lev.unew = lev.unew, lev.uold
            lev.make_virtual_fine_dp()
            lev.uold, lev.unew = lev.unew, lev.uold

        out = {"n": n, "build": "fast" if fast else "strict",
               "sweep_ms": timed(lambda: lev.godunov_fine(dt)),
               "shell_ms": timed(lambda: lev.godunov_fine_shell(dt)),
               "interior_ms": timed(lambda: lev.godunov_fine_interior(dt)),
               "self_fill_ms": timed(lev.make_virtual_fine_dp),
               "serial_step_ms": timed(serial), "overlapped_step_ms": timed(overlapped)}
        print(json.dumps(out), flush=True)
        del lev
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
