"""Single-GPU probe of the overlapped step: ng=2 brick, the 'exchange' is the periodic self-fill."""
import sys, time
sys.path.insert(0, ".")
import torch
import ramses_amd
from ramses_amd.parallel import BrickDecomposition
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dec = BrickDecomposition((1, 1, 1), 0, n, boxlen=0.5)
lev = dec.make_level(ramses_amd.make_params(courant_factor=0.8, fast_math=True))
dec.init_sedov(lev)
dec.make_virtual_fine_dp(lev)
dt = lev.courant_fine()[0]
for mode in ("plain", "overlap", "plain", "overlap"):
    for it in range(3):
        (dec.step_overlapped(lev, dt) if mode == "overlap" else (lev.godunov_fine(dt), lev.set_uold(), dec.make_virtual_fine_dp(lev)))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for it in range(10):
        (dec.step_overlapped(lev, dt) if mode == "overlap" else (lev.godunov_fine(dt), lev.set_uold(), dec.make_virtual_fine_dp(lev)))
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 10
    print(mode, "ms/step %.3f" % (t * 1e3), "Gcell/s %.2f" % (n ** 3 / t / 1e9))
# pieces: sweep on the ghost brick alone, shell alone, interior alone, self-exchange alone
def timeit(fn, k=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3
print("sweep(ng=2) ms %.3f" % timeit(lambda: lev.godunov_fine(dt)))
print("shell ms %.3f" % timeit(lambda: lev.godunov_fine_shell(dt)))
print("interior ms %.3f" % timeit(lambda: lev.godunov_fine_interior(dt)))
print("self-exchange ms %.3f" % timeit(lambda: dec.make_virtual_fine_dp(lev)))
