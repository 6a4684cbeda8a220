"""GPU tests of the halo slab movers (pack / unpack / periodic fill) through
the C ABI, against the torch-slicing restatement in test_halo_gloo.py."""
import ctypes as C

import numpy as np
import pytest

from test_halo_gloo import slab_slices

pytestmark = pytest.mark.gpu


def _level(n, nvar=5, ng=2):
    import torch
    import ramses_amd
    from ramses_amd.hydro import HydroLevel
    lev = HydroLevel(n[0], n[1], n[2], 1.0 / n[0], params=ramses_amd.make_params(), ng=ng)
    g = torch.Generator(device="cpu").manual_seed(7)
    lev.uold.copy_(torch.rand(lev.uold.shape, generator=g, dtype=torch.float64))
    return lev


@pytest.mark.parametrize("n", [(8, 6, 10), (70, 12, 6), (16, 16, 16)])
def test_pack_unpack_match_slicing(gpu_lib, n):
    import torch
    from ramses_amd.parallel import BrickDecomposition
    lev = _level(n)
    dec = BrickDecomposition((1, 1, 1), 0, n[0])
    ref = lev.uold.cpu()
    for face in range(6):
        size = dec._slab_size(lev, lev.nvar, face)
        s = slab_slices(n, 2, face, False)
        assert size == ref[(slice(None),) + s].numel()
        buf = torch.empty(size, dtype=torch.float64, device="cuda")
        dec._pack(lev, lev.uold, lev.nvar, face, buf)
        torch.cuda.synchronize()
        assert torch.equal(buf.cpu(), ref[(slice(None),) + s].reshape(-1))
        # unpack a marker buffer into the ghost slab and check only that slab changed
        mark = torch.arange(size, dtype=torch.float64, device="cuda") + 1000.0
        before = lev.uold.clone()
        dec._unpack(lev, lev.uold, lev.nvar, face, mark)
        torch.cuda.synchronize()
        g = slab_slices(n, 2, face, True)
        after = lev.uold.cpu()
        exp = before.cpu()
        exp[(slice(None),) + g] = mark.cpu().reshape(exp[(slice(None),) + g].shape)
        assert torch.equal(after, exp)
        lev.uold.copy_(before)


def test_periodic_fill_matches_global_wrap(gpu_lib):
    import torch
    n = (20, 10, 12)
    lev = _level(n)
    inner = lev.interior(lev.uold).cpu().numpy()
    lev.make_virtual_fine_dp()
    torch.cuda.synchronize()
    got = lev.uold.cpu().numpy()
    iz = np.arange(-2, n[2] + 2) % n[2]
    iy = np.arange(-2, n[1] + 2) % n[1]
    ix = np.arange(-2, n[0] + 2) % n[0]
    exp = inner[:, iz][:, :, iy][:, :, :, ix]
    assert np.array_equal(got, exp)
