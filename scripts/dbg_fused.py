import sys, ctypes as C; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
import ramses_amd
from oracle import pyoracle as po
L = ramses_amd.lib(); OL = po.lib()
def dev(a): return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64).cuda()
def p(t): return C.c_void_p(t.data_ptr())
for n,npass in [(64,2),(64,4)]:
    rng = np.random.default_rng(1)
    phi = rng.normal(size=(n,n,n)); rhs = rng.normal(size=(n,n,n)); dx=1/64
    ref = phi.copy()
    for q in range(npass): OL.ora_mg_gauss_seidel(ref, rhs, n, dx*dx, 1 if q%2==0 else 0)
    din, dout = dev(phi), dev(np.zeros_like(phi))
    rc = L.ramses_amd_mg_smooth_fused(p(din), p(dout), p(dev(rhs)), None, None, None, n, dx, npass, None)
    torch.cuda.synchronize()
    out = dout.cpu().numpy()
    bad = out != ref
    print(n, npass, 'rc', rc, 'nbad', bad.sum(), 'of', bad.size)
    if bad.any():
        k,j,i = np.nonzero(bad)
        print(' z range', k.min(), k.max(), 'y', j.min(), j.max(), 'x', i.min(), i.max())
        print(' bad per z (first 10):', [int(bad[z].sum()) for z in range(10)])
        print(' bad per y:', [int(bad[:,y].sum()) for y in range(0,64,4)])
        print(' bad per x:', [int(bad[:,:,x].sum()) for x in range(0,64,4)])
        print(' parity of bad cells:', np.bincount((k+j+i)%2))
        print(' zero outputs:', (out==0).sum())
