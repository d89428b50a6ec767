import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pxmcmc_amd import ops
from pxmcmc_amd._lib import lib, PxmError, check
def expect(exc, fn, *a, **k):
    try:
        fn(*a, **k)
    except exc as e:
        print("ok:", type(e).__name__, str(e)[:90]); return
    raise SystemExit(f"no {exc} from {fn}")
expect(PxmError, ops.ShtPlan, 0, 0)
expect(PxmError, ops.ShtPlan, 4, 7)
expect(PxmError, ops.WavPlan, 8, 1.0, 1)
w = ops.WavPlan(8, 2.0, 1, max_chains=2)
X = torch.zeros(3, w.ncoefs, dtype=torch.complex128).cuda()
expect(ValueError, w.synthesis, X)                      # more chains than the plan holds
expect(AssertionError, w.synthesis, X[:2, :5])          # wrong length
x2 = X[:2].contiguous()
expect(ValueError, w.ring_step, x2, 1.0, 0.1, 1e-3, 1e-3, out=x2)   # aliasing
expect(PxmError, lambda: check(lib.pxm_wav_ring_step(w._h, C.c_void_p(x2.data_ptr()), 1.0, 0.0, None, 0.1, 1e-3, 1e-3, None, 0, 0, 0, 0, C.c_void_p(X[1:3].data_ptr()), 2, None)))  # ring_set_data missing
w.ring_set_data(torch.zeros(w.npix, dtype=torch.complex128).cuda())
expect(PxmError, lambda: check(lib.pxm_wav_ring_step(w._h, C.c_void_p(x2.data_ptr()), 1.0, 0.0, None, 0.1, 1e-3, 1e-3, None, 7, 0, 0, 0, C.c_void_p(X[1:3].data_ptr()), 1, None)))  # bad mode
expect(PxmError, lambda: check(lib.pxm_soft(None, None, 0.1, None, 10, 1, 0, None)))
import scipy.sparse as sp
from pxmcmc_amd.measurements import PathIntegral, WeakLensing
pi = PathIntegral(sp.csr_matrix((0, 10)))
print("empty path matrix:", pi.forward(np.zeros(10)).shape, pi.adjoint(np.zeros(0)).shape)
expect(ValueError, WeakLensing, 0)
expect(ValueError, WeakLensing, 8, np.ones((3, 3)))
wl = WeakLensing(8, np.zeros((8, 15), dtype=int))   # everything masked
print("all-masked WL ndata:", wl.ndata, wl.forward(np.zeros(8 * 15, dtype=complex)).shape, wl.adjoint(np.zeros(0, dtype=complex)).shape)
print("error paths fine")
