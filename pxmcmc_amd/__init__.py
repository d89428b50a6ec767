"""
pxmcmc_amd -- MI355X (gfx950) implementation of pxmcmc's proximal-Langevin hot path
(MYULA / PxMALA) behind the reference's ForwardOperator / Prior / PxMCMCParams plugin API.
All compute runs in hand-written HIP kernels reached through the C-ABI in
include/pxmcmc_amd.h; there is no CPU fallback.
"""
__version__ = "0.1.0"
