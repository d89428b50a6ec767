"""
oracle/ -- CPU restatement of the reference's proximal-Langevin hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
anything under ``oracle/``; the product package ``pxmcmc_amd`` never does.

What it restates (file:line are relative to the reference tree, pxmcmc v1.0.1):

* ``oracle.pxmcmc_np``  -- the pure-numpy half of the path: ``soft/_sign``
  (pxmcmc/utils.py:55-67,84-88), ``MYULA.chain_step`` (pxmcmc/mcmc.py:185-201),
  inverse covariance incl. the complex-variance quirk (pxmcmc/forward.py:74-88),
  ``calc_gradg`` (pxmcmc/forward.py:48-72), ``logpi`` (pxmcmc/mcmc.py:71-82),
  ``calc_logtransition`` / ``_tune_delta`` (pxmcmc/mcmc.py:277-289), the
  ``MYULA.run`` / ``PxMALA.run`` loops (pxmcmc/mcmc.py:150-275), MW quadrature
  weights (pxmcmc/utils.py:249-283), the weak-lensing harmonic kernel
  (pxmcmc/measurements.py:151-171).
  PARITY: PINNED against golden vectors captured from the reference itself
  (tests/golden/*.npz, generator tests/golden/make_golden.py).  The wavelet-path glue
  (SphericalWaveletTransform, WeakLensing, SphericalWaveletTransformOperator,
  S2_Wavelets_L1 and the samplers on them) is pinned by G14: the reference's own
  classes executed over ``oracle.ext_stub`` (tests/golden/make_golden_r5.py).

* ``oracle.ext_stub``  -- the ``pys2let`` / ``pyssht`` names and call shapes the
  reference uses, served by ``oracle.s2let`` / ``oracle.ssht``: what G14's generator
  installs in ``sys.modules`` so that the reference's own code runs in the build
  container.  It pins the reference's glue, not the third-party numerics.

* ``oracle.wigner`` / ``oracle.ssht`` / ``oracle.s2let`` -- the O(L^3) half the
  reference delegates to un-vendored third-party wheels: pyssht 1.5.2
  (poetry.lock:1263-1264) and pys2let 2.2.6 (poetry.lock:1234-1235).  Neither is
  present in this image, so these modules restate the *published* algorithms
  (McEwen & Wiaux 2011 MW sampling theorem; Leistedt et al. 2013 / McEwen et al.
  2015 scale-discretised wavelets) and are pinned by the reference's own
  property tests (round trip, adjoint dot tests, int f = f00 sqrt(4pi); see
  tests/test_transforms.py, tests/test_measurements.py, tests/test_utils.py in
  the reference) plus analytic spin-weighted harmonics and scipy's sph_harm_y.
  PARITY: the absolute wavelet-coefficient scale (sqrt(2pi) convention), the
  kappa_j profile and spin-2 sign conventions versus pys2let/pyssht are
  "parity unpinned" -- no fixture from those libraries exists to check against.
"""
