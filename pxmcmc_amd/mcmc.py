"""
Samplers of the reference (pxmcmc/mcmc.py:6-289: PxMCMCParams, PxMCMC, MYULA, PxMALA) driving
the HIP kernels.  The plugin protocol is the reference's: the sampler only touches
``forward.forward / calc_gradg / data / invcov / nparams`` and ``prior.proxf / prior.prior``.

Extensions (all optional, defaults reproduce the reference's one-chain behaviour):

* ``nchains``      -- C independent chains advanced together as one ``[C, N]`` batch;
* ``rng``          -- ``"philox"`` (device counter-based stream keyed (seed, chain, iteration),
  independent of how chains are spread over GPUs) or ``"numpy"`` (the reference's global
  ``np.random`` stream in the reference's draw order, for parity runs);
* ``chain_offset`` -- global index of this process's first chain (multi-GPU sharding);
* ``use_graph``    -- replay the fused iteration from a captured HIP graph (default on);
* ``ring_shortcut`` -- with a scalar ``sig_d`` apply the residual on the ring transforms (default on);
* ``noise_bits``   -- 64 (default) or 32: arithmetic of the Box-Muller step of the device Philox stream.  64 evaluates
  log / sqrt / sincos in double precision (table look-ups + short polynomials, csrc/philox.h) as the reference's
  ``np.random.randn`` does (pxmcmc/mcmc.py:193); 32 runs it on the f32 transcendental units (deviates ~1e-6 relative,
  exact fp64 exponent: tail to 8.5 sigma), ~2 % faster at the benchmark size.  Same Philox counters and uniforms either
  way: the two streams agree to ~1e-6;
* ``real_pairs``   -- with REAL data, a real start point and ``params.complex == False`` the reference's
  complex128 state has a zero imaginary part (every operator of the path maps real fields to real
  fields); the fused wavelet path then carries two real chains per complex slot -- chain 2c in the real
  part, chain 2c+1 in the imaginary part -- through the complex-linear transforms and applies prox and
  noise per component (real-signal symmetry).  Same results, half the transform
  work (default on; complex data, as in the reference-literal topography set-up, is never paired).

SKROCK (pxmcmc/mcmc.py:292-383) is out of scope.
"""
import time

import numpy as np
import torch
from scipy.stats import laplace

from . import ops
from .measurements import Identity
from .prior import L1
from .transforms import SphericalWaveletTransform


class PxMCMCParams:
    """
    Tuning and runtime parameters (pxmcmc/mcmc.py:6-43).

    :param lmda: prox parameter
    :param delta: Forward-Euler step size
    :param mu: regularisation parameter
    :param s: max order of Chebyshev polynomials (SKROCK; unused here)
    :param nsamples: number of samples to save
    :param nburn: burn-in size
    :param ngap: thinning: iterations between saved samples
    :param complex: ``True`` if the sampled parameters are complex
    :param verbosity: print every ``verbosity`` iterations
    :param track: list of variables to keep track of
    """

    def __init__(
        self,
        lmda=3e-5,
        delta=1e-5,
        s=1,
        mu=1,
        nsamples=int(1e6),
        nburn=int(1e3),
        ngap=int(1e2),
        complex=False,
        verbosity=100,
        track=["logposterior", "L2", "prior", "chain"],
    ):
        self.lmda = lmda
        self.delta = delta
        self.mu = mu
        self.s = s
        self.nsamples = nsamples
        self.nburn = nburn
        self.ngap = ngap
        self.complex = complex
        self.verbosity = verbosity
        self.track = track


def _is_stock_l1(prior):
    """True when prior.proxf is the library's own synthesis soft threshold (safe to fuse)."""
    return isinstance(prior, L1) and type(prior).proxf is L1.proxf and type(prior)._proxf_synthesis is L1._proxf_synthesis and prior.setting == "synthesis"


class PxMCMC:
    """
    Base class with the general functions (pxmcmc/mcmc.py:46-140).

    :param forward: :class:`forward.ForwardOperator`-like object
    :param prior: prior object implementing ``prior`` and ``proxf``
    :param mcmcparams: :class:`PxMCMCParams`
    """

    def __init__(self, forward, prior, mcmcparams=PxMCMCParams(), nchains=1, rng="philox", seed=0, chain_offset=0,
                 use_graph=True, ring_shortcut=True, real_pairs=True, noise_bits=64):
        self.forward = forward
        self.prior = prior
        for attr in mcmcparams.__dict__.keys():
            setattr(self, attr, getattr(mcmcparams, attr))
        if rng not in ("philox", "numpy"):
            raise ValueError("rng must be 'philox' or 'numpy'")
        self.nchains = int(nchains)
        self.rng = rng
        self.seed = int(seed)
        self.chain_offset = int(chain_offset)
        self.use_graph = bool(use_graph)
        self.ring_shortcut = bool(ring_shortcut)
        self.real_pairs = bool(real_pairs)
        if noise_bits not in (32, 64):
            raise ValueError("noise_bits must be 32 or 64")
        self.noise_bits = int(noise_bits)
        self.noise64 = self.noise_bits == 64
        self._pairs = False
        self.nsamples = int(self.nsamples)
        for op in (getattr(forward, "transform", None), getattr(forward, "measurement", None)):
            if hasattr(op, "ensure_chains"):
                op.ensure_chains(self.nchains)
        self._initialise_tracking_arrays()

    def run(self, start_point=None):
        raise NotImplementedError

    # ---- device-side pieces -----------------------------------------------------------
    def _state_dtype(self, start=None):
        cplx = bool(self.complex) or isinstance(getattr(self.forward, "transform", None), SphericalWaveletTransform)
        d = self.forward.data
        cplx = cplx or (d.is_complex() if isinstance(d, torch.Tensor) else np.iscomplexobj(d))
        if start is not None:
            cplx = cplx or (start.is_complex() if isinstance(start, torch.Tensor) else np.iscomplexobj(start))
        return torch.complex128 if cplx else torch.float64

    def _logpi_dev(self, X, preds):
        """per-chain (logPi, L2, prior) tensors; pxmcmc/mcmc.py:71-82 (L2 carries no factor 1/2)."""
        p = ops.as_device(preds)
        dt = self.forward._resid_dtype(p) if hasattr(self.forward, "_resid_dtype") else p.dtype
        data = ops.as_device(self.forward.data).reshape(-1).to(dt)
        if hasattr(self.forward.invcov, "matvec"):  # full inverse covariance (forward.py:75-78): vdot(d, invcov @ d)
            d = ops.residual_grad(p.to(dt), data, self.forward.invcov.ones)  # preds - data: vdot(-d, W(-d)) = vdot(d, W d)
            L2 = ops.reduce_vdot(d, self.forward.invcov.matvec(d))
        else:
            invcov = self.forward.invcov.diag if hasattr(self.forward.invcov, "diag") else ops.as_device(self.forward.invcov.diagonal())
            L2 = ops.reduce_l2(p.to(dt), data, invcov)
        prior = self.prior.prior(X)
        if not isinstance(prior, torch.Tensor):
            prior = torch.as_tensor(np.atleast_1d(np.asarray(prior, dtype=float)), device=L2.device)
        logPi = -self.mu * prior - L2
        return logPi, L2, prior

    def logpi(self, X, preds):
        """log posterior, L2 norm and prior norm of a model (pxmcmc/mcmc.py:71-82)."""
        logPi, L2, prior = self._logpi_dev(X, preds)
        batched = (X.dim() if isinstance(X, torch.Tensor) else np.ndim(X)) == 2
        if batched:
            return logPi, L2, prior
        cplx = L2.is_complex() and bool(abs(L2[0].imag.item()) > 0)
        f = (lambda v: complex(v[0].item())) if cplx else (lambda v: float(v[0].real.item()))
        return f(logPi), f(L2), float(prior[0].item())

    def _print_progress(self, i, logpi, **kwargs):
        print(
            f"{i+1:,}/{self.nsamples:,} - logposterior: {logpi:.8e} - "
            + " - ".join([f"{k}: {kwargs[k]:.8e}" for k in kwargs]),
        )

    def _initial_sample(self, initial_sample=None):
        """pxmcmc/mcmc.py:97-111, returning GPU tensors [C, nparams], [C, ndata]."""
        C, N = self.nchains, self.forward.nparams
        if initial_sample is None:
            if self.rng == "numpy":
                draw = lambda: np.stack([laplace.rvs(size=N) for _ in range(C)])
            else:
                gens = [np.random.default_rng([self.seed, self.chain_offset + c, 0x1A91ACE]) for c in range(C)]
                draw = lambda: np.stack([laplace.rvs(size=N, random_state=g) for g in gens])
            X0 = draw()
            if self.complex:
                X0 = X0 + draw() * 1j
        else:
            if isinstance(initial_sample, torch.Tensor):
                X0 = initial_sample
            elif isinstance(initial_sample, np.ndarray):
                X0 = initial_sample
            else:
                raise TypeError("Expected a 1D numpy array as an initial sample")
            nd = X0.dim() if isinstance(X0, torch.Tensor) else np.ndim(X0)
            if nd == 1:
                if X0.shape[0] != N:
                    raise ValueError("Inital sample given has incorrect size")
                X0 = X0[None, :]
                if C > 1:
                    X0 = X0.repeat(C, 1) if isinstance(X0, torch.Tensor) else np.repeat(X0, C, axis=0)
            elif nd == 2 and C > 1:
                if tuple(X0.shape) != (C, N):
                    raise ValueError("Inital sample given has incorrect size")
            else:
                raise TypeError("Expected a 1D numpy array as an initial sample")
        X_curr = ops.as_device(X0, self._state_dtype(X0)).clone()  # never write into the caller's start point
        curr_preds = ops.as_device(self.forward.forward(X_curr))
        return X_curr, curr_preds

    def _initialise_tracking_arrays(self):
        """pxmcmc/mcmc.py:113-128; with nchains > 1 every array gains a leading chain axis."""
        lead = () if self.nchains == 1 else (self.nchains,)
        if "logposterior" in self.track:
            self.logPi = np.zeros(lead + (self.nsamples,))
        if "predictions" in self.track:
            self.preds = np.zeros(lead + (self.nsamples, len(self.forward.data)), dtype=float)
        if "chain" in self.track:
            self.chain = np.zeros(lead + (self.nsamples, self.forward.nparams), dtype=complex if self.complex else float)
        if "L2" in self.track:
            self.L2s = np.zeros(lead + (self.nsamples,), dtype=float)
        if "prior" in self.track:
            self.priors = np.zeros(lead + (self.nsamples,), dtype=float)

    def _tracking(self, j, X_curr, curr_preds, logPi, L2, prior, chains=None):
        """pxmcmc/mcmc.py:130-140.  ``j`` is an int (all chains) or per-chain indices with ``chains``."""
        def put(arr, val):
            val = val.detach().cpu().numpy() if isinstance(val, torch.Tensor) else np.asarray(val)
            if not np.iscomplexobj(arr):
                val = np.real(val)  # the reference's float arrays silently drop the imaginary part
            if self.nchains == 1:
                arr[j] = val[0]
            elif chains is None:
                arr[:, j] = val
            else:
                for c, jc in zip(chains, j):
                    arr[c, jc] = val[c]

        if hasattr(self, "logPi"):
            put(self.logPi, logPi)
        if hasattr(self, "L2s"):
            put(self.L2s, L2)
        if hasattr(self, "priors"):
            put(self.priors, prior)
        if hasattr(self, "preds"):
            put(self.preds, curr_preds)
        if hasattr(self, "chain"):
            put(self.chain, X_curr)

    # ---- device status -----------------------------------------------------------------
    def _device_plans(self):
        """every live transform plan of this process (``ops.live_plans``: a weak registry filled at plan creation) --
        the operators' own plans AND those owned by a prior built on another transform object, by user operators or by
        analysis-setting transforms under any attribute name"""
        return ops.live_plans()

    def _check_device_status(self):
        """Fail loudly (PxmError) if a kernel reported an expired bounded wait since the last check -- called where the
        sampler synchronises with the device anyway: saved samples, progress prints, end of run (the reference raises
        on bad state, pxmcmc/mcmc.py:104-109; a silently corrupted chain is not an outcome)."""
        # The status words are LEFT SET: with two samplers in one process each of them sees a latched fault whichever polls
        # first (a fault is fatal for every chain that ran on the plan; ``plan.status(clear=True)`` is the explicit reset).
        for pl in self._device_plans():
            pl.raise_on_fault(clear=False)

    # ---- noise ---------------------------------------------------------------------------
    def _host_noise(self, shape_like):
        """the reference's draw order: randn(N) [+ 1j randn(N)] per chain (pxmcmc/mcmc.py:193-195)."""
        C, N = shape_like.shape
        w = np.stack([np.random.randn(N) + (np.random.randn(N) * 1j if self.complex else 0) for _ in range(C)])
        return ops.as_device(w)


class _DevCounter:
    """caller-owned device iteration counter of the generic stepping engine (the fused engines use the plan's)"""

    def __init__(self, start):
        self.t = torch.full((1,), int(start), dtype=torch.int64, device=ops.device())

    def set(self, v):
        self.t.fill_(int(v))

    def add(self, inc=1):
        ops.counter_add(self.t, inc)

    def close(self):
        pass


class MYULA(PxMCMC):
    """The MYULA chain (pxmcmc/mcmc.py:143-201)."""

    def __init__(self, forward, prox, mcmcparams=PxMCMCParams(), **kwargs):
        super().__init__(forward, prox, mcmcparams, **kwargs)

    def _fusable_wavelet(self):
        f = self.forward
        return (
            getattr(f, "setting", None) == "synthesis"
            and hasattr(getattr(f, "invcov", None), "diag")  # (a full covariance matrix goes through the generic kernels)
            and type(f).calc_gradg.__qualname__.startswith("ForwardOperator")
            and isinstance(getattr(f, "transform", None), SphericalWaveletTransform)
            and isinstance(getattr(f, "measurement", None), Identity)
            and f.measurement.ndata == f.measurement.npix
            and _is_stock_l1(self.prior)
            and type(self).chain_step is MYULA.chain_step
        )

    def _advance(self, X, preds, i, delta=None):
        """one MYULA update X -> X_prop (pxmcmc/mcmc.py:158-160), fused where the operators allow"""
        delta = self.delta if delta is None else delta
        if self._pairs:  # X, preds are pair-packed [ceil(C/2), .]
            noise = self._host_noise_pairs(X) if self.rng == "numpy" else None
            return self._pair_plan.gradg_step(
                X, preds, self._pair_data, self.forward.invcov.diag, self.prior.T_dev, delta, self.lmda, noise=noise,
                seed=self.seed, chain0=self.chain_offset, it=i, pairs=True, noise64=self.noise64,
            )
        noise = self._host_noise(X) if self.rng == "numpy" else None
        kw = dict(noise=noise, noise_complex=bool(self.complex), seed=self.seed, chain0=self.chain_offset, it=i,
                  noise64=self.noise64)
        if self._fused_wav:
            f = self.forward
            return f.transform._plan.gradg_step(
                X, preds, f.data_dev_c128, f.invcov.diag, self.prior.T_dev, delta, self.lmda, **kw
            )
        gradg = ops.as_device(self.forward.calc_gradg(preds), X.dtype)
        if self._fused_prox:
            return ops.myula_step(X, gradg, self.prior.T_dev, delta, self.lmda, **kw)
        proxf = ops.as_device(self.prior.proxf(X), X.dtype)
        if type(self).chain_step is not MYULA.chain_step:
            return ops.as_device(self.chain_step(X, proxf, gradg), X.dtype)
        return ops.chain_step(X, proxf, gradg, delta, self.lmda, **kw)

    def _prepare(self):
        self._fused_wav = self._fusable_wavelet() and isinstance(self.delta, float)
        self._fused_prox = _is_stock_l1(self.prior) and type(self).chain_step is MYULA.chain_step
        self._it = 0
        self._pairs = False

    # ---- two real chains per complex slot (real data, real state) ---------------------------------
    def _pairs_ok(self, X):
        """The reference's state is real-valued (stored as complex128 with a zero imaginary part) when the
        data, the inverse covariance and the start point are real and params.complex is False."""
        f = self.forward
        return bool(
            self._fused_wav and self.real_pairs and not self.complex
            and not f.data_dev.is_complex() and not f.invcov.diag.is_complex()
            and (not X.is_complex() or not bool((X.imag != 0).any()))
        )

    def _pairs_start(self):
        tr = self.forward.transform
        Cs = (self.nchains + 1) // 2
        if getattr(self, "_pair_plan", None) is None or self._pair_plan.max_chains != Cs:
            self._pair_plan = ops.WavPlan(tr.L, tr.B, tr.J_min, max_chains=Cs)  # tables are shared with tr._plan
        d = self.forward.data_dev.to(torch.float64)
        self._pair_data = torch.complex(d, d).contiguous()  # both chains of a slot see the same data
        self._pairs = True

    def _pack(self, X):
        """[C, n] (real-valued) -> [ceil(C/2), n] complex128: chain 2c + i chain 2c+1"""
        re = X.real if X.is_complex() else X
        if re.shape[0] % 2:
            re = torch.cat((re, re[-1:]))  # odd chain count: the last slot's partner is a discarded copy
        return torch.complex(re[0::2].contiguous(), re[1::2].contiguous())

    def _unpack(self, Xp):
        """inverse of _pack, returned as complex128 [C, n] (the reference's state dtype)"""
        Cs, n = Xp.shape
        out = torch.stack((Xp.real, Xp.imag), dim=1).reshape(2 * Cs, n)[: self.nchains]
        return out.to(torch.complex128)

    def _host_noise_pairs(self, Xp):
        """the reference's draw order (one randn(N) per chain, pxmcmc/mcmc.py:193) as a real [2 slots, N] array"""
        Cs, N = Xp.shape
        w = np.zeros((2 * Cs, N))
        for c in range(self.nchains):
            w[c] = np.random.randn(N)
        return ops.as_device(w)

    # ---- HIP-graph engine for the fused wavelet path ----------------------------------------------
    _GRAPH_PAIRS = 4  # iterations per graph replay = 2 * _GRAPH_PAIRS (fewer, longer launches of the host)

    def _graph_ok(self):
        return (self._fused_wav or self._generic_engine_ok()) and self.rng == "philox" and self.use_graph

    def _generic_engine_ok(self):
        """Operators without a fused kernel path (PathIntegralOperator, the analysis setting, user plugins ...) step
        through the same engine: calc_gradg / proxf / chain_step / forward on static buffers with the device Philox
        counter, replayed from a HIP graph when the operators can be captured (eager stepping otherwise)."""
        return (not self._fused_wav and self.rng == "philox" and self.use_graph and isinstance(self.delta, float)
                and type(self).chain_step is MYULA.chain_step)

    def _engine_start(self, X, preds, i0):
        """Static ping-pong state (XA, XB, P), a device iteration counter and a captured graph of 2 * _GRAPH_PAIRS
        iterations."""
        self._engine_stop()  # an engine left over from an interrupted run gives its counter / buffers back first
        f = self.forward
        if not self._fused_wav:
            return self._engine_start_generic(X, preds, i0)
        plan = f.transform._plan
        data = f.data_dev_c128
        self._eng = eng = {}
        eng["pairs"] = self._pairs
        if self._pairs:  # two real chains per complex slot: X, preds are carried pair-packed
            plan, data = self._pair_plan, self._pair_data
            X, preds = self._pack(X), self._pack(preds)
        eng["plan"] = plan
        eng["XA"], eng["XB"], eng["P"] = X.clone(), torch.empty_like(X), preds.clone()
        eng["cnt"] = ops.IterCounter(plan, i0)  # per-plan device counter: the steps below read it at execution time
        eng["side"] = "A"  # which buffer holds the current state
        args = (data, f.invcov.diag, self.prior.T_dev, float(self.delta), self.lmda)
        # params.complex: randn + 1j randn (pxmcmc/mcmc.py:193-195) -> PXM_MODE_CPLX_NOISE in the fused epilogues
        kw = dict(noise_complex=bool(self.complex), seed=self.seed, chain0=self.chain_offset, it=0, pairs=self._pairs,
                  noise64=self.noise64)
        # Uniform inverse covariance (scalar sig_d): the image-space residual is applied on the rings and the
        # L-level iDFT/DFT pair between forward() and calc_gradg() drops out (pxm_wav_ring_step); preds is
        # then materialised only when it is observed.
        d = f.invcov.diag
        eng["ring"] = bool(self.ring_shortcut and d.numel() > 0 and bool((d == d[0]).all()))
        eng["P_valid"] = True
        eng["cnt0"] = lambda i: i  # counter value that makes the next step use Philox iteration i
        if eng["ring"]:
            w = complex(d[0].item())
            plan.ring_set_data(data)
            plan.ring_init(eng["XA"])
            eng["reset"] = lambda: plan.ring_init(eng["XA"])  # plan-carried state of (XA, P)
            eng["cnt0"] = lambda i: i - 1  # ring_step increments the counter before using it
            eng["cnt"].set(eng["cnt0"](i0))

            def one(src, dst):
                # (ring_step advances the registered iteration counter itself, before using it)
                plan.ring_step(src, w, self.prior.T_dev, float(self.delta), self.lmda, out=dst, **kw)
                eng["P_valid"] = False
        else:
            plan.image_init(eng["P"], data, f.invcov.diag)  # residual rings of the start state, carried by the plan
            eng["reset"] = lambda: plan.image_init(eng["P"], data, f.invcov.diag)

            def one(src, dst):
                # calc_gradg + proxf + chain_step + forward of the new state (preds written in place)
                plan.image_step(src, *args, out=dst, preds_out=eng["P"], **kw)
                eng["cnt"].add(1)

        eng["one"] = one
        self._engine_capture(eng, X, preds, i0)
        return eng

    def _engine_start_generic(self, X, preds, i0):
        """The engine for operators without a fused path: one iteration = the reference's four calls
        (pxmcmc/mcmc.py:158-163) on static buffers; the noise kernels read the iteration number from a device counter."""
        f = self.forward
        self._eng = eng = {"pairs": False, "plan": None, "ring": False, "P_valid": True, "side": "A", "generic": True}
        X = ops.as_device(X).contiguous()
        eng["XA"], eng["XB"], eng["P"] = X.clone(), torch.empty_like(X), ops.as_device(preds).clone()
        eng["cnt"] = _DevCounter(i0)
        eng["cnt0"] = lambda i: i
        eng["reset"] = lambda: None
        kw = dict(noise_complex=bool(self.complex), seed=self.seed, chain0=self.chain_offset, it=0, iter_dev=eng["cnt"].t,
                  noise64=self.noise64)
        delta, lmda = float(self.delta), self.lmda

        def one(src, dst):
            gradg = ops.as_device(f.calc_gradg(eng["P"]), src.dtype)
            if self._fused_prox:
                ops.myula_step(src, gradg, self.prior.T_dev, delta, lmda, out=dst, **kw)
            else:
                proxf = ops.as_device(self.prior.proxf(src), src.dtype)
                ops.chain_step(src, proxf, gradg, delta, lmda, out=dst, **kw)
            eng["P"].copy_(ops.as_device(f.forward(dst)))
            eng["cnt"].add(1)

        eng["one"] = one
        self._engine_capture(eng, X, preds, i0)
        return eng

    def _engine_capture(self, eng, X, preds, i0):
        """capture the two graphs of an engine (2 and 2 * _GRAPH_PAIRS iterations); state is (X, preds, i0) afterwards"""
        eng["graph"] = eng["graph_long"] = None
        if self._graph_ok():
            try:
                stream = torch.cuda.Stream()
                stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(stream):
                    eng["one"](eng["XA"], eng["XB"])  # warm-up outside capture (lazy allocations, attributes)
                    eng["one"](eng["XB"], eng["XA"])
                torch.cuda.current_stream().wait_stream(stream)
                torch.cuda.synchronize()
                eng["XA"].copy_(X)
                eng["P"].copy_(preds)
                eng["cnt"].set(eng["cnt0"](i0))
                eng["reset"]()
                # two graphs: 2 iterations (short advances between observable events) and 2 * _GRAPH_PAIRS
                # iterations (long advances: fewer launches by the host).  A plan torn down while a capture is in
                # progress (the garbage collector may run at any point) only queues its device frees: the
                # library empties the queue at the end of the scope.
                graphs = []
                for pairs in (1, self._GRAPH_PAIRS):
                    g = torch.cuda.CUDAGraph()
                    with ops.capture_scope(), torch.cuda.graph(g):
                        for _ in range(pairs):
                            eng["one"](eng["XA"], eng["XB"])
                            eng["one"](eng["XB"], eng["XA"])
                    graphs.append(g)
                # capture does not execute: state is still (X, preds, i0)
                eng["graph"], eng["graph_long"] = graphs
            except Exception as exc:  # capture unsupported in this environment: eager stepping, same results
                eng["graph"] = eng["graph_long"] = None
                eng["graph_error"] = repr(exc)
                eng["XA"].copy_(X)
                eng["P"].copy_(preds)
                eng["cnt"].set(eng["cnt0"](i0))
                eng["reset"]()

    def _engine_advance(self, k):
        """advance the engine's state by k MYULA iterations (graph replays of 2 * _GRAPH_PAIRS + eager remainder)"""
        eng = self._eng
        if eng["side"] == "B" and k > 0:  # realign so that replays start from XA
            eng["one"](eng["XB"], eng["XA"])
            eng["side"] = "A"
            k -= 1
        if eng["graph"] is not None:
            per = 2 * self._GRAPH_PAIRS
            while k >= per:
                eng["graph_long"].replay()
                eng["P_valid"] = not eng["ring"]
                k -= per
            while k >= 2:
                eng["graph"].replay()
                eng["P_valid"] = not eng["ring"]
                k -= 2
        while k >= 2:
            eng["one"](eng["XA"], eng["XB"])
            eng["one"](eng["XB"], eng["XA"])
            k -= 2
        if k == 1:
            eng["one"](eng["XA"], eng["XB"])
            eng["side"] = "B"

    def _engine_state(self):
        """(X, preds) of the current state as [C, .] arrays (pair-packed engines unpack here: observation only)"""
        eng = self._eng
        if eng["ring"] and not eng["P_valid"]:  # forward(X) of the carried rings, on demand
            eng["plan"].ring_preds(eng["P"].shape[0], out=eng["P"])
            eng["P_valid"] = True
        # observation point: the host is about to read the state -- an expired device wait since the last one raises
        if eng.get("plan") is not None:
            eng["plan"].raise_on_fault()
        X = eng["XA"] if eng["side"] == "A" else eng["XB"]
        if eng["pairs"]:
            return self._unpack(X), self._unpack(eng["P"])
        return X, eng["P"]

    def _engine_stop(self):
        """unregister the iteration counter and drop the engine's buffers, graph and closures (the closures
        reference the sampler: without this the plan would only be released by a later garbage collection)"""
        eng = getattr(self, "_eng", None)
        if eng is not None and eng.get("cnt") is not None:
            eng["cnt"].close()
            eng["graph"] = eng["graph"] is not None  # keep the flags (ring, pairs, graph) for inspection
            for k in ("one", "XA", "XB", "P", "plan", "cnt", "cnt0", "graph_long", "reset"):
                eng[k] = None

    def run(self, start_point=None):
        """Run the algorithm (pxmcmc/mcmc.py:150-183)."""
        self._prepare()
        i = 0  # total samples
        j = 0  # saved samples (excludes burn-in and thinned samples)
        X_curr, curr_preds = self._initial_sample(start_point)
        if self._pairs_ok(X_curr):
            self._pairs_start()
        if self.rng == "philox" and (self._fused_wav or self._generic_engine_ok()):
            return self._run_engine(X_curr, curr_preds)
        if self._pairs:
            X_curr, curr_preds = self._pack(X_curr), self._pack(curr_preds)
        while j < self.nsamples:
            X_prop = self._advance(X_curr, curr_preds, i)
            if self._pairs:  # identity measurement: forward = synthesis, on the pair-packed state
                prop_preds = self._pair_plan.synthesis(X_prop)
            else:
                prop_preds = ops.as_device(self.forward.forward(X_prop))

            X_curr = X_prop
            curr_preds = prop_preds

            if i >= self.nburn:
                if self.ngap == 0 or (i - self.nburn) % self.ngap == 0:
                    Xs, Ps = (self._unpack(X_curr), self._unpack(curr_preds)) if self._pairs else (X_curr, curr_preds)
                    self._check_device_status()
                    logPi, L2, prior = self._logpi_dev(Xs, Ps)
                    self._tracking(j, Xs, Ps, logPi, L2, prior)
                    j += 1
                if self.verbosity > 0 and (i + 1) % self.verbosity == 0:
                    first = (lambda a: a[j - 1] if self.nchains == 1 else a[0, j - 1])
                    self._print_progress(j - 1, first(self.logPi), L2=first(self.L2s), prior=first(self.priors))
            else:
                if self.verbosity > 0 and (i + 1) % self.verbosity == 0:
                    print("Burning in...")
            i += 1
        if self._pairs:
            X_curr, curr_preds = self._unpack(X_curr), self._unpack(curr_preds)
        self._check_device_status()
        self.X_curr, self.curr_preds, self.niter = X_curr, curr_preds, i
        print("\nDONE")

    def _run_engine(self, X_curr, curr_preds):
        """
        Same schedule as the reference loop (pxmcmc/mcmc.py:157-181), but iterations between two events
        (save / progress print) are advanced together: by HIP-graph replays when capture is available.
        """
        nburn, ngap, verb = int(self.nburn), int(self.ngap), int(self.verbosity)
        self._engine_start(X_curr, curr_preds, 0)
        try:
            i = 0  # iterations done
            j = 0
            while j < self.nsamples:
                # next iteration index (0-based) at which something observable happens
                if i < nburn:
                    nxt_save = nburn
                elif ngap == 0:
                    nxt_save = i
                else:
                    nxt_save = i + (-(i - nburn)) % ngap
                nxt_print = i + (verb - 1 - i % verb) if verb > 0 else nxt_save
                stop = min(nxt_save, nxt_print)
                self._engine_advance(stop - i + 1)  # run iterations i..stop
                i = stop
                X_curr, curr_preds = self._engine_state()
                if i >= nburn:
                    if ngap == 0 or (i - nburn) % ngap == 0:
                        if self._eng.get("plan") is None:  # generic engine: the operators' own plans
                            self._check_device_status()
                        logPi, L2, prior = self._logpi_dev(X_curr, curr_preds)
                        self._tracking(j, X_curr, curr_preds, logPi, L2, prior)
                        j += 1
                    if verb > 0 and (i + 1) % verb == 0:
                        first = (lambda a: a[j - 1] if self.nchains == 1 else a[0, j - 1])
                        self._print_progress(j - 1, first(self.logPi), L2=first(self.L2s), prior=first(self.priors))
                elif verb > 0 and (i + 1) % verb == 0:
                    print("Burning in...")
                i += 1
            X_curr, curr_preds = self._engine_state()
            self._check_device_status()
            self.X_curr, self.curr_preds, self.niter = X_curr.clone(), curr_preds.clone(), i
            self.used_graph = self._eng["graph"] is not None
            self.graph_error = self._eng.get("graph_error")
        finally:
            self._engine_stop()
        print("\nDONE")

    def chain_step(self, X, proxf, gradg):
        """
        Takes a step in the chain (pxmcmc/mcmc.py:185-201):
        ``(1 - delta/lmda) X + (delta/lmda) proxf - delta gradg + sqrt(2 delta) w``.
        """
        x = ops.as_device(X)
        if self.rng == "numpy":
            noise = self._host_noise(x if x.dim() == 2 else x[None])
        else:
            noise = None
            self._it = getattr(self, "_it", 0) + 1
        out = ops.chain_step(
            x, proxf, gradg, self.delta, self.lmda, noise=noise, noise_complex=bool(self.complex),
            seed=self.seed, chain0=self.chain_offset, it=getattr(self, "_it", 0), noise64=self.noise64,
        )
        return out if isinstance(X, torch.Tensor) else out.cpu().numpy()


class PxMALA(MYULA):
    """
    PxMALA = MYULA proposal + Metropolis-Hastings acceptance (pxmcmc/mcmc.py:204-289).

    :param bool tune_delta: tune ``delta`` towards an acceptance probability of 0.5
    """

    _CHUNK = 1024
    fuse_tail = True  # totals + Metropolis test of an iteration in pxm_pxmala_finish (False: the separate calls, same numbers)

    def __init__(self, forward, prox, mcmcparams=PxMCMCParams(), tune_delta=True, track_transitions=False, max_iter=None,
                 lap_every=0, **kwargs):
        super().__init__(forward, prox, mcmcparams, **kwargs)
        self.tune_delta = tune_delta
        # extension: every ``lap_every`` iterations the loop synchronises the device and appends (iterations done, seconds
        # since the loop started) to ``laps`` -- the time of any stretch of a long run (e.g. after delta has settled)
        # without a second run; 0 = never
        self.lap_every = int(lap_every)
        self.laps = []
        # extension: stop after this many iterations even if fewer than nsamples were saved (the reference's loop,
        # pxmcmc/mcmc.py:230, only ends on accepted samples: a chain that stops accepting never returns)
        self.max_iter = None if max_iter is None else int(max_iter)
        # extension: keep both calc_logtransition values of every iteration in ``transitions_trace`` (a list of
        # (q(X'|X), q(X|X')) complex128 [C] pairs; the static buffers of a graph replay are read after each replay)
        # ... and (prior(X'), L2(X')) of every proposal (pxmcmc/mcmc.py:242) in ``proposals_trace``
        self.track_transitions = bool(track_transitions)
        self.transitions_trace = []
        self.proposals_trace = []

    def _l2_dev(self, preds):
        """L2 = vdot(d, invcov @ d) of a [C, ndata] prediction batch -> complex128 [C] (pxmcmc/mcmc.py:78-79)"""
        inputs = self._l2_inputs(preds)
        if inputs is not None:
            return ops.reduce_l2(*inputs)
        p = ops.as_device(preds)  # full inverse covariance: d = data - preds, then vdot(d, invcov @ d)
        dt = self.forward._resid_dtype(p) if hasattr(self.forward, "_resid_dtype") else p.dtype
        data = self._l2_data_dev(dt)
        d = ops.residual_grad(p.to(dt), data, self.forward.invcov.ones)
        return ops.reduce_vdot(d, self.forward.invcov.matvec(d))

    def _l2_data_dev(self, dt):
        """the data vector on the device in the residual's dtype (cached: no host copy per iteration)"""
        cache = getattr(self, "_l2_data", None)
        if cache is None or cache[0] is not self.forward.data or cache[1].dtype != dt:
            src = getattr(self.forward, "data_dev", None)
            src = ops.as_device(self.forward.data) if src is None else src
            self._l2_data = cache = (self.forward.data, src.reshape(-1).to(dt).contiguous())
        return cache[1]

    def _l2_inputs(self, preds):
        """(preds, data, diagonal of invcov) as the L2 reduction takes them; None with a full inverse covariance"""
        if hasattr(self.forward.invcov, "matvec"):
            return None
        p = ops.as_device(preds)
        dt = self.forward._resid_dtype(p) if hasattr(self.forward, "_resid_dtype") else p.dtype
        ic = getattr(self, "_l2_invcov", None)
        if ic is None or ic[0] is not self.forward.invcov:
            inv = self.forward.invcov
            diag = inv.diag if hasattr(inv, "diag") else ops.as_device(inv.diagonal())
            self._l2_invcov = ic = (inv, ops.as_device(diag).reshape(-1).contiguous())
        return p.to(dt), self._l2_data_dev(dt), ic[1]

    def run(self, start_point=None):
        """Run the algorithm (pxmcmc/mcmc.py:218-275); every chain carries its own delta and accept flag.

        One iteration is a fixed sequence of device operations on static buffers -- proposal + prox + forward
        transition + prior in one pass (pxm_pxmala_propose), forward model, gradient, L2, reverse transition,
        Metropolis test / delta adaptation / traces on the device (pxm_pxmala_accept2), one conditional copy of the
        accepted states -- and, with the device Philox stream, is replayed from a captured HIP graph between
        observable events (save candidates, progress prints, trace flushes)."""
        self._prepare()
        self.laps = []
        self._fused_wav = False  # PxMALA needs gradg and proxf of the proposal separately
        C = self.nchains
        dev = ops.device()
        acc_chunks, delta_chunks = [], []
        acc_buf = torch.zeros((self._CHUNK, C), dtype=torch.int32, device=dev)
        delta_buf = torch.zeros((self._CHUNK, C), dtype=torch.float64, device=dev)
        delta_dev = torch.full((C,), float(self.delta), dtype=torch.float64, device=dev)
        delta0 = float(self.delta)
        j = np.zeros(C, dtype=int)
        X_curr, curr_preds = self._initial_sample(start_point)
        dt = X_curr.dtype
        X_curr = X_curr.contiguous()
        curr_preds = ops.as_device(curr_preds).clone()
        gradg_curr = ops.as_device(self.forward.calc_gradg(curr_preds), dt).clone()
        proxf_curr = ops.as_device(self.prior.proxf(X_curr), dt).clone()
        logpiXc, L2Xc, priorXc = self._logpi_dev(X_curr, curr_preds)
        logpiXc, L2Xc = logpiXc.to(torch.complex128).contiguous(), L2Xc.to(torch.complex128).contiguous()
        priorXc = priorXc.to(torch.float64).contiguous()
        # stock prior (library L1 / S2 soft threshold + weighted L1 norm): the fused proposal kernel applies
        stock = _is_stock_l1(self.prior) and type(self.prior).prior is L1.prior and type(self).chain_step is MYULA.chain_step
        T_dev = self.prior.T_dev if stock else None
        w_prior = getattr(self.prior, "_weights_dev", None) if stock else None
        X_prop, proxf_prop = torch.empty_like(X_curr), torch.empty_like(X_curr)
        lt_cp = torch.empty(C, dtype=torch.complex128, device=dev)
        prior_p = torch.empty(C, dtype=torch.float64, device=dev)
        accept = torch.zeros(C, dtype=torch.int32, device=dev)
        it_dev = torch.zeros(1, dtype=torch.int64, device=dev)  # device-resident iteration number (graph replay)
        host_rng = self.rng == "numpy"

        # stock prior + diagonal inverse covariance: the totals of the proposal pass are deferred and everything between
        # the proposal's gradient and the conditional copy is two launches (pxm_pxmala_finish) instead of seven; the sums
        # are added in the same order either way (``fuse_tail = False``: the separate calls, bit-identical)
        fused_tail = bool(stock and self.fuse_tail and not hasattr(self.forward.invcov, "matvec"))
        if fused_tail:
            prop_scratch = ops.pxmala_propose_scratch(C, dev)
            fin_scratch = torch.empty(2 * ops.reduce_scratch_doubles(C), dtype=torch.float64, device=dev)
            lt_pc = torch.empty(C, dtype=torch.complex128, device=dev)
            L2_p = torch.empty(C, dtype=torch.complex128, device=dev)

        def iteration(i_host, counter, bump=None):
            """one PxMALA iteration; Philox / adaptation use iteration number i_host + *counter"""
            kw = dict(seed=self.seed, chain0=self.chain_offset, it=i_host)
            noise = self._host_noise(X_curr) if host_rng else None
            if stock:
                # (fused tail: no prox arrays at all -- soft(X, T) is formed where it is needed)
                ops.pxmala_propose(X_curr, None if fused_tail else proxf_curr, gradg_curr, T_dev, w_prior, delta_dev, self.lmda,
                                   X_prop, None if fused_tail else proxf_prop,
                                   None if fused_tail else lt_cp, None if fused_tail else prior_p, noise=noise,
                                   noise_complex=bool(self.complex), iter_dev=counter, noise64=self.noise64,
                                   scratch=prop_scratch if fused_tail else None, **kw)
                Xp, pxp, ltc, prp = X_prop, proxf_prop, lt_cp, prior_p
            else:  # user-supplied prior / chain_step: the reference's own sequence of calls (mcmc.py:231-242)
                if type(self).chain_step is MYULA.chain_step:
                    Xp = ops.chain_step(X_curr, proxf_curr, gradg_curr, delta_dev, self.lmda, noise=noise,
                                        noise_complex=bool(self.complex), noise64=self.noise64, **kw)
                else:
                    Xp = ops.as_device(self.chain_step(X_curr, proxf_curr, gradg_curr), dt)
                pxp = ops.as_device(self.prior.proxf(Xp), dt)
                ltc = ops.logtransition(X_curr, Xp, proxf_curr, gradg_curr, delta_dev, self.lmda)
                prp = self.prior.prior(Xp)
                if not isinstance(prp, torch.Tensor):
                    prp = torch.as_tensor(np.atleast_1d(np.asarray(prp, dtype=float)), device=dev)
                prp = prp.to(torch.float64).contiguous()
            pp = ops.as_device(self.forward.forward(Xp))
            gp = ops.as_device(self.forward.calc_gradg(pp), dt)
            if fused_tail:
                p_, data_, ic_ = self._l2_inputs(pp)
                u = np.array([np.random.rand() for _ in range(C)]) if host_rng else None
                ops.pxmala_finish(Xp, X_curr, None, gp.contiguous(), p_, data_, ic_, prop_scratch, self.mu, self.lmda, logpiXc,
                                  L2Xc, priorXc, accept, delta_dev, self.tune_delta, lt_pc, lt_cp, prior_p, L2_p, fin_scratch,
                                  u=u, iter_dev=counter, acc_trace=acc_buf, delta_trace=delta_buf, bump=bump,
                                  T=T_dev if T_dev is not None else 0.0, **kw)
                self._last_transitions = (lt_cp, lt_pc)
                self._last_proposal = (prior_p, L2_p)
                ops.select_copy_many(accept, [(Xp, X_curr), (pp.to(curr_preds.dtype), curr_preds), (gp, gradg_curr)])
                return
            L2p = self._l2_dev(pp)
            ltp = ops.logtransition(Xp, X_curr, pxp, gp, delta_dev, self.lmda)
            self._last_transitions = (ltc, ltp)  # q(X'|X), q(X|X') of this iteration (pxmcmc/mcmc.py:240-241)
            self._last_proposal = (prp, L2p)
            u = np.array([np.random.rand() for _ in range(C)]) if host_rng else None
            ops.pxmala_accept2(ltp, ltc, prp, L2p, self.mu, logpiXc, L2Xc, priorXc, accept, delta_dev, self.tune_delta,
                               self.lmda, u=u, iter_dev=counter, acc_trace=acc_buf, delta_trace=delta_buf, **kw)
            ops.select_copy_many(accept, [(Xp, X_curr), (pp.to(curr_preds.dtype), curr_preds), (gp, gradg_curr), (pxp, proxf_curr)])

        # HIP graph of one iteration (device Philox stream only; any operator that synchronises or cannot be
        # captured falls back to eager stepping -- same results)
        graph = None
        self.graph_error = None
        if self.use_graph and not host_rng and stock:
            snap = [t.clone() for t in (X_curr, curr_preds, gradg_curr, proxf_curr, logpiXc, L2Xc, priorXc, delta_dev)]
            try:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    iteration(0, it_dev)  # warm-up outside capture (lazy allocations, attributes)
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                for t, s_ in zip((X_curr, curr_preds, gradg_curr, proxf_curr, logpiXc, L2Xc, priorXc, delta_dev), snap):
                    t.copy_(s_)
                g = torch.cuda.CUDAGraph()
                with ops.capture_scope(), torch.cuda.graph(g):
                    if fused_tail:
                        iteration(0, it_dev, bump=it_dev)  # (the counter advances inside pxm_pxmala_finish)
                    else:
                        iteration(0, it_dev)
                        ops.counter_add(it_dev, 1)
                graph = g
            except Exception as exc:
                graph = None
                self.graph_error = repr(exc)
                for t, s_ in zip((X_curr, curr_preds, gradg_curr, proxf_curr, logpiXc, L2Xc, priorXc, delta_dev), snap):
                    t.copy_(s_)
            it_dev.zero_()
        self.used_graph = graph is not None

        i = 0
        n_acc = 0
        torch.cuda.synchronize()
        t_loop = time.perf_counter()  # (loop_seconds: the iterations alone, without set-up and graph capture)
        while j.min() < self.nsamples and (self.max_iter is None or i < self.max_iter):
            if graph is not None:
                graph.replay()
            else:
                iteration(i, None)
            if self.track_transitions:  # observation only (synchronises): both calc_logtransition values per iteration
                self.transitions_trace.append(tuple(t.cpu().numpy().copy() for t in self._last_transitions))
                self.proposals_trace.append(tuple(t.cpu().numpy().copy() for t in self._last_proposal))
            k = i % self._CHUNK
            if k == self._CHUNK - 1:
                acc_chunks.append(acc_buf.cpu().numpy().copy())
                delta_chunks.append(delta_buf.cpu().numpy().copy())
                n_acc += int(acc_chunks[-1][:, 0].sum())  # running count of chain 0 (progress print)

            gap_it = i >= self.nburn and (self.ngap == 0 or (i - self.nburn) % self.ngap == 0)
            if gap_it:
                acc_h = accept.cpu().numpy()  # the only per-iteration host sync, on save candidates only
                chains = [c for c in range(C) if acc_h[c] and j[c] < self.nsamples]
                if chains:
                    self._check_device_status()
                    self._tracking(j[chains] if C > 1 else int(j[0]), X_curr, curr_preds, logpiXc, L2Xc, priorXc,
                                   chains=chains if C > 1 else None)
                    j[chains] += 1
            if self.lap_every > 0 and (i + 1) % self.lap_every == 0:
                torch.cuda.synchronize()
                self.laps.append((i + 1, time.perf_counter() - t_loop))
            if self.verbosity > 0 and (i + 1) % self.verbosity == 0:
                pending = 0 if k == self._CHUNK - 1 else int(acc_buf[: k + 1, 0].sum().item())  # rows not yet flushed
                rate = (n_acc + pending) / (i + 1)
                self._print_progress(
                    int(j[0]) - 1, float(logpiXc[0].real), L2=float(L2Xc[0].real), prior=float(priorXc[0]), acceptanceRate=rate
                )
            i += 1
        torch.cuda.synchronize()
        self.loop_seconds = time.perf_counter() - t_loop
        k = i % self._CHUNK
        if k:
            acc_chunks.append(acc_buf[:k].cpu().numpy().copy())
            delta_chunks.append(delta_buf[:k].cpu().numpy().copy())
        acc_all = np.concatenate(acc_chunks) if acc_chunks else np.zeros((0, C), dtype=np.int32)
        del_all = np.concatenate([np.full((1, C), delta0)] + (delta_chunks if self.tune_delta else []))
        if C == 1:
            self.acceptance_trace = [int(v) for v in acc_all[:, 0]]
            self.deltas_trace = [float(v) for v in del_all[:, 0]]
        else:
            self.acceptance_trace = acc_all
            self.deltas_trace = del_all
        self.delta = float(delta_dev[0].item())
        self.delta_dev = delta_dev
        self._check_device_status()
        self.X_curr, self.curr_preds, self.niter = X_curr, curr_preds, i
        # rows of chain / logPi / ... beyond nsaved[c] were never written (zeros): a ``max_iter`` stop says so
        self.nsaved = j.copy() if C > 1 else int(j[0])
        self.stopped_early = bool(j.min() < self.nsamples)
        print("\nDONE")

    def _tune_delta(self, i):
        """pxmcmc/mcmc.py:277-279 (host form, one chain; the run loop adapts on the device)."""
        delta = self.delta * (1 + (self.acceptance_trace[i] - 0.5) / ((i + 1) ** 0.75))
        self.delta = min(max(delta, self.lmda * 1e-8), self.lmda / 2)

    def calc_logtransition(self, X1, X2, proxf, gradg):
        """q(X2|X1), literal (pxmcmc/mcmc.py:281-289)."""
        r = ops.logtransition(X1, X2, proxf, gradg, self.delta, self.lmda)
        batched = (X1.dim() if isinstance(X1, torch.Tensor) else np.ndim(X1)) == 2
        if batched:
            return r
        v = complex(r[0].item())
        return v if v.imag != 0 else v.real
