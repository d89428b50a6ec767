"""
Thin tensor-level wrappers over the C-ABI: torch supplies device memory and the stream,
every operation is a hand-written HIP kernel behind ``include/pxmcmc_amd.h``.

Conventions: arrays are ``[C, n]`` (chain batch first) or ``[n]`` (one chain); dtype is
float64 or complex128; outputs are fresh tensors (the reference never mutates inputs,
SURVEY.md section 8b).
"""
import ctypes as C
import threading
import weakref

import numpy as np
import torch

from ._lib import NOISE_F64, STATUS_FLOW_WAIT, STATUS_PAIR_SYNC, PxmError, check, lib, require_gpu

_CPLX, _REAL = torch.complex128, torch.float64


def device():
    require_gpu()
    return torch.device("cuda", torch.cuda.current_device())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def as_device(x, dtype=None):
    """numpy / torch / sequence -> contiguous tensor on the GPU (float64 or complex128)."""
    if isinstance(x, torch.Tensor):
        t = x
    else:
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(x)))
    if dtype is None:
        dtype = _CPLX if t.is_complex() else _REAL
    return t.to(device=device(), dtype=dtype).contiguous()


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _batched(x):
    """[n] -> ([1, n], True); [C, n] -> (x, False)."""
    if x.dim() == 1:
        return x.unsqueeze(0), True
    if x.dim() != 2:
        raise ValueError("expected a 1-D (single chain) or 2-D (chain batch) array")
    return x, False


def _dt(x):
    if x.dtype == _CPLX:
        return 1
    if x.dtype == _REAL:
        return 0
    raise TypeError(f"unsupported dtype {x.dtype}: float64 or complex128 only")


def _vecT(T, n, dev):
    """threshold / weight argument: python scalar -> (null, value); vector -> (tensor, 0)."""
    if T is None:
        return None, 0.0
    if isinstance(T, (int, float)):
        return None, float(T)
    t = as_device(T, _REAL)
    if t.numel() == 1:
        return None, float(t.item())
    if t.numel() != n:
        raise ValueError("threshold / weight vector has the wrong length")
    return t.reshape(-1), 0.0


# ---- elementwise -----------------------------------------------------------------------
def soft(X, T=0.1):
    """utils.soft (pxmcmc/utils.py:55-67)."""
    x, squeeze = _batched(as_device(X))
    Tv, Ts = _vecT(T, x.shape[1], x.device)
    out = torch.empty_like(x)
    check(lib.pxm_soft(_p(x), _p(Tv), Ts, _p(out), x.shape[1], x.shape[0], _dt(x), _stream()))
    return out[0] if squeeze else out


def residual_grad(preds, data, invcov):
    """invcov .* (preds - data), invcov diagonal (pxmcmc/forward.py:66-69)."""
    p, squeeze = _batched(as_device(preds))
    n = p.shape[1]
    d = as_device(data, p.dtype).reshape(-1)
    ic = as_device(invcov)
    if ic.is_complex() and not p.is_complex():
        raise TypeError("complex inverse covariance needs complex predictions")
    ic = ic.reshape(-1)
    if d.numel() != n or ic.numel() != n:
        raise ValueError("data / invcov length mismatch")
    out = torch.empty_like(p)
    check(lib.pxm_residual_grad(_p(p), _p(d), _p(ic), int(ic.is_complex()), _p(out), n, p.shape[0], _dt(p), _stream()))
    return out[0] if squeeze else out


def _nf(noise64):
    """PXM_NOISE_F64 flag of the noise-drawing entry points: the Philox stream's Box-Muller step in double precision"""
    return NOISE_F64 if noise64 else 0


def raise_on_status(status, what):
    """A plan's device status word (pxm_wav_status / pxm_sht_status) -> PxmError when a bounded wait expired"""
    if status:
        bits = []
        if status & STATUS_PAIR_SYNC:
            bits.append("a wave-pair wait or ring-group wait of the fused phi-DFT kernels expired (csrc/dft5.hip: d5_pair_sync, pfa_group_sync)")
        if status & STATUS_FLOW_WAIT:
            bits.append("a wait of the dataflow GEMM launch expired (PXM_FLOW=1)")
        if status & ~(STATUS_PAIR_SYNC | STATUS_FLOW_WAIT):
            bits.append(f"unknown status bits {status:#x}")
        raise PxmError(f"{what}: device status {status:#x}: " + "; ".join(bits) + " -- the results of this plan since its last "
                       "status check are invalid")


def _delta_args(delta, C_, dev):
    if isinstance(delta, torch.Tensor):
        dd = delta.to(device=dev, dtype=_REAL).contiguous()
        if dd.numel() != C_:
            raise ValueError("per-chain delta must have one entry per chain")
        return dd, 0.0
    return None, float(delta)


def _pair_noise_args(noise, x):
    """mode 2 (two real chains per complex slot): injected noise is a real [2 * slots, N] array"""
    if noise is None:
        return None, 2
    w = as_device(noise, _REAL)
    if w.dim() != 2 or w.shape != (2 * x.shape[0], x.shape[1]):
        raise ValueError("real-pair noise must be a float64 [2 * slots, N] array")
    return w, 2


def _noise_args(noise, x, noise_complex):
    if noise is None:
        return None, int(bool(noise_complex))
    w = as_device(noise)
    if w.dim() == 1:
        w = w.unsqueeze(0)
    if w.shape != x.shape:
        raise ValueError("injected noise must have the state's shape")
    if w.is_complex() and not x.is_complex():
        raise TypeError("complex noise needs a complex state")
    return w, int(w.is_complex())


def myula_step(X, gradg, T, delta, lmda, noise=None, noise_complex=False, seed=0, chain0=0, it=0, iter_dev=None, out=None,
               noise64=False):
    """chain_step(X, soft(X, T), gradg) in one pass (pxmcmc/mcmc.py:185-201 + prior.py:49-50).
    iter_dev: int64 device counter added to ``it`` when the kernel runs (graph replay); out: result buffer."""
    x, squeeze = _batched(as_device(X))
    g, _ = _batched(as_device(gradg, x.dtype))
    if g.shape != x.shape:
        raise ValueError("gradg shape mismatch")
    Tv, Ts = _vecT(T, x.shape[1], x.device)
    dd, ds = _delta_args(delta, x.shape[0], x.device)
    w, wc = _noise_args(noise, x, noise_complex)
    out = torch.empty_like(x) if out is None else _out_like(out, x)
    check(
        lib.pxm_myula_step_it(
            _p(x), _p(g), _p(Tv), Ts, _p(dd), ds, float(lmda), _p(w), wc | _nf(noise64), seed, chain0, it, _p(iter_dev), _p(out), x.shape[1],
            x.shape[0], _dt(x), _stream()
        )
    )
    return out[0] if squeeze else out


def _out_like(out, x):
    o = out if out.dim() == 2 else out.unsqueeze(0)
    if o.shape != x.shape or o.dtype != x.dtype or not o.is_contiguous() or o.device != x.device:
        raise ValueError("out must be a contiguous device array of the state's shape and dtype")
    return o


def chain_step(X, proxf, gradg, delta, lmda, noise=None, noise_complex=False, seed=0, chain0=0, it=0, iter_dev=None, out=None,
               noise64=False):
    """MYULA.chain_step (pxmcmc/mcmc.py:185-201); iter_dev / out as in :func:`myula_step`."""
    x, squeeze = _batched(as_device(X))
    px, _ = _batched(as_device(proxf, x.dtype))
    g, _ = _batched(as_device(gradg, x.dtype))
    if g.shape != x.shape or px.shape != x.shape:
        raise ValueError("shape mismatch")
    dd, ds = _delta_args(delta, x.shape[0], x.device)
    w, wc = _noise_args(noise, x, noise_complex)
    out = torch.empty_like(x) if out is None else _out_like(out, x)
    check(
        lib.pxm_chain_step_it(
            _p(x), _p(px), _p(g), _p(dd), ds, float(lmda), _p(w), wc | _nf(noise64), seed, chain0, it, _p(iter_dev), _p(out), x.shape[1],
            x.shape[0], _dt(x), _stream()
        )
    )
    return out[0] if squeeze else out


def randn(n, C_=1, complex_=False, seed=0, chain0=0, it=0, noise64=False):
    """N(0,1) draws of the device Philox stream keyed (seed, chain0 + c, it); noise64: Box-Muller in double precision"""
    out = torch.empty((C_, n), dtype=_CPLX if complex_ else _REAL, device=device())
    check(lib.pxm_randn(_p(out), n, C_, int(complex_) | _nf(noise64), seed, chain0, it, _stream()))
    return out


def box_muller(u1, u2, noise64=False):
    """the Box-Muller step of the device noise stream on given uniforms (test aid, include/pxmcmc_amd.h) -> (z0, z1)"""
    a, b = as_device(u1, _REAL).reshape(-1), as_device(u2, _REAL).reshape(-1)
    if a.shape != b.shape:
        raise ValueError("box_muller: u1 and u2 must have the same length")
    z0, z1 = torch.empty_like(a), torch.empty_like(a)
    check(lib.pxm_box_muller(_p(a), _p(b), _p(z0), _p(z1), a.numel(), int(bool(noise64)), _stream()))
    return z0, z1


def _red_scratch(C_, dev):
    """caller-owned scratch of the two-stage reductions (torch's caching allocator: stream-ordered reuse)"""
    return torch.empty(int(lib.pxm_reduce_scratch_doubles(int(C_))), dtype=_REAL, device=dev)


def reduce_l1(X, w=None):
    """sum |w X| per chain (pxmcmc/prior.py:28-35,83-84) -> float64 [C]."""
    x, _ = _batched(as_device(X))
    wv = None if w is None else as_device(w, _REAL).reshape(-1)
    if wv is not None and wv.numel() != x.shape[1]:
        raise ValueError("weight length mismatch")
    out = torch.empty(x.shape[0], dtype=_REAL, device=x.device)
    check(lib.pxm_reduce_l1(_p(x), _p(wv), _p(out), _p(_red_scratch(x.shape[0], x.device)), x.shape[1], x.shape[0], _dt(x), _stream()))
    return out


def quantile_range(chain, alpha=0.05):
    """Q(1 - alpha/2) - Q(alpha/2) of every column of a float64 [nsamples, nparams] tensor on the device
    (pxmcmc/uncertainty.py:7-16; numpy.quantile's default method) -> float64 [nparams]."""
    require_gpu()
    c = chain if isinstance(chain, torch.Tensor) else as_device(np.asarray(chain, dtype=np.float64), _REAL)
    if c.dim() != 2 or c.dtype != _REAL:
        raise TypeError("quantile_range: a float64 [nsamples, nparams] array is expected")
    if not c.is_cuda:
        c = c.to(device())
    if c.stride(1) != 1:
        c = c.contiguous()
    out = torch.empty(c.shape[1], dtype=_REAL, device=c.device)
    check(lib.pxm_quantile_range(_p(c), c.shape[0], c.shape[1], c.stride(0), float(alpha), _p(out), _stream()))
    return out


def reduce_l2(preds, data, invcov):
    """vdot(d, invcov d), d = data - preds (pxmcmc/mcmc.py:78-79) -> complex128 [C]."""
    p, _ = _batched(as_device(preds))
    n = p.shape[1]
    d = as_device(data, p.dtype).reshape(-1)
    ic = as_device(invcov).reshape(-1)
    if ic.is_complex() and not p.is_complex():
        raise TypeError("complex inverse covariance needs complex predictions")
    if d.numel() != n or ic.numel() != n:
        raise ValueError("data / invcov length mismatch")
    out = torch.empty(p.shape[0], dtype=_CPLX, device=p.device)
    check(lib.pxm_reduce_l2(_p(p), _p(d), _p(ic), int(ic.is_complex()), _p(out), _p(_red_scratch(p.shape[0], p.device)), n, p.shape[0], _dt(p), _stream()))
    return out


def reduce_vdot(a, b):
    """np.vdot(a, b) per chain -> complex128 [C]"""
    x, _ = _batched(as_device(a))
    y, _ = _batched(as_device(b, x.dtype))
    if x.shape != y.shape:
        raise ValueError("vdot: shape mismatch")
    out = torch.empty(x.shape[0], dtype=_CPLX, device=x.device)
    check(lib.pxm_reduce_vdot(_p(x), _p(y), _p(out), _p(_red_scratch(x.shape[0], x.device)), x.shape[1], x.shape[0], _dt(x), _stream()))
    return out


def logtransition(X1, X2, proxf, gradg, delta, lmda):
    """PxMALA.calc_logtransition, literal (pxmcmc/mcmc.py:281-289) -> complex128 [C]."""
    x1, _ = _batched(as_device(X1))
    x2, _ = _batched(as_device(X2, x1.dtype))
    px, _ = _batched(as_device(proxf, x1.dtype))
    g, _ = _batched(as_device(gradg, x1.dtype))
    dd, ds = _delta_args(delta, x1.shape[0], x1.device)
    out = torch.empty(x1.shape[0], dtype=_CPLX, device=x1.device)
    check(lib.pxm_logtransition(_p(x1), _p(x2), _p(px), _p(g), _p(dd), ds, float(lmda), _p(out), _p(_red_scratch(x1.shape[0], x1.device)), x1.shape[1], x1.shape[0], _dt(x1), _stream()))
    return out


def pxmala_accept(terms, delta_dev, tune, lmda, it_index, u=None, seed=0, chain0=0, it=0):
    """Metropolis test + delta adaptation per chain (pxmcmc/mcmc.py:244-260,277-279)."""
    t = as_device(terms, _REAL)
    C_ = t.shape[0]
    uu = None if u is None else as_device(u, _REAL).reshape(-1)
    acc = torch.empty(C_, dtype=torch.int32, device=t.device)
    check(lib.pxm_pxmala_accept(_p(t), _p(uu), seed, chain0, it, _p(acc), _p(delta_dev), int(bool(tune)), float(lmda), int(it_index), C_, _stream()))
    return acc


def reduce_scratch_doubles(C_):
    return int(lib.pxm_reduce_scratch_doubles(int(C_)))


def pxmala_propose_scratch(C_, dev):
    """scratch of pxmala_propose for C_ chains (kept by the caller when the totals are deferred to pxmala_finish)"""
    return torch.empty(4 * int(lib.pxm_reduce_scratch_doubles(int(C_))), dtype=_REAL, device=dev)


def pxmala_propose(X, proxf, gradg, T, prior_weights, delta_dev, lmda, Xp, proxf_p, lt_out, prior_out, noise=None,
                   noise_complex=False, seed=0, chain0=0, it=0, iter_dev=None, noise64=False, scratch=None):
    """chain_step + soft + calc_logtransition(X, X') + prior(X') in one pass; writes into the given buffers
    (Xp, proxf_p [C, n]; lt_out complex128 [C]; prior_out float64 [C]).  lt_out = prior_out = None with a caller-owned
    ``scratch`` (pxmala_propose_scratch): the totals are left to pxmala_finish.  proxf = proxf_p = None: the prox arrays
    are neither read nor written (soft(X, T) is formed in the kernel)."""
    if (proxf is None) != (proxf_p is None):
        raise ValueError("pxmala_propose: proxf and proxf_p are given together or not at all")
    x, _ = _batched(X)
    Tv, Ts = _vecT(T, x.shape[1], x.device)
    w, wc = _noise_args(noise, x, noise_complex)
    wp = None if prior_weights is None else as_device(prior_weights, _REAL).reshape(-1)
    if (lt_out is None) != (prior_out is None) or (lt_out is None and scratch is None):
        raise ValueError("pxmala_propose: deferred totals need lt_out = prior_out = None and a caller-owned scratch")
    if scratch is None:
        scratch = pxmala_propose_scratch(x.shape[0], x.device)
    check(
        lib.pxm_pxmala_propose(
            _p(x), _p(proxf), _p(gradg), _p(Tv), Ts, _p(wp), _p(delta_dev), float(lmda), _p(w), wc | _nf(noise64), seed, chain0, int(it),
            _p(iter_dev), _p(Xp), _p(proxf_p), _p(lt_out), _p(prior_out), _p(scratch), x.shape[1], x.shape[0], _dt(x), _stream(),
        )
    )


def pxmala_accept2(lt_pc, lt_cp, prior_p, L2_p, mu, logpi_c, L2_c, prior_c, accept, delta_dev, tune, lmda, u=None, seed=0,
                   chain0=0, it=0, iter_dev=None, acc_trace=None, delta_trace=None):
    """Metropolis test + state scalars + delta adaptation + traces on the device (pxmcmc/mcmc.py:244-260,277-279)."""
    C_ = accept.shape[0]
    uu = None if u is None else as_device(u, _REAL).reshape(-1)
    chunk = 0 if acc_trace is None else acc_trace.shape[0]
    check(
        lib.pxm_pxmala_accept2(
            _p(lt_pc), _p(lt_cp), _p(prior_p), _p(L2_p), float(mu), _p(logpi_c), _p(L2_c), _p(prior_c), _p(uu), seed, chain0,
            int(it), _p(iter_dev), _p(accept), _p(delta_dev), int(bool(tune)), float(lmda), _p(acc_trace), _p(delta_trace),
            int(chunk), C_, _stream(),
        )
    )


def pxmala_finish(Xp, X, proxf_p, gradg_p, preds_p, data, invcov, propose_scratch, mu, lmda, logpi_c, L2_c, prior_c, accept,
                  delta_dev, tune, lt_pc_out, lt_cp_out, prior_p_out, L2_p_out, scratch, u=None, seed=0, chain0=0, it=0,
                  iter_dev=None, acc_trace=None, delta_trace=None, bump=None, T=None):
    """Reverse transition sum + L2 of the proposal in one grid, then totals (incl. the deferred ones of pxmala_propose) +
    Metropolis test + state scalars + delta adaptation + traces in one workgroup (pxmcmc/mcmc.py:239-260,277-279);
    ``scratch``: 2 * pxm_reduce_scratch_doubles(C) doubles, caller-owned; ``bump``: device iteration counter to advance;
    ``proxf_p = None`` with the threshold ``T``: proxf' = soft(X', T) is formed in the kernel instead of being read."""
    xp, _ = _batched(Xp)
    if proxf_p is None and T is None:
        raise ValueError("pxmala_finish: without proxf_p the threshold T is needed")
    Tv, Ts = _vecT(T, xp.shape[1], xp.device) if proxf_p is None else (None, 0.0)
    pp, _ = _batched(preds_p)
    nd = pp.shape[1]
    d = data.reshape(-1)
    ic = invcov.reshape(-1)
    if d.dtype != pp.dtype or d.numel() != nd or ic.numel() != nd or (ic.is_complex() and not pp.is_complex()):
        raise ValueError("pxmala_finish: data / invcov do not match the predictions")
    for t in (X, proxf_p, gradg_p):
        if t is not None and (t.shape != xp.shape or t.dtype != xp.dtype or not t.is_contiguous()):
            raise ValueError("pxmala_finish: state arrays must share shape, dtype and be contiguous")
    C_ = accept.shape[0]
    uu = None if u is None else as_device(u, _REAL).reshape(-1)
    chunk = 0 if acc_trace is None else acc_trace.shape[0]
    check(
        lib.pxm_pxmala_finish(
            _p(xp), _p(X), _p(proxf_p), _p(Tv), Ts, _p(gradg_p), xp.shape[1], _dt(xp), _p(pp), _p(d), _p(ic), int(ic.is_complex()), nd, _dt(pp),
            _p(propose_scratch), float(mu), float(lmda), _p(logpi_c), _p(L2_c), _p(prior_c), _p(uu), seed, chain0, int(it),
            _p(iter_dev), _p(accept), _p(delta_dev), int(bool(tune)), _p(acc_trace), _p(delta_trace), int(chunk), _p(lt_pc_out),
            _p(lt_cp_out), _p(prior_p_out), _p(L2_p_out), _p(scratch), _p(bump), C_, _stream(),
        )
    )


def select_copy_many(flag, pairs):
    """dst[c] = src[c] for chains with flag[c] != 0, for up to four (src, dst) pairs in one launch."""
    k = len(pairs)
    if not 1 <= k <= 4:
        raise ValueError("select_copy_many takes 1 to 4 array pairs")
    srcs = (C.c_void_p * k)(*[p[0].data_ptr() for p in pairs])
    dsts = (C.c_void_p * k)(*[p[1].data_ptr() for p in pairs])
    ns = (C.c_int64 * k)(*[p[0][0].numel() if p[0].dim() == 2 else p[0].numel() for p in pairs])
    es = (C.c_int * k)(*[p[0].element_size() for p in pairs])
    for s_, d_ in pairs:
        if s_.shape != d_.shape or s_.dtype != d_.dtype or not s_.is_contiguous() or not d_.is_contiguous():
            raise ValueError("select_copy_many: shape / dtype / layout mismatch")
    check(lib.pxm_select_copy_many(_p(flag), k, srcs, dsts, ns, es, flag.shape[0], _stream()))


def counter_add(counter, inc=1):
    check(lib.pxm_counter_add(_p(counter), int(inc), _stream()))


def select_copy(flag, src, dst):
    """dst[c] = src[c] for chains with flag[c] != 0 (in place on dst)."""
    s, _ = _batched(src)
    d, _ = _batched(dst)
    if s.shape != d.shape or s.dtype != d.dtype:
        raise ValueError("select_copy: shape / dtype mismatch")
    check(lib.pxm_select_copy(_p(flag), _p(s), _p(d), s.shape[1], s.element_size(), s.shape[0], _stream()))
    return dst


# ---- sparse measurement ----------------------------------------------------------------
class CsrMatrix:
    """A scipy.sparse matrix resident on the GPU in CSR form (int64 indptr, int32 indices, f64 / c128 values)."""

    def __init__(self, mat):
        import scipy.sparse as sp

        m = sp.csr_matrix(mat)
        m.sum_duplicates()
        m.sort_indices()
        self.shape = m.shape
        self.is_complex = np.iscomplexobj(m.data)
        dev = device()
        self.indptr = torch.from_numpy(m.indptr.astype(np.int64)).to(dev)
        self.indices = torch.from_numpy(m.indices.astype(np.int32)).to(dev)
        self.vals = torch.from_numpy(np.ascontiguousarray(m.data.astype(np.complex128 if self.is_complex else np.float64))).to(dev)
        self.nnz = int(m.nnz)

    def matvec(self, X):
        """[ncols] or [C, ncols] -> [nrows] or [C, nrows] (float64 stays float64 under a real matrix)"""
        x, squeeze = _batched(as_device(X))
        if self.is_complex and not x.is_complex():
            x = x.to(_CPLX)
        if x.shape[1] != self.shape[1]:
            raise AssertionError(f"expected length {self.shape[1]}, got {x.shape[1]}")
        out = torch.empty((x.shape[0], self.shape[0]), dtype=x.dtype, device=x.device)
        # chain batches gather from a chain-minor copy of the operand (caller-owned scratch, stream-ordered reuse)
        scratch = torch.empty(x.numel(), dtype=x.dtype, device=x.device) if x.shape[0] > 1 else None
        check(lib.pxm_csr_matvec_batched(_p(self.indptr), _p(self.indices), _p(self.vals), int(self.is_complex), self.shape[0],
                                         self.shape[1], _p(x), _p(out), x.shape[0], _dt(x), _p(scratch), _stream()))
        return out[0] if squeeze else out


# ---- transform plans -------------------------------------------------------------------
_LIVE_PLANS = weakref.WeakSet()  # every ShtPlan / WavPlan with a live handle, whoever owns it (prior, user operator, ...)
_LIVE_LOCK = threading.Lock()     # plans may be created on one thread while a sampler on another polls the registry


def _register_plan(plan):
    with _LIVE_LOCK:
        _LIVE_PLANS.add(plan)


def live_plans():
    """the plans of this process that still hold a device handle: what a sampler polls for expired bounded waits"""
    with _LIVE_LOCK:
        plans = list(_LIVE_PLANS)
    return [pl for pl in plans if getattr(pl, "_h", None)]


class ShtPlan:
    """MW spin spherical-harmonic transforms at bandlimit L (replaces the pyssht calls)."""

    def __init__(self, L, spin=0, max_chains=1):
        require_gpu()
        self.L, self.spin, self.max_chains = int(L), int(spin), int(max_chains)
        self.npix, self.nlm = L * (2 * L - 1), L * L
        h = C.c_void_p()
        check(lib.pxm_sht_plan_create(self.L, self.spin, self.max_chains, 0, C.byref(h)))
        self._h = h
        _register_plan(self)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and lib is not None:  # lib is None during interpreter shutdown
            lib.pxm_sht_plan_destroy(h)
            self._h = None

    def _run(self, fn, x, n_in, n_out):
        x, squeeze = _batched(as_device(x, _CPLX))
        if x.shape[1] != n_in:
            raise AssertionError(f"expected length {n_in}, got {x.shape[1]}")
        if x.shape[0] > self.max_chains:
            raise ValueError("more chains than the plan was created for")
        out = torch.empty((x.shape[0], n_out), dtype=_CPLX, device=x.device)
        check(fn(self._h, _p(x), _p(out), x.shape[0], _stream()))
        return out[0] if squeeze else out

    def inverse(self, flm):
        return self._run(lib.pxm_sht_inverse, flm, self.nlm, self.npix)

    def forward(self, f):
        return self._run(lib.pxm_sht_forward, f, self.npix, self.nlm)

    def inverse_adjoint(self, f):
        return self._run(lib.pxm_sht_inverse_adjoint, f, self.npix, self.nlm)

    def forward_adjoint(self, flm):
        return self._run(lib.pxm_sht_forward_adjoint, flm, self.nlm, self.npix)

    def table_bytes(self, op):
        return int(lib.pxm_sht_table_bytes(self._h, op))

    def uses_recursion(self):
        """0, or 16 * ring blocks per wavefront + complex columns per order when inverse / inverse_adjoint take the
        table-free recursion kernels (csrc/sht_rec.hip) instead of the ring-table GEMM"""
        return int(check(lib.pxm_sht_uses_recursion(self._h)))

    def status(self, clear=False):
        """bit mask of the bounded device waits of this plan that expired (0 = none); synchronises"""
        return int(check(lib.pxm_sht_status(self._h, int(bool(clear)), _stream())))

    def raise_on_fault(self, clear=True):
        raise_on_status(self.status(clear=clear), f"ShtPlan(L={self.L}, spin={self.spin})")


class WavPlan:
    """Axisymmetric scale-discretised wavelet transforms (replaces the pys2let calls)."""

    def __init__(self, L, B, J_min, max_chains=1):
        require_gpu()
        self.L, self.B, self.J_min, self.max_chains = int(L), float(B), int(J_min), int(max_chains)
        self.npix = L * (2 * L - 1)
        nscal = C.c_int64()
        self.ncoefs = int(check(lib.pxm_wav_ncoefs(self.L, self.B, self.J_min, C.byref(nscal))))
        self.nscal = int(nscal.value)
        h = C.c_void_p()
        check(lib.pxm_wav_plan_create(self.L, self.B, self.J_min, self.max_chains, 0, C.byref(h)))
        self._h = h
        _register_plan(self)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and lib is not None:
            lib.pxm_wav_plan_destroy(h)
            self._h = None

    def _run(self, fn, x, n_in, n_out, out=None):
        x, squeeze = _batched(as_device(x, _CPLX))
        if x.shape[1] != n_in:
            raise AssertionError(f"expected length {n_in}, got {x.shape[1]}")
        if x.shape[0] > self.max_chains:
            raise ValueError("more chains than the plan was created for")
        if out is None:
            out = torch.empty((x.shape[0], n_out), dtype=_CPLX, device=x.device)
        elif out.shape != (x.shape[0], n_out) or out.dtype != _CPLX or not out.is_contiguous():
            raise ValueError("out= buffer has the wrong shape / dtype / layout")
        check(fn(self._h, _p(x), _p(out), x.shape[0], _stream()))
        return out[0] if squeeze else out

    def synthesis(self, X, out=None):
        return self._run(lib.pxm_wav_synthesis, X, self.ncoefs, self.npix, out=out)

    def synthesis_adjoint(self, f):
        return self._run(lib.pxm_wav_synthesis_adjoint, f, self.npix, self.ncoefs)

    def analysis(self, f):
        return self._run(lib.pxm_wav_analysis, f, self.npix, self.ncoefs)

    def analysis_adjoint(self, X):
        return self._run(lib.pxm_wav_analysis_adjoint, X, self.ncoefs, self.npix)

    def gradg_step(self, X, preds, data, invcov, T, delta, lmda, noise=None, noise_complex=False, seed=0, chain0=0, it=0, out=None,
                   pairs=False, noise64=False):
        """Fused calc_gradg + proxf + chain_step (pxmcmc/mcmc.py:158-160) for the synthesis setting.
        ``pairs``: every complex slot of X carries two real chains (PXM_MODE_REAL_PAIRS)."""
        x, squeeze = _batched(as_device(X, _CPLX))
        p, _ = _batched(as_device(preds, _CPLX))
        if x.shape[1] != self.ncoefs or p.shape[1] != self.npix or p.shape[0] != x.shape[0]:
            raise AssertionError("gradg_step: shape mismatch")
        d = as_device(data, _CPLX).reshape(-1)
        ic = as_device(invcov).reshape(-1)
        if d.numel() != self.npix or ic.numel() != self.npix:
            raise ValueError("data / invcov length mismatch")
        Tv, Ts = _vecT(T, self.ncoefs, x.device)
        w, wc = _pair_noise_args(noise, x) if pairs else _noise_args(noise, x, noise_complex)
        if out is None:
            out = torch.empty_like(x)
        elif out.shape != x.shape or out.dtype != _CPLX or not out.is_contiguous() or out.data_ptr() == x.data_ptr():
            raise ValueError("out= buffer must be a distinct contiguous complex128 tensor of the state's shape")
        check(
            lib.pxm_wav_gradg_step(
                self._h, _p(x), _p(p), _p(d), _p(ic), int(ic.is_complex()), _p(Tv), Ts, float(delta), float(lmda),
                _p(w), wc | _nf(noise64), seed, chain0, it, _p(out), x.shape[0], _stream(),
            )
        )
        return out[0] if squeeze else out

    # ---- fused iteration for a diagonal inverse covariance (residual rings carried inside the plan) ----
    def _image_args(self, data, invcov):
        d = as_device(data, _CPLX).reshape(-1)
        ic = as_device(invcov).reshape(-1)
        if d.numel() != self.npix or ic.numel() != self.npix:
            raise ValueError("data / invcov length mismatch")
        return d, ic

    def image_init(self, preds, data, invcov):
        p, _ = _batched(as_device(preds, _CPLX))
        if p.shape[1] != self.npix or p.shape[0] > self.max_chains:
            raise AssertionError("image_init: shape mismatch")
        d, ic = self._image_args(data, invcov)
        check(lib.pxm_wav_image_init(self._h, _p(p), _p(d), _p(ic), int(ic.is_complex()), p.shape[0], _stream()))

    def image_step(self, X, data, invcov, T, delta, lmda, noise=None, noise_complex=False, seed=0, chain0=0, it=0,
                   out=None, preds_out=None, pairs=False, noise64=False):
        """calc_gradg + proxf + chain_step + forward of the new state; the residual rings of the current state come
        from the previous ``image_step`` / ``image_init`` on this plan."""
        x, squeeze = _batched(as_device(X, _CPLX))
        if x.shape[1] != self.ncoefs or x.shape[0] > self.max_chains:
            raise AssertionError("image_step: shape mismatch")
        d, ic = self._image_args(data, invcov)
        Tv, Ts = _vecT(T, self.ncoefs, x.device)
        w, wc = _pair_noise_args(noise, x) if pairs else _noise_args(noise, x, noise_complex)
        if out is None:
            out = torch.empty_like(x)
        elif out.shape != x.shape or out.dtype != _CPLX or not out.is_contiguous() or out.data_ptr() == x.data_ptr():
            raise ValueError("out= buffer must be a distinct contiguous complex128 tensor of the state's shape")
        if preds_out is None:
            preds_out = torch.empty((x.shape[0], self.npix), dtype=_CPLX, device=x.device)
        elif preds_out.shape != (x.shape[0], self.npix) or preds_out.dtype != _CPLX or not preds_out.is_contiguous():
            raise ValueError("preds_out= buffer has the wrong shape / dtype / layout")
        check(
            lib.pxm_wav_image_step(
                self._h, _p(x), _p(d), _p(ic), int(ic.is_complex()), _p(Tv), Ts, float(delta), float(lmda),
                _p(w), wc | _nf(noise64), seed, chain0, it, _p(out), _p(preds_out), x.shape[0], _stream(),
            )
        )
        return (out[0], preds_out[0]) if squeeze else (out, preds_out)

    # ---- ring-space MYULA step (identity measurement + uniform inverse covariance) ----
    def ring_set_data(self, data):
        d = as_device(data, _CPLX).reshape(-1)
        if d.numel() != self.npix:
            raise ValueError("data length mismatch")
        check(lib.pxm_wav_ring_set_data(self._h, _p(d), _stream()))

    def ring_init(self, X):
        x, _ = _batched(as_device(X, _CPLX))
        if x.shape[1] != self.ncoefs or x.shape[0] > self.max_chains:
            raise AssertionError("ring_init: shape mismatch")
        check(lib.pxm_wav_ring_init(self._h, _p(x), x.shape[0], _stream()))

    def ring_step(self, X, w, T, delta, lmda, noise=None, noise_complex=False, seed=0, chain0=0, it=0, out=None, pairs=False,
                  noise64=False):
        """calc_gradg + proxf + chain_step + forward for a uniform inverse covariance ``w``; the rings of the
        new state stay inside the plan (``ring_preds`` materialises forward(X) when it is observed)."""
        x, squeeze = _batched(as_device(X, _CPLX))
        if x.shape[1] != self.ncoefs or x.shape[0] > self.max_chains:
            raise AssertionError("ring_step: shape mismatch")
        Tv, Ts = _vecT(T, self.ncoefs, x.device)
        wn, wc = _pair_noise_args(noise, x) if pairs else _noise_args(noise, x, noise_complex)
        if out is None:
            out = torch.empty_like(x)
        elif out.shape != x.shape or out.dtype != _CPLX or not out.is_contiguous() or out.data_ptr() == x.data_ptr():
            raise ValueError("out= buffer must be a distinct contiguous complex128 tensor of the state's shape")
        w = complex(w)
        check(
            lib.pxm_wav_ring_step(
                self._h, _p(x), w.real, w.imag, _p(Tv), Ts, float(delta), float(lmda), _p(wn), wc | _nf(noise64), seed, chain0, it,
                _p(out), x.shape[0], _stream(),
            )
        )
        return out[0] if squeeze else out

    def ring_preds(self, C_, out=None):
        if out is None:
            out = torch.empty((C_, self.npix), dtype=_CPLX, device=device())
        check(lib.pxm_wav_ring_preds(self._h, _p(out), C_, _stream()))
        return out

    def table_bytes(self, op):
        return int(lib.pxm_wav_table_bytes(self._h, op))

    def flow_status(self):
        """0: every wait of the dataflow GEMM launches of this plan was satisfied (include/pxmcmc_amd.h); synchronises"""
        return int(check(lib.pxm_wav_flow_status(self._h, _stream())))

    def exact_dft_scales(self):
        """scales whose 511-point rings the fused step transforms with the exact-length unit (csrc/dft_pfa.h); 0 = Bluestein"""
        return int(check(lib.pxm_wav_exact_dft_scales(self._h)))

    def flow_enabled(self):
        """True when the ring-space step of this plan takes the dataflow launch (PXM_FLOW=1; known after ring_set_data)"""
        return bool(check(lib.pxm_wav_flow_enabled(self._h)))

    def status(self, clear=False):
        """bit mask of the bounded device waits of this plan that expired (0 = none; include/pxmcmc_amd.h); synchronises"""
        return int(check(lib.pxm_wav_status(self._h, int(bool(clear)), _stream())))

    def raise_on_fault(self, clear=True):
        """raise PxmError if a kernel of this plan reported an expired wait since the last check (``clear``: reset the word)"""
        raise_on_status(self.status(clear=clear), f"WavPlan(L={self.L})")

    # ---- weak-lensing measurement fused with the synthesis (pxm_wav_wl_*) ----
    def wl_attach(self, pix2data, weight, ndata):
        """pix2data: int32 [npix] pixel -> index in the masked data vector (< 0 masked) or None; weight: float64
        [ndata] (WeakLensing.inv_cov) or None.  The plan keeps the tensors alive."""
        if pix2data is not None:
            pix2data = pix2data.to(device=device(), dtype=torch.int32).contiguous()
            if pix2data.numel() != self.npix:
                raise ValueError("pix2data must have one entry per pixel")
        if weight is not None:
            weight = as_device(weight, _REAL).reshape(-1)
            if weight.numel() != int(ndata):
                raise ValueError("weight must have one entry per datum")
        self._wl = (pix2data, weight, int(ndata))
        check(lib.pxm_wav_wl_attach(self._h, _p(pix2data), _p(weight), int(ndata)))

    def wl_uses_recursion(self):
        """non-zero when the attached spin-2 stage runs the table-free recursion kernels (csrc/sht_rec.hip)"""
        return int(check(lib.pxm_wav_wl_uses_recursion(self._h)))

    def wl_forward(self, X, out=None):
        x, squeeze = _batched(as_device(X, _CPLX))
        if x.shape[1] != self.ncoefs or x.shape[0] > self.max_chains:
            raise AssertionError("wl_forward: shape mismatch")
        nd = self._wl[2]
        if out is None:
            out = torch.empty((x.shape[0], nd), dtype=_CPLX, device=x.device)
        check(lib.pxm_wav_wl_forward(self._h, _p(x), _p(out), x.shape[0], _stream()))
        return out[0] if squeeze else out

    def wl_adjoint(self, gamma, data=None, invcov=None, out=None):
        g, squeeze = _batched(as_device(gamma, _CPLX))
        nd = self._wl[2]
        if g.shape[1] != nd or g.shape[0] > self.max_chains:
            raise AssertionError("wl_adjoint: shape mismatch")
        d = ic = None
        if data is not None:
            d = as_device(data, _CPLX).reshape(-1)
            ic = as_device(invcov).reshape(-1)
            if d.numel() != nd or ic.numel() != nd:
                raise ValueError("data / invcov length mismatch")
        if out is None:
            out = torch.empty((g.shape[0], self.ncoefs), dtype=_CPLX, device=g.device)
        check(lib.pxm_wav_wl_adjoint(self._h, _p(g), _p(d), _p(ic), int(ic.is_complex()) if ic is not None else 0, _p(out),
                                     g.shape[0], _stream()))
        return out[0] if squeeze else out

    def workspace_nonfinite(self):
        """test aid: non-finite values anywhere in the plan's workspace (padding chains' columns included)"""
        return int(check(lib.pxm_wav_workspace_nonfinite(self._h, _stream())))

    # ---- live kernel timing of this plan (bench.py roofline leg) ----
    def profile_enable(self, max_launches):
        check(lib.pxm_wav_profile_enable(self._h, int(max_launches)))

    def profile_read_launches(self, cap):
        """per-launch (ms, algorithmic bytes, workgroups) of the bracketed ring-GEMM launches, in launch order"""
        import numpy as np

        ms, nb, wg, n = np.zeros(cap), np.zeros(cap), np.zeros(cap, dtype=np.int32), C.c_int64()
        check(lib.pxm_wav_profile_read_launches(self._h, ms.ctypes.data, nb.ctypes.data, wg.ctypes.data, int(cap), C.byref(n)))
        k = min(int(n.value), cap)
        return ms[:k], nb[:k], wg[:k]

    def profile_read(self):
        """(gemm: ms, launches, algorithmic bytes, flops), (grouped phi-DFT: ms, launches, algorithmic bytes)"""
        ms, nl, nb, nf = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
        check(lib.pxm_wav_profile_read(self._h, C.byref(ms), C.byref(nl), C.byref(nb), C.byref(nf)))
        dms, dnl, dnb = C.c_double(), C.c_int64(), C.c_double()
        check(lib.pxm_wav_profile_read_dft(self._h, C.byref(dms), C.byref(dnl), C.byref(dnb)))
        return (ms.value, nl.value, nb.value, nf.value), (dms.value, dnl.value, dnb.value)


# ---- device-resident iteration counter (HIP-graph replay) -----------------------------------
class IterCounter:
    """A device int64 registered as the Philox iteration counter of ONE wavelet plan for the lifetime of the
    object.  A plan holds one live counter: a second engine on the same plan (two samplers built on one
    ForwardOperator) raises ``PxmError`` until the first has stopped; ``close`` only releases the registration if
    it is still this object's."""

    def __init__(self, plan, start=0):
        self.plan = plan
        self.t = torch.full((1,), int(start), dtype=torch.int64, device=device())
        check(lib.pxm_wav_set_iter_counter(plan._h, C.c_void_p(self.t.data_ptr())))
        self.active = True

    def set(self, value):
        self.t.fill_(int(value))

    def add(self, inc=1):
        check(lib.pxm_wav_iter_counter_add(self.plan._h, int(inc), _stream()))

    def close(self):
        if self.active:
            if getattr(self.plan, "_h", None):
                lib.pxm_wav_release_iter_counter(self.plan._h, C.c_void_p(self.t.data_ptr()))
            self.active = False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class capture_scope:
    """Tells the library that a stream capture is (about to be) in progress: plan teardown inside the scope only
    queues its frees (include/pxmcmc_amd.h, pxm_capture_begin / pxm_capture_end)."""

    def __enter__(self):
        check(lib.pxm_capture_begin())
        return self

    def __exit__(self, *exc):
        lib.pxm_capture_end()
        return False


def tables_trim():
    """free every cached ring table no live plan holds (the cache is per device and shared by plans); MiB released"""
    return int(check(lib.pxm_tables_trim()))


def noise_bits():
    """32: the precision of the Box-Muller step of the device noise stream WITHOUT the flag (the samplers of mcmc.py pass
    the flag by default: ``noise_bits=64``); every noise-drawing call takes
    ``noise64=True`` for the double-precision evaluation (PXM_NOISE_F64, include/pxmcmc_amd.h)"""
    return int(lib.pxm_noise_bits())


# ---- host helpers ------------------------------------------------------------------------
def j_max(L, B):
    return int(check(lib.pxm_j_max(int(L), float(B))))


def wav_bandlimits(L, B, J_min):
    buf = (C.c_int * 64)()
    n = check(lib.pxm_wav_bandlimits(int(L), float(B), int(J_min), buf, 64))
    return [int(buf[i]) for i in range(n)]


def tiling_axisym(L, B, J_min):
    J = j_max(L, B)
    k0 = np.zeros(L)
    k = np.zeros((J + 1, L))
    check(lib.pxm_tiling_axisym(int(L), float(B), int(J_min), k0.ctypes.data, k.ctypes.data))
    return k0, k


def mw_ring_weights(L):
    q = np.zeros(L)
    check(lib.pxm_mw_ring_weights(int(L), q.ctypes.data))
    return q
