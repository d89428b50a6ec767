#!/usr/bin/env python
"""
bench.py -- MYULA samples/sec at L=256 synthesis (BASELINE.json metric).

A "step" is one MYULA iteration (calc_gradg -> proxf -> chain_step -> forward,
pxmcmc/mcmc.py:158-161) of a batch of 16 chains per GPU: spherical-wavelet synthesis
operator (L=256, B=2, J_min=2), identity measurement, S2_Wavelets_L1 prox, complex128 state
(the reference's layout), synthetic band-limited data already resident in HBM.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Started plainly with --gpus N > 1 (no torchrun environment) the script launches its own N ranks as a child
``python -m torch.distributed.run`` BEFORE it imports torch or touches the GPU, relays rank 0's JSON line and
returns the child's exit code.

Prints ONE JSON line on rank 0 with the throughput, the roofline of the dominant kernel
(the SHT ring GEMM, timed live with HIP events on its own stream) and the CPU baseline
(the oracle's numpy restatement of the same iteration, one chain, on this host).
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

L, B, J_MIN, CHAINS_PER_GPU = 256, 2.0, 2, 16
LMDA, MU, SIGMA = 1e-6, 1.0, 0.05  # lmda: the reference's topography value (experiments/earthtopography/main.py:128)


def stable_delta(transform, sigma, lmda, iters=30):
    """MYULA step size delta = 0.8 / (L_f + 1/lmda) (Durmus, Moulines & Pereyra 2018), L_f = ||S||^2 / sigma^2 the
    Lipschitz constant of the data-fidelity gradient, ||S||^2 by power iteration on S^H S.  (delta = lmda / 2 is
    beyond this bound at sigma = 0.05 -- ||S||^2 = 1.35e4 at L = 256 -- and the chain diverges after ~600
    iterations.)"""
    import torch

    g = torch.Generator().manual_seed(0)
    x = torch.randn(transform.ncoefs, dtype=torch.complex128, generator=g).cuda()
    lam = 0.0
    for _ in range(iters):
        y = transform.inverse_adjoint(transform.inverse(x))
        lam = float(torch.linalg.norm(y) / torch.linalg.norm(x))
        x = y / torch.linalg.norm(y)
    return 0.8 / (lam / sigma ** 2 + 1.0 / lmda), lam
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)
FP64_VECTOR_SPEC_TFLOPS = 78.6   # dense fp64 vector peak of the part
FP64_VECTOR_FMA_TFLOPS = 62.3    # what a v_fma_f64-only loop sustains (scripts/probes/mfma_valu_mix.hip, rec_inst_rates.hip)
MFMA_SUSTAINED_TFLOPS = 47.1  # fp64 MFMA-only loop on this part (scripts/probes/mfma_rate.hip; spec 78.6)


def synthetic_field(plan_inverse, L, seed, slope=-1.0):
    """random real band-limited field, amplitude (1+l)^slope (C_l ~ (1+l)^-2 at slope -1), unit RMS."""
    rng = np.random.default_rng(seed)
    flm = np.zeros(L * L, dtype=complex)
    for el in range(L):
        amp = (1.0 + el) ** slope
        flm[el * el + el] = amp * rng.normal()
        m = np.arange(1, el + 1)
        v = amp * (rng.normal(size=el) + 1j * rng.normal(size=el)) / np.sqrt(2)
        flm[el * el + el + m] = v
        flm[el * el + el - m] = (-1.0) ** m * np.conj(v)
    f = plan_inverse(flm).real
    return f / np.sqrt(np.mean(f ** 2)), rng


_CPU = {}  # oracle operator of the CPU legs: built once in this process, inherited by forked workers


def _cpu_operator(data):
    from oracle import pxmcmc_np as ref

    if "op" not in _CPU:
        tr = ref.SphericalWaveletTransform(L, int(B), J_MIN)
        P = data.size
        _CPU["op"] = ref.ForwardOperator(data, SIGMA, "synthesis", tr, ref.Identity(P, P), tr.ncoefs)
    return _CPU["op"]


def _cpu_chain(args):
    """n_iter literal MYULA iterations (pxmcmc/mcmc.py:158-161) of ONE chain with the oracle; returns seconds"""
    from threadpoolctl import threadpool_limits

    from oracle import pxmcmc_np as ref

    T, n_iter, delta, seed = args
    op = _CPU["op"]
    with threadpool_limits(limits=1):
        X = np.zeros(op.nparams, dtype=complex)
        preds = op.forward(X)
        rng = np.random.default_rng(seed)
        t0 = time.perf_counter()
        for _ in range(n_iter):
            gradg = op.calc_gradg(preds)
            px = ref.soft(X, T)
            X = ref.chain_step(X, px, gradg, delta, LMDA, rng.normal(size=op.nparams))
            preds = op.forward(X)
        dt = time.perf_counter() - t0
    assert np.isfinite(X).all()
    return dt


def cpu_baseline(data, T, n_iter, delta, procs=1):
    """The oracle's literal MYULA iteration on this host -- baseline only.  ``procs`` == 1: one chain, BLAS / OpenMP
    pools pinned to one thread.  ``procs`` > 1: one chain per PROCESS (the reference's own way to use cores:
    ``--jobid``, experiments/earthtopography/main.py:31-36), forked from this process so the read-only tables are
    shared; throughput = chain-iterations of all processes / wall time."""
    _cpu_operator(data)
    if procs == 1:
        dt = _cpu_chain((T, n_iter, delta, 0))
        return n_iter / dt, dt
    import multiprocessing as mp

    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(procs) as pool:  # (children only run numpy; they never touch the GPU runtime)
        pool.map(_cpu_chain, [(T, n_iter, delta, k) for k in range(procs)])
    wall = time.perf_counter() - t0
    return procs * n_iter / wall, wall


def self_launch(n, argv, script=None):
    """Start the N ranks of ``bench.py --gpus N`` as a child ``python -m torch.distributed.run`` (one rank per GPU,
    rendezvous on 127.0.0.1 at a free port).  Called before torch is imported: the parent holds no GPU state, the
    child is an ordinary subprocess (never an exec), rank 0 prints the JSON line on the inherited stdout.
    (``script``: another program for the ranks -- the CPU rehearsal of the 8-rank rendezvous in tests/test_distributed.py.)"""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), script or os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("OMP_NUM_THREADS", "1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


NOISE_NOTES = {
    32: "philox4x32-10 counter stream, Box-Muller on the f32 transcendental units (v_log_f32 / v_sin_f32 / v_cos_f32, exact "
        "fp64 exponent: deviates ~1e-6 relative, tail to 8.5 sigma); the reference draws fp64 randn (pxmcmc/mcmc.py:193)",
    64: "philox4x32-10 counter stream, Box-Muller in fp64 (branch-free log / sqrt / sincos polynomials in double, 1e-15 vs numpy: "
        "flag PXM_NOISE_F64 of the stepping calls, MYULA(noise_bits=64))",
}

NOMINAL_S_NORM2 = 1.35e4  # ||S||^2 at L=256, B=2, J_min=2 (power iteration on the GPU at start-up: checked below)


def cpu_legs(args):
    """Both CPU legs, run BEFORE this process imports torch or initialises the GPU (the worker pool is forked from a
    process that holds no HIP runtime state).  The oracle operator is built from host data: the same seeded field as
    the GPU leg, synthesised by the oracle's own inverse SHT (equal to the device-made field to round-off; the timing
    does not depend on the values).  Returns the ``cpu_baseline`` object of the JSON line."""
    from oracle import pxmcmc_np as ref
    from oracle import ssht

    truth, rng = synthetic_field(lambda flm: ssht.inverse(flm, L, 0).ravel(), L, seed=2)
    data = truth + SIGMA * rng.normal(size=truth.size)
    T = ref.S2_Wavelets_L1("synthesis", None, None, LMDA * MU, L, int(B), J_MIN).T
    delta = 0.8 / (NOMINAL_S_NORM2 / SIGMA ** 2 + 1.0 / LMDA)
    v, secs = cpu_baseline(data, T, args.cpu_iters, delta, procs=1)
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    procs = args.cpu_procs or min(ncpu, 32)
    it_all = max(8, args.cpu_iters // 3)
    v_all, secs_all = cpu_baseline(data, T, it_all, delta, procs=procs)
    return {
        "value": v,
        "unit": "samples/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{args.cpu_iters} MYULA iterations of ONE chain at L=256 (oracle numpy restatement, "
                  f"table+FFT SHT, BLAS / OpenMP pools pinned to 1 thread, {secs:.1f} s) on this host, before the GPU is touched",
        "value_all_cores": v_all,
        "cores_all": procs,
        "cores_rule": "min(usable cores, 32): the host share of one GPU on an 8-GPU node with 256 cores",
        "sample_all_cores": f"{procs} independent chains, one single-threaded process each (the reference's --jobid "
                            f"model), {it_all} iterations per chain, {secs_all:.1f} s wall incl. process start; "
                            f"host reports {ncpu} usable cores",
    }


def parity_leg(plan, data, T_dev, T, delta, n_iter=3, chains=(0, 9), C=CHAINS_PER_GPU):
    """Full-size parity on the box the benchmark runs on: n_iter iterations of the benchmarked step (ring-space +
    Gram + real pairs) with injected noise against the oracle's literal loop on the same noise.  Returns the max
    error relative to max |X|."""
    import torch

    from oracle import pxmcmc_np as ref
    from pxmcmc_amd import ops

    op = _cpu_operator(data)
    N = op.nparams
    rng = np.random.default_rng(123)
    X0 = rng.normal(size=(C, N)) * 1e-3
    noise = rng.normal(size=(n_iter, C, N))
    d = ops.as_device(data, torch.float64)
    X = torch.complex(ops.as_device(X0[0::2]), ops.as_device(X0[1::2]))
    out = torch.empty_like(X)
    w = complex(1.0 / SIGMA ** 2)
    plan.ring_set_data(torch.complex(d, d).contiguous())
    plan.ring_init(X)
    for k in range(n_iter):
        plan.ring_step(X, w, T_dev, delta, LMDA, noise=ops.as_device(noise[k]), out=out, pairs=True)
        X, out = out, X
    Xg = X.cpu().numpy()
    err = 0.0
    for c in chains:
        Xo = X0[c].astype(complex)
        preds = op.forward(Xo)
        for k in range(n_iter):
            Xo = ref.chain_step(Xo, ref.soft(Xo, T), op.calc_gradg(preds), delta, LMDA, noise[k][c])
            preds = op.forward(Xo)
        got = Xg[c // 2].real if c % 2 == 0 else Xg[c // 2].imag
        err = max(err, float(np.abs(got - Xo.real).max() / np.abs(Xo).max()))
    return err


def pmc_passes(child_args, timeout_s=150):
    """HBM bytes per k_sht_gemm launch class, measured NOW: two child runs of this script (``child_args``) under
    ``rocprofv3 --pmc`` (FETCH_SIZE, then WRITE_SIZE: separate passes, MI355X_MICROARCH.md section HBM), started before
    this process touches the GPU.  KiB units; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads
    (doubled here), WRITE_SIZE is taken as is.  Returns ({(kernel, workgroups): {...}}, seconds) or (None, reason)."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile

    rocprof = shutil.which("rocprofv3")
    if not rocprof:
        return None, "rocprofv3 not on PATH"
    vals = {}
    t0 = time.perf_counter()
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="pxm_pmc_", dir="/tmp")
        cmd = [rocprof, "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__)] + child_args
        # own session: on a timeout the whole group (rocprofv3 AND the profiled python) is killed and waited for
        # before this process goes near the GPU
        proc = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL,
                                stderr=subprocess.DEVNULL, start_new_session=True)
        try:
            rc = proc.wait(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            with contextlib.suppress(ProcessLookupError):
                os.killpg(proc.pid, signal.SIGKILL)
            proc.wait()
            time.sleep(1.0)
            shutil.rmtree(d, ignore_errors=True)
            return None, f"rocprofv3 --pmc {counter} pass exceeded {timeout_s} s"
        files = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))
        per = {}
        if rc == 0 and files:
            with open(files[0]) as fh:
                for row in csv.DictReader(fh):
                    # every ring-GEMM launch, keyed by launch class = (kernel variant, workgroups)
                    # (k_sht_gemm<..>, k_sht_gemm_pk<..>: ring-table GEMMs; k_rec_e2r / k_rec_r2e: table-free recursion stage)
                    if row["Counter_Name"] == counter and row["Kernel_Name"].startswith(("void pxm::k_sht_gemm", "void pxm::k_rec_e2r", "void pxm::k_rec_r2e")):
                        wgs = int(row["Grid_Size"]) // max(int(row["Workgroup_Size"]), 1)
                        per.setdefault((row["Kernel_Name"].split("(")[0], wgs), []).append(float(row["Counter_Value"]))
        shutil.rmtree(d, ignore_errors=True)
        if not per:
            return None, f"rocprofv3 --pmc {counter} pass gave no k_sht_gemm records (exit {rc})"
        vals[counter] = per
    classes = {}
    for key, rd_list in vals["FETCH_SIZE"].items():
        wr_list = vals["WRITE_SIZE"].get(key)
        if wr_list is None:
            continue
        rd = 2 * 1024 * sum(rd_list) / len(rd_list)
        wr = 1024 * sum(wr_list) / len(wr_list)
        classes[key] = {"kernel": key[0].replace("void pxm::", ""), "workgroups": key[1], "launches": len(rd_list),
                        "read_MB": rd / 1e6, "write_MB": wr / 1e6, "hbm_MB": (rd + wr) / 1e6}
    return classes, time.perf_counter() - t0


def live_traffic(timeout_s=120):
    """PMC traffic of the timed iteration's k_sht_gemm launches: child passes of the headline step with --steps 10.
    Returns (bytes per launch, description, per-launch-class list) or (None, reason, None)."""
    classes, info = pmc_passes(["--steps", "10", "--warmup", "2", "--ramp", "0", "--no-cpu-baseline", "--no-layout-compare",
                                "--no-live-traffic", "--no-config-legs", "--no-noise-leg"], timeout_s)
    if classes is None:
        return None, info, None
    # classes of the stepping loop: the ones launched (almost) once per step of the 12-step child
    classes = {k: c for k, c in classes.items() if c["launches"] >= 10}
    if not classes:
        return None, "rocprofv3 --pmc passes gave no per-step k_sht_gemm launch class", None
    n = sum(c["launches"] for c in classes.values())
    total = sum(c["hbm_MB"] * c["launches"] for c in classes.values()) * 1e6 / n
    return total, (f"measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of this command with --steps 10 "
                   f"({n} k_sht_gemm launches of {len(classes)} per-step launch classes, FETCH_SIZE x2 gfx950 correction, "
                   f"{info:.0f} s)"), sorted(classes.values(), key=lambda c: c["workgroups"])


# ---- side legs: BASELINE configs[1] (L=64 topography, 1 chain) and configs[4] (L=512 weak lensing, PxMALA, 1 chain) ----
C2_L, C2_B, C2_JMIN = 64, 1.5, 2      # experiments/earthtopography/main.py:72-74
C5_L, C5_B, C5_JMIN = 512, 2, 2       # experiments/weaklensing/main.py:85-87
C5_DELTA0, C5_NGAL = 1e-6, 30.0       # SURVEY.md section 8d C5 (main.py:92: ngal = 30)


def launch_classes(plan, cap):
    """ring-GEMM launch classes of the launches bracketed since profile_enable: (workgroups, algorithmic bytes) -> avg us"""
    l_ms, l_bytes, l_wgs = plan.profile_read_launches(cap)
    out = []
    for wgs, nbytes in sorted(set(zip(l_wgs.tolist(), np.round(l_bytes).tolist()))):
        sel = (l_wgs == wgs) & (np.round(l_bytes) == nbytes)
        us = float(l_ms[sel].mean() * 1e3)
        out.append({"workgroups": int(wgs), "alg_MB": nbytes / 1e6, "launches": int(sel.sum()), "avg_us": us,
                    "alg_GBs": nbytes / us / 1e3, "alg_frac": nbytes / us / 1e3 / HBM_PEAK_GBS})
    return out


PMC_CHILD_REPS = {"config2": 12, "config5": 6}  # loop trips of the side legs' rocprofv3 --pmc child passes (pmc_child)


def join_pmc(classes, pmc, min_launches=1):
    """PMC bytes of the child passes onto the event-timed launch classes (key: workgroup count) -> measured HBM rate.
    ``min_launches``: only kernel variants launched at least that often in the child pass are loop launches (the set-up
    transforms of the problem share grid sizes with them but run once or twice)."""
    by_wgs = {}
    for c in (pmc or {}).values():
        if c["launches"] >= min_launches:
            by_wgs.setdefault(c["workgroups"], []).append(c)
    for g in classes:
        cs = by_wgs.get(g["workgroups"])
        if cs and g.get("kernel_match"):  # (two kernels of one grid size told apart by name: the recursion pair)
            cs = [c for c in cs if g["kernel_match"] in c["kernel"]] or None
        if cs and max(c["hbm_MB"] for c in cs) > 1.02 * min(c["hbm_MB"] for c in cs):
            # several kernel variants of this grid size with different traffic in the child pass: no honest join
            g["pmc_ambiguous"] = sorted(c["kernel"] for c in cs)
            continue
        if cs:
            n = sum(c["launches"] for c in cs)
            mb = sum(c["hbm_MB"] * c["launches"] for c in cs) / n
            g.update(pmc_MB=mb, pmc_over_alg=mb / g["alg_MB"], hbm_GBs=mb / g["avg_us"] * 1e3,
                     hbm_frac=mb / g["avg_us"] * 1e3 / HBM_PEAK_GBS)
    return classes


def config2_problem():
    """BASELINE configs[1]: L=64, B=1.5, J_min=2, one chain, complex data as the reference's alm2map_mw returns it
    (experiments/earthtopography/main.py:82 => complex-variance rule, forward.py:81-82), C_l ~ (1+l)^-2."""
    from pxmcmc_amd import ops
    from pxmcmc_amd.forward import SphericalWaveletTransformOperator
    from pxmcmc_amd.mcmc import MYULA, PxMCMCParams
    from pxmcmc_amd.prior import S2_Wavelets_L1

    sht = ops.ShtPlan(C2_L, 0, max_chains=1)
    truth, rng = synthetic_field(lambda flm: sht.inverse(flm).cpu().numpy(), C2_L, seed=1, slope=-2.0)
    del sht
    data = (truth + SIGMA * rng.normal(size=truth.size)).astype(complex)
    op = SphericalWaveletTransformOperator(data, SIGMA, "synthesis", C2_L, C2_B, C2_JMIN, max_chains=1)
    reg = S2_Wavelets_L1("synthesis", op.transform.inverse, op.transform.inverse_adjoint, LMDA * MU, L=C2_L, B=C2_B, J_min=C2_JMIN)
    delta, _ = stable_delta(op.transform, SIGMA, LMDA)
    params = PxMCMCParams(lmda=LMDA, delta=delta, mu=MU, nsamples=1, nburn=0, ngap=1, verbosity=0)
    s = MYULA(op, reg, params, nchains=1, rng="philox", seed=1, noise_bits=64)
    s._prepare()
    with contextlib.redirect_stdout(io.StringIO()):
        X, preds = s._initial_sample(np.zeros(op.nparams))
    return s, X, preds


def config2_leg(steps=400, warm=100, n_prof=50, pmc=None):
    import torch

    s, X, preds = config2_problem()
    assert s._fused_wav and not s._pairs_ok(X)
    eng = s._engine_start(X, preds, 0)
    s._engine_advance(warm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s._engine_advance(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    plan = eng["plan"]
    graphs = eng["graph"], eng["graph_long"]
    plan.profile_enable(3 * n_prof + 8)
    eng["graph"] = eng["graph_long"] = None
    s._engine_advance(n_prof)
    eng["graph"], eng["graph_long"] = graphs
    torch.cuda.synchronize()
    classes = join_pmc(launch_classes(plan, 3 * n_prof + 8), pmc, min_launches=PMC_CHILD_REPS["config2"] - 2)
    plan.profile_enable(0)
    Xs, _ = s._engine_state()
    finite = bool(torch.isfinite(Xs.real).all())
    s._engine_stop()
    return {"workload": f"MYULA, wavelet synthesis L={C2_L} B={C2_B} J_min={C2_JMIN} (N=28390, P=8128), identity measurement, 1 chain, "
                        "complex data (reference-literal: complex-variance rule), ring-space + Gram step, HIP graph",
            "iterations": steps, "ms_per_iteration": dt / steps * 1e3, "samples_per_s": steps / dt, "finite": finite,
            "gemm_launch_classes": classes}


def config5_mask(L):
    """equatorial band |90 deg - theta| < 10 deg plus a great-circle band tilted by 60 deg (the
    stand-in for utils.build_mask's galactic + ecliptic cuts, pxmcmc/utils.py:320-349)"""
    theta = np.pi * (2 * np.arange(L) + 1) / (2 * L - 1)
    phi = 2 * np.pi * np.arange(2 * L - 1) / (2 * L - 1)
    st, ct = np.sin(theta)[:, None], np.cos(theta)[:, None]
    x, y, z = st * np.cos(phi)[None, :], st * np.sin(phi)[None, :], ct * np.ones_like(phi)[None, :]
    t = np.radians(60.0)
    lat2 = np.degrees(np.arcsin(np.clip(np.cos(t) * z + np.sin(t) * y, -1, 1)))  # latitude about the tilted pole
    mask = np.ones((L, 2 * L - 1), dtype=int)
    mask[np.abs(90 - np.degrees(theta)) < 10, :] = 0
    mask[np.abs(lat2) < 10] = 0
    return mask


def config5_problem():
    """BASELINE configs[4] on one GPU: L=512 weak-lensing shear operator + wavelet synthesis, one chain"""
    from pxmcmc_amd.forward import ForwardOperator
    from pxmcmc_amd.measurements import WeakLensing
    from pxmcmc_amd.prior import S2_Wavelets_L1
    from pxmcmc_amd.transforms import SphericalWaveletTransform

    L = C5_L
    rng = np.random.default_rng(3)
    tr = SphericalWaveletTransform(L, C5_B, C5_JMIN, max_chains=1)
    mask = config5_mask(L)
    wl = WeakLensing(L, mask, ngal=np.full(mask.shape, C5_NGAL), max_chains=1)
    # convergence: Gaussian flm with C_l ~ (1+l)^-1.5 exp(-(l/200)^2), klm[:4] = 0, real field
    flm = np.zeros(L * L, dtype=complex)
    for el in range(2, L):
        amp = np.sqrt((1.0 + el) ** -1.5 * np.exp(-((el / 200.0) ** 2)))
        flm[el * el + el] = amp * rng.normal()
        m = np.arange(1, el + 1)
        v = amp * (rng.normal(size=el) + 1j * rng.normal(size=el)) / np.sqrt(2)
        flm[el * el + el + m] = v
        flm[el * el + el - m] = (-1.0) ** m * np.conj(v)
    kappa = wl._sht0.inverse(flm)
    kappa = kappa / kappa.abs().max() * 0.05
    gamma = wl.forward(kappa)  # covariance-weighted shear: unit-variance noise in these units
    data = gamma.cpu().numpy() + (rng.normal(size=wl.ndata) + 1j * rng.normal(size=wl.ndata)) / np.sqrt(2)
    op = ForwardOperator(data, 1 / wl.inv_cov, "synthesis", transform=tr, measurement=wl, nparams=tr.ncoefs)
    lmda = C5_DELTA0 / 2
    reg = S2_Wavelets_L1("synthesis", tr.inverse, tr.inverse_adjoint, lmda * MU, L=L, B=C5_B, J_min=C5_JMIN)
    return op, reg, tr, wl, lmda


def config5_operator_loop(op, nrep, seed=0):
    """nrep x (forward, calc_gradg) of the fused wavelet + weak-lensing operator: the four ring GEMMs of a PxMALA iteration"""
    import torch

    g = torch.Generator().manual_seed(seed)
    Xd = torch.randn(1, op.nparams, dtype=torch.float64, generator=g).cuda() * 1e-3
    for _ in range(nrep):
        op.calc_gradg(op.forward(Xd))
    torch.cuda.synchronize()


def rec_stage_flops(L, spin, ncols):
    """fp64 operations of one launch of the table-free ring stage (csrc/sht_rec.hip): per (ring, el, order) two fused
    multiply-adds of the recursion and two per complex column"""
    steps = sum(L - max(abs(m), abs(spin)) for m in range(-(L - 1), L) if max(abs(m), abs(spin)) < L)
    return steps * L * (2 + 2 * ncols) * 2.0


C5_TUNE_CAP, C5_WINDOW, C5_TIMED, C5_LAP = 6000, 200, 150, 50   # tuned PxMALA leg: iteration cap, acceptance window, timed stretch, lap
C5_ACC_LO, C5_ACC_HI = 0.3, 0.7


def tuned_iteration(acc, window=C5_WINDOW, lap=C5_LAP, timed=C5_TIMED, lo=C5_ACC_LO, hi=C5_ACC_HI):
    """First iteration count n (a multiple of ``lap``, n >= window, n + timed <= len(acc)) at which the acceptance rate of
    the last ``window`` iterations lies in [lo, hi]: the adaptation of pxmcmc/mcmc.py:277-279 has brought delta to its
    working value.  Returns (n, window acceptance, True) or (len(acc) - timed rounded down to a lap, its window acceptance,
    False) when the trace never gets there."""
    acc = np.asarray(acc, dtype=float)
    last = (len(acc) - timed) // lap * lap
    for n in range((window + lap - 1) // lap * lap, last + 1, lap):
        a = float(acc[n - window:n].mean())
        if lo <= a <= hi:
            return n, a, True
    n = max(last, 0)
    return n, float(acc[max(n - window, 0):n].mean()) if n else 0.0, False


def config5_tuned(op, reg, tr, lmda, chain_offset=0, cap=C5_TUNE_CAP):
    """The PxMALA chain of BASELINE configs[4] as a user runs it (experiments/weaklensing/main.py:110-147): delta_0 = 1e-6
    from a zero start, ``tune_delta`` on (pxmcmc/mcmc.py:258-260, 277-279), ONE uninterrupted run of ``cap`` iterations with
    a device synchronise + host time stamp every C5_LAP iterations.  delta_0 is rejected throughout until the adaptation has
    shrunk it by ~6 decades (the data term moves by ~2 delta ||A||^2 per proposal); the timed stretch is the C5_TIMED iterations
    that follow the first lap at which the acceptance over the last C5_WINDOW iterations is inside [0.3, 0.7]."""
    import torch

    from pxmcmc_amd.mcmc import PxMALA, PxMCMCParams

    # (nburn beyond the run: no save candidates, i.e. no per-iteration host synchronisation; max_iter bounds the run --
    # the reference's loop only ends on accepted samples)
    p = PxMCMCParams(nsamples=1, nburn=10 ** 9, ngap=1, delta=C5_DELTA0, lmda=lmda, mu=MU, verbosity=0, track=[])
    s = PxMALA(op, reg, p, tune_delta=True, nchains=1, seed=3, chain_offset=chain_offset, max_iter=cap, lap_every=C5_LAP,
               noise_bits=64)
    with contextlib.redirect_stdout(io.StringIO()):
        s.run(start_point=np.zeros(tr.ncoefs))
    assert s.niter == cap
    acc = np.asarray(s.acceptance_trace, dtype=float)
    deltas = np.asarray(s.deltas_trace, dtype=float)
    laps = dict(s.laps)
    n0, a0, ok = tuned_iteration(acc)
    n1 = n0 + C5_TIMED
    ms = (laps[n1] - (laps[n0] if n0 else 0.0)) / C5_TIMED * 1e3
    first = laps[C5_TIMED] / C5_TIMED * 1e3
    return {"ms_per_iteration": ms, "samples_per_s": 1e3 / ms, "tuned": ok, "iterations_before_timed_stretch": int(n0),
            "window_acceptance_at_start": a0, "acceptance_in_timed_stretch": float(acc[n0:n1].mean()),
            "acceptance_after_tuning": float(acc[n0:].mean()), "delta_at_start_of_stretch": float(deltas[n0]),
            "delta_final": float(deltas[-1]), "delta_0": C5_DELTA0, "timed_iterations": C5_TIMED, "iterations_run": cap,
            "first_iterations": {"iterations": C5_TIMED, "ms_per_iteration": first, "acceptance": float(acc[:C5_TIMED].mean()),
                                 "what": "the first iterations of the same run: delta_0 = 1e-6 is rejected throughout, no "
                                         "conditional copy of an accepted state in the clock"},
            "hip_graph": bool(s.used_graph), "finite": bool(torch.isfinite(s.X_curr.real).all()),
            "loop_ms_per_iteration_all": s.loop_seconds / cap * 1e3,
            "clock": f"host time stamps after a device synchronise every {C5_LAP} iterations of one run "
                     "(PxMALA(lap_every=...)); the stretch is the difference of two stamps"}


def config5_leg(nrep=10, pmc=None, cap=C5_TUNE_CAP):
    import torch

    t_setup = time.perf_counter()
    op, reg, tr, wl, lmda = config5_problem()
    plan = op._wl_plan()
    assert plan is not None
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup
    tuned = config5_tuned(op, reg, tr, lmda, cap=cap)
    plan.profile_enable(4 * nrep + 8)
    config5_operator_loop(op, nrep)
    rec = bool(plan.wl_uses_recursion())
    classes = launch_classes(plan, 4 * nrep + 8)
    for c in classes:  # (ring -> el: 25 MB; el -> ring incl. its operand pass: 42 MB; both 512 workgroups)
        if rec and c["alg_MB"] < 100.0:
            c["kernel_match"] = "k_rec_r2e" if c["alg_MB"] < 30.0 else "k_rec_e2r"
    classes = join_pmc(classes, pmc, min_launches=PMC_CHILD_REPS["config5"])
    plan.profile_enable(0)
    for c in classes:  # the launches of the table-free spin-2 ring stage against BOTH rooflines
        if rec and c["alg_MB"] < 100.0:
            fl = rec_stage_flops(C5_L, 2, 1)
            c.update(kernel=c["kernel_match"] + (" (+ its operand pass k_rec_pack)" if c["kernel_match"] == "k_rec_e2r" else "") + ": Wigner rows by recursion, no table", fp64_GFLOP=fl / 1e9,
                     fp64_TFLOPs=fl / c["avg_us"] / 1e6, fp64_frac_of_spec=fl / c["avg_us"] / 1e6 / FP64_VECTOR_SPEC_TFLOPS,
                     fp64_frac_of_fma_rate=fl / c["avg_us"] / 1e6 / FP64_VECTOR_FMA_TFLOPS, bound="fp64 vector")
        else:
            c.update(kernel="k_sht_gemm_pk (packed column tile) / k_sht_gemm: per-scale spin-0 ring tables", bound="hbm")
    gemm_us = sum(c["avg_us"] * c["launches"] for c in classes) / nrep
    out = {"workload": f"PxMALA (tune_delta), wavelet synthesis L={C5_L} B={C5_B} J_min={C5_JMIN} (N=1221796) + weak-lensing shear "
                       f"measurement with a mask ({wl.ndata} of {wl.npix} pixels kept) and ngal = {C5_NGAL:.0f}, 1 chain, fused operator, "
                       "one-pass propose / accept kernels, HIP graph; timed in the TUNED state (acceptance of the last "
                       f"{C5_WINDOW} iterations inside [{C5_ACC_LO}, {C5_ACC_HI}])",
           "iterations": C5_TIMED}
    out.update(tuned)
    out.update({"acceptance": tuned["acceptance_in_timed_stretch"], "setup_s": t_setup,
                "ring_stage_us_per_iteration": gemm_us, "gemm_launch_classes": classes,
                "spin2_stage": "table-free recursion (csrc/sht_rec.hip)" if rec else "ring-table GEMM",
                "note": "four ring stages per iteration: two spin-0 group launches on the 8 + 1 wavelet scales (packed column tile, the two "
                        "512-band-limited scales in one pass over their table and one twin ring array) and two spin-2 stages (Wigner "
                        "rows by three-term recursion on the vector pipe: no table, 25 MB instead of 1.09 GB per launch)"})
    return out


def config5_multirank_leg(rank, world, D, cap=C5_TUNE_CAP):
    """BASELINE configs[4] as it is named: one PxMALA chain per GPU (the reference runs one chain per process,
    experiments/weaklensing/main.py:110-147).  Every rank builds the same problem, runs ITS chain (Philox keyed by the global
    chain id = rank) through ``config5_tuned`` on its own clock -- no torch.distributed call between the first and the last
    time stamp of a rank -- and the per-rank figures are all-gathered afterwards."""
    import torch

    err = 0.0
    res = {"ms_per_iteration": float("nan"), "acceptance_in_timed_stretch": float("nan"), "delta_at_start_of_stretch": float("nan"),
           "iterations_before_timed_stretch": -1, "tuned": False, "finite": False}
    try:
        op, reg, tr, wl, lmda = config5_problem()
        assert op._wl_plan() is not None
        torch.cuda.synchronize()
        res = config5_tuned(op, reg, tr, lmda, chain_offset=rank, cap=cap)
    except Exception as exc:  # (the gather below must still be entered by every rank)
        err = 1.0
        print(f"bench.py rank {rank}: configs[4] leg failed: {exc!r}", file=sys.stderr)
    ms = D.all_gather_float(res["ms_per_iteration"])
    out = {"workload": f"PxMALA (tune_delta), wavelet synthesis L={C5_L} B={C5_B} J_min={C5_JMIN} + weak-lensing shear measurement "
                       f"with a mask, ngal = {C5_NGAL:.0f}: ONE chain per GPU, {world} GPUs, chain id = rank, timed in the tuned state",
           "per_rank_ms_per_iteration": ms,
           "ms_per_iteration": max(ms),
           "samples_per_s": world * 1e3 / max(ms),
           "samples_per_s_sum_of_ranks": sum(1e3 / m for m in ms),
           "per_rank_acceptance": D.all_gather_float(res["acceptance_in_timed_stretch"]),
           "per_rank_delta": D.all_gather_float(res["delta_at_start_of_stretch"]),
           "per_rank_iterations_before_timed_stretch": [int(v) for v in D.all_gather_float(float(res["iterations_before_timed_stretch"]))],
           "per_rank_tuned": [bool(v) for v in D.all_gather_float(float(bool(res["tuned"])))],
           "timed_iterations": C5_TIMED, "iterations_run": cap,
           "ranks_seen": len(ms),
           "finite": all(bool(v) for v in D.all_gather_float(float(bool(res["finite"])))),
           "timing": "per rank: its own run, its own device synchronises and host time stamps; ms_per_iteration = max over "
                     "ranks, samples_per_s = ranks / that (the chains advance side by side)"}
    if sum(D.all_gather_float(err)) > 0:
        out["error"] = "a rank's configs[4] leg raised (see stderr)"
    return out


PARITY_TOL = 1e-9  # full-size parity leg: max |X_hip - X_oracle| / max |X| after 3 iterations


def leg_failures(out):
    """what must turn the exit code red although the headline line is printed: a side leg that raised or produced
    non-finite state, a parity leg beyond PARITY_TOL"""
    bad = []
    for name, leg in (out.get("configs") or {}).items():
        if isinstance(leg, dict) and "error" in leg:
            bad.append(f"{name}: {leg['error']}")
        elif isinstance(leg, dict) and leg.get("finite") is False:
            bad.append(f"{name}: non-finite state")
    par = out.get("parity")
    if par is not None:
        err = par.get("max_rel_err_X")
        if "error" in par or err is None or not (err <= PARITY_TOL):
            bad.append(f"parity: {par.get('error', err)} (tolerance {PARITY_TOL})")
    return bad


def pmc_child(which):
    """body of the rocprofv3 --pmc child passes of the side legs: the same launches, a few of each, nothing printed"""
    import torch

    if "config2" in which:
        s, X, preds = config2_problem()
        s.use_graph = False
        s._engine_start(X, preds, 0)
        s._engine_advance(PMC_CHILD_REPS["config2"])
        torch.cuda.synchronize()
        s._engine_stop()
    if "config5" in which:
        op, *_ = config5_problem()
        config5_operator_loop(op, PMC_CHILD_REPS["config5"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--ramp", type=int, default=400, help="untimed iterations between the literal W + K run (value_no_ramp) and the headline regions (clock ramp; declared in the JSON)")
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of K steps back to back; value = the median region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=48)
    ap.add_argument("--cpu-procs", type=int, default=0, help="processes of the all-core CPU leg (0 = min(usable cores, 32))")
    ap.add_argument("--no-real-pairs", action="store_true", help="one complex128 slot per chain (reference layout)")
    ap.add_argument("--no-layout-compare", action="store_true", help="skip the reference-layout side measurement")
    ap.add_argument("--no-live-traffic", action="store_true", help="roofline.traffic from profiles/pmc_summary.json instead of two rocprofv3 --pmc child passes")
    ap.add_argument("--no-config-legs", action="store_true", help="skip the side legs of BASELINE configs[1] (L=64) and configs[4] (L=512 weak lensing)")
    ap.add_argument("--noise-bits", type=int, default=64, choices=(32, 64),
                    help="Box-Muller arithmetic of the HEADLINE's Philox stream: 64 (default) = fp64 as the reference's randn "
                         "(pxmcmc/mcmc.py:193); 32 = the f32 transcendental units.  The other one is timed as the side leg")
    ap.add_argument("--no-noise-leg", "--no-f64-noise-leg", dest="no_noise_leg", action="store_true",
                    help="skip the second timing with the other Box-Muller precision (value_f32_noise)")
    ap.add_argument("--config5-cap", type=int, default=C5_TUNE_CAP,
                    help="iterations of the configs[4] PxMALA run (delta adapts from 1e-6; the timed stretch follows the first lap "
                         "with the window acceptance inside [0.3, 0.7])")
    ap.add_argument("--pmc-child", default="", help=argparse.SUPPRESS)  # internal: body of the side legs' rocprofv3 --pmc passes
    args = ap.parse_args()
    if args.pmc_child:
        pmc_child(args.pmc_child)
        return

    # --gpus N > 1 without a torchrun environment: this process has not imported torch nor touched the GPU -- it
    # starts the N ranks as a CHILD torchrun (never an exec), lets rank 0's JSON line through on the inherited
    # stdout and leaves with the child's exit code
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))

    # CPU legs and the PMC traffic of the dominant kernel, live: before anything here initialises the GPU (the
    # worker pool / the profiler children are started by a process that holds no GPU state yet); N = 1 only
    single = int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.gpus == 1
    cpu = cpu_legs(args) if single and not args.no_cpu_baseline else None
    live = (None, None, None)
    legs_pmc, legs_pmc_note = None, "not collected"
    if single and not args.no_live_traffic and not args.no_cpu_baseline:
        live = live_traffic()
        if not args.no_config_legs:
            # one pair of child passes PER LEG: launches of equal grid size from different problems never share a key
            legs_pmc, notes = {}, []
            for leg, which in (("configs[1]", "config2"), ("configs[4]", "config5")):
                legs_pmc[leg], info = pmc_passes(["--pmc-child", which], timeout_s=200)
                notes.append(f"{leg}: {info:.0f} s" if legs_pmc[leg] is not None else f"{leg}: child passes failed: {info}")
            legs_pmc_note = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of each side leg's launches in this run ("
                             + "; ".join(notes) + "), joined on the workgroup count within the leg")

    import torch
    import torch.distributed as dist

    from pxmcmc_amd import distributed as D

    rank, local_rank, world = D.env_rank_world()
    failures = []
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} started with WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP kernels are the only compute path")
    # rehearsal aid for a one-GPU box: PXM_BENCH_REHEARSE=1 puts every rank on cuda:0 and uses gloo for the
    # barrier / max-reduce (RCCL refuses two ranks on one device); never set by the driver
    rehearse = bool(os.environ.get("PXM_BENCH_REHEARSE"))
    dev_index = 0 if rehearse else local_rank
    torch.cuda.set_device(dev_index)
    if rehearse:
        D.init(backend="gloo")
    else:
        D.init(backend="nccl", device_id=torch.device("cuda", dev_index))

    from pxmcmc_amd import ops
    from pxmcmc_amd.forward import SphericalWaveletTransformOperator
    from pxmcmc_amd.mcmc import MYULA, PxMCMCParams
    from pxmcmc_amd.prior import S2_Wavelets_L1

    C = CHAINS_PER_GPU
    # ---- synthetic problem, identical on every rank (chains differ by their Philox key) ----
    sht = ops.ShtPlan(L, 0, max_chains=1)
    truth, rng = synthetic_field(lambda flm: sht.inverse(flm).cpu().numpy(), L, seed=2)
    del sht
    data = truth + SIGMA * rng.normal(size=truth.size)
    op = SphericalWaveletTransformOperator(data, SIGMA, "synthesis", L, B, J_MIN, max_chains=C)
    reg = S2_Wavelets_L1("synthesis", op.transform.inverse, op.transform.inverse_adjoint, LMDA * MU, L=L, B=B, J_min=J_MIN)
    delta, s_norm2 = stable_delta(op.transform, SIGMA, LMDA)
    params = PxMCMCParams(lmda=LMDA, delta=delta, mu=MU, nsamples=1, nburn=0, ngap=1, verbosity=0)
    first_chain, _ = D.shard_chains(world * C, rank, world)  # weak scaling: C chains per GPU
    sampler = MYULA(op, reg, params, nchains=C, rng="philox", seed=2, chain_offset=first_chain,
                    real_pairs=not args.no_real_pairs, noise_bits=args.noise_bits)
    sampler._prepare()
    assert sampler._fused_wav, "the fused wavelet path must be the one benchmarked"
    with contextlib.redirect_stdout(io.StringIO()):
        X, preds = sampler._initial_sample(np.zeros(op.nparams))

    # The timed region drives the sampler's own stepping engine (MYULA._engine_*): the fused iteration
    # replayed from a captured HIP graph (2 iterations per replay) when capture is available.
    if sampler._pairs_ok(X):  # real data + real state: two real chains per complex128 slot (MYULA real_pairs)
        sampler._pairs_start()
    eng = sampler._engine_start(X, preds, 0)
    barrier = D.barrier

    # Timed regions.  Each one: start barrier (device idle + ranks aligned), then each rank times ITS OWN K steps up to its
    # own device synchronise -- no torch.distributed call lies between t0 and dt (nothing crosses ranks on the data path, so
    # a collective inside a 3-ms clock would only measure the collective).  The stop-side barrier follows, outside the
    # clock; a region's time is the MAX over ranks of the per-rank times.
    def timed_region():
        barrier()
        t0 = time.perf_counter()
        sampler._engine_advance(args.steps)
        torch.cuda.synchronize()
        dt_r = time.perf_counter() - t0
        barrier()
        return dt_r

    # (i) what the command asked for, literally: W warm-up steps from a cold start, then K timed steps -> value_no_ramp.
    sampler._engine_advance(args.warmup)
    dt_no_ramp_rank = timed_region()
    # (ii) Clock ramp: the driver's default run times 20 steps = 3 ms, shorter than the time the GPU takes to reach its
    # sustained clocks from idle; --ramp further untimed iterations precede the regions the headline is taken from.
    # "warmup" in the JSON is the W of the command line; config.untimed_steps_before_headline is everything that ran
    # before the first headline region (W + K of region (i) + ramp).
    sampler._engine_advance(args.ramp)
    untimed = args.warmup + args.steps + args.ramp
    # (iii) the timed region --repeats times back to back; value = MEDIAN region, min / max beside it
    dt_ranks = [timed_region() for _ in range(max(args.repeats, 1))]
    dt_rank = dt_ranks[0]
    # roofline leg: the same steps once more through the same engine, launched eagerly (no graph replay) so that
    # every k_sht_gemm launch can be bracketed by HIP events on its stream (events cannot be read back from
    # inside a graph replay)
    n_prof = min(args.steps, 100)
    plan = eng["plan"]
    plan.profile_enable(3 * n_prof + 8)
    graphs = eng["graph"], eng["graph_long"]
    eng["graph"] = eng["graph_long"] = None
    sampler._engine_advance(n_prof)
    eng["graph"], eng["graph_long"] = graphs
    torch.cuda.synchronize()
    (gms, gnl, gnb, gnf), (dms_, dnl_, dnb_) = plan.profile_read()
    # launch classes of the ring GEMM (Gram / forward-adjoint group / forward group): the same steps once more
    plan.profile_enable(3 * n_prof + 8)
    eng["graph"] = eng["graph_long"] = None
    sampler._engine_advance(n_prof)
    eng["graph"], eng["graph_long"] = graphs
    torch.cuda.synchronize()
    l_ms, l_bytes, l_wgs = plan.profile_read_launches(3 * n_prof + 8)
    plan.profile_enable(0)
    gemm_classes = []  # one per (workgroups, algorithmic bytes): Gram, forward-adjoint group, forward group
    for wgs, nbytes in sorted(set(zip(l_wgs.tolist(), np.round(l_bytes).tolist()))):
        sel = (l_wgs == wgs) & (np.round(l_bytes) == nbytes)
        us = float(l_ms[sel].mean() * 1e3)
        gemm_classes.append({"workgroups": int(wgs), "alg_MB": nbytes / 1e6, "launches": int(sel.sum()), "avg_us": us,
                             "GBs": nbytes / us / 1e3, "frac": nbytes / us / 1e3 / HBM_PEAK_GBS})

    class _V:  # (keeps the field names of the report below)
        def __init__(self, v):
            self.value = v

    ms, nl, nb, nf, dms, dnl, dnb = _V(gms), _V(gnl), _V(gnb), _V(gnf), _V(dms_), _V(dnl_), _V(dnb_)
    X, preds = sampler._engine_state()
    if not os.environ.get("PXM_BENCH_ABLATION"):  # (timing-only ablation builds of the library compute garbage)
        assert bool(torch.isfinite(X.real).all()) and bool(torch.isfinite(preds.real).all())

    region_dts = [D.max_over_ranks(v) for v in dt_ranks]          # per region: max over ranks
    order = sorted(range(len(region_dts)), key=lambda k: region_dts[k])
    k_med = order[(len(order) - 1) // 2]                           # median region (lower median for an even count)
    dt = region_dts[k_med]
    per_rank_ms = [v * 1e3 / args.steps for v in D.all_gather_float(dt_ranks[k_med])]
    dt_no_ramp = D.max_over_ranks(dt_no_ramp_rank)
    # cost of one empty start/stop barrier of this process group (synchronise + dist.barrier + synchronise), after the
    # run: what the clock WOULD have contained had the stop barrier been inside it
    barrier()
    tb = time.perf_counter()
    for _ in range(5):
        barrier()
    barrier_us = D.max_over_ranks((time.perf_counter() - tb) / 5) * 1e6
    ranks_seen = D.count_ranks()  # all-reduced over the process group (RCCL): the record shows N ranks took part
    used_graph = eng["graph"] is not None

    # informative side figure (rank 0, outside the timed region): the same iteration with one complex128 slot
    # per chain, i.e. the reference's state layout without the real-pair packing
    ref_layout_rate = None
    if rank == 0 and world == 1 and eng["pairs"] and not args.no_layout_compare:
        sampler._engine_stop()
        s2 = MYULA(op, reg, params, nchains=C, rng="philox", seed=2, chain_offset=first_chain, real_pairs=False,
                   noise_bits=args.noise_bits)
        s2._prepare()
        with contextlib.redirect_stdout(io.StringIO()):
            X2, P2 = s2._initial_sample(np.zeros(op.nparams))
        s2._engine_start(X2, P2, 0)
        s2._engine_advance(untimed)  # the same untimed count as the main leg
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        s2._engine_advance(args.steps)
        torch.cuda.synchronize()
        ref_layout_rate = C * args.steps / (time.perf_counter() - t1)
        s2._engine_stop()

    # The same timed region with the OTHER Box-Muller precision (headline: fp64 log / sqrt / sincos, as the reference's
    # np.random.randn, pxmcmc/mcmc.py:193; side leg: the f32 transcendental units): same library, same run, same counts.
    other_bits = 96 - args.noise_bits
    noise_leg = None
    if rank == 0 and world == 1 and not args.no_noise_leg:
        s3 = MYULA(op, reg, params, nchains=C, rng="philox", seed=2, chain_offset=first_chain,
                   real_pairs=not args.no_real_pairs, noise_bits=other_bits)
        s3._prepare()
        with contextlib.redirect_stdout(io.StringIO()):
            X3, P3 = s3._initial_sample(np.zeros(op.nparams))
        if s3._pairs_ok(X3):
            s3._pairs_start()
        e3 = s3._engine_start(X3, P3, 0)
        s3._engine_advance(untimed)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        s3._engine_advance(args.steps)
        torch.cuda.synchronize()
        dt3 = time.perf_counter() - t1
        p3 = e3["plan"]
        p3.profile_enable(3 * n_prof + 8)
        g3 = e3["graph"], e3["graph_long"]
        e3["graph"] = e3["graph_long"] = None
        s3._engine_advance(n_prof)
        e3["graph"], e3["graph_long"] = g3
        torch.cuda.synchronize()
        _, (d3ms, d3n, _) = p3.profile_read()
        p3.profile_enable(0)
        X3, _ = s3._engine_state()
        assert bool(torch.isfinite(X3.real).all())
        s3._engine_stop()
        noise_leg = {"noise_bits": other_bits, "value": C * args.steps / dt3, "ms_per_step": dt3 / args.steps * 1e3,
                     "dft_kernel_avg_launch_us": d3ms * 1e3 / max(d3n, 1), "noise": NOISE_NOTES[other_bits],
                     "what": f"the timed region repeated with MYULA(noise_bits={other_bits}): same library, same run, same "
                             "warm-up and step counts"}
        del s3, e3, p3

    config_legs = None
    if rank == 0 and world == 1 and not args.no_config_legs:
        sampler._engine_stop()
        config_legs = {"pmc_source": legs_pmc_note}
        for name, fn in (("configs[1]", config2_leg), ("configs[4]", config5_leg)):
            t1 = time.perf_counter()
            try:
                kw = {"cap": args.config5_cap} if fn is config5_leg else {}
                config_legs[name] = fn(pmc=(legs_pmc or {}).get(name), **kw)
            except Exception as exc:  # a side leg must never take the headline down with it
                config_legs[name] = {"error": repr(exc)}
            config_legs[name]["leg_wall_s"] = time.perf_counter() - t1
            ops.tables_trim()

    if world > 1 and not args.no_config_legs:
        # BASELINE configs[4] "8 chains on 8 GPUs": every rank runs its own PxMALA chain after the headline
        sampler._engine_stop()
        t1 = time.perf_counter()
        leg = config5_multirank_leg(rank, world, D, cap=args.config5_cap)
        leg["leg_wall_s"] = time.perf_counter() - t1
        config_legs = {"configs[4]": leg}

    if rank == 0:
        value = world * C * args.steps / dt
        gemm_avg_us = ms.value * 1e3 / max(nl.value, 1)
        achieved = nb.value / (ms.value * 1e-3) / 1e9 if ms.value > 0 else 0.0
        traffic, traffic_src, traffic_classes = live
        pmc = os.path.join(ROOT, "profiles", "pmc_summary.json")
        if traffic is None and os.path.exists(pmc):  # no live passes (flag, N > 1, or they failed): the committed figure
            live_note = f"; live passes: {live[1]}" if live[1] else ""
            with open(pmc) as fh:
                traffic = json.load(fh).get("k_sht_gemm_hbm_bytes_per_launch")
            traffic_src = "profiles/pmc_summary.json (static: rocprofv3 --pmc passes of this command, not measured in this run)" + live_note
        # measured HBM rate: PMC bytes of the per-step launch classes (child passes) over the live event time of the
        # same classes in this process, joined class by class in launch order of size (Gram < groups)
        measured = None
        by_wgs = {c["workgroups"]: c for c in traffic_classes or []}
        if gemm_classes and all(g["workgroups"] in by_wgs for g in gemm_classes):
            joined = [dict(by_wgs[g["workgroups"]], alg_MB=g["alg_MB"], avg_us=g["avg_us"],
                           hbm_GBs=by_wgs[g["workgroups"]]["hbm_MB"] / g["avg_us"] * 1e3) for g in gemm_classes]
            tb = sum(c["hbm_MB"] * 1e6 * g["launches"] for c, g in zip(joined, gemm_classes))
            tt = sum(g["avg_us"] * 1e-6 * g["launches"] for g in gemm_classes)
            # roofline.traffic: HBM bytes per launch of the timed iteration's launches only (the child passes also see
            # the set-up transforms of the power iteration: other launch classes, dropped by the join)
            traffic = tb / sum(g["launches"] for g in gemm_classes)
            traffic_src += "; averaged over the launch classes of the timed iteration (joined on the workgroup count)"
            measured = {"hbm_GBs": tb / tt / 1e9, "hbm_frac": tb / tt / 1e9 / HBM_PEAK_GBS,
                        "what": "PMC bytes (FETCH_SIZE x2 + WRITE_SIZE, child passes) of each per-step launch class, joined "
                                "on the workgroup count, / the live event time of the same class in this process",
                        "classes": joined}
        out = {
            "metric": "MYULA samples/sec at L=256 synthesis",
            "value": value,
            "unit": "samples/s",
            "n_gpus": world,
            "ranks_seen": ranks_seen,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            # the command as written -- W warm-up steps from a cold start, K timed steps, no clock ramp:
            "value_no_ramp": world * C * args.steps / dt_no_ramp,
            "ms_per_step_no_ramp": dt_no_ramp / args.steps * 1e3,
            # the headline region repeated back to back; `value` / `ms_per_step` are the median region's
            "repeats": {"n": len(region_dts), "ms_per_step": [v / args.steps * 1e3 for v in region_dts],
                        "value_min": world * C * args.steps / max(region_dts), "value_max": world * C * args.steps / min(region_dts),
                        "value_median": world * C * args.steps / dt,
                        "what": "the K-step timed region (barrier, K steps, own device synchronise; max over ranks) run "
                                "n times back to back after the clock ramp; value = the median region"},
            "per_rank_ms_per_step": per_rank_ms,
            "barrier_us": barrier_us,
            "timing": "per rank: start barrier, t0, K steps, own device synchronise, stop clock (no collective inside); "
                      "ms_per_step = max over ranks; barrier_us = one empty barrier of the group, measured after the run",
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "MYULA, spherical-wavelet synthesis L=256 B=2 J_min=2 (N=305060, P=130816), identity measurement, "
                            "S2_Wavelets_L1 prox, real synthetic data, 16 chains batched per GPU; state layout: "
                            + ("two real chains per complex128 slot (real-signal symmetry, SURVEY 8d)" if eng["pairs"]
                               else "one complex128 slot per chain (reference layout)"),
                "real_pairs": bool(eng["pairs"]),
                "samples_per_s_one_gpu_reference_layout": ref_layout_rate,
                "sigma": SIGMA, "lmda": LMDA, "delta": delta, "synthesis_norm2": s_norm2,
                "chains_per_gpu": C,
                "global_chains": world * C,
                "parallelism": f"chains sharded over {world} GPU(s), no collective on the data path",
                "hip_graph": used_graph,
                "graph_iterations_per_replay": 2 * sampler._GRAPH_PAIRS if used_graph else 0,
                "warmup_requested": args.warmup,
                "clock_ramp_steps": args.ramp,
                "untimed_steps_before_headline": untimed,
                "noise_bits": args.noise_bits,
                "noise": NOISE_NOTES[args.noise_bits],
            },
            # both noise precisions by name: the headline `value` is the one of config.noise_bits (64 unless --noise-bits 32),
            # the other is the side leg (null when skipped)
            "value_f64_noise": value if args.noise_bits == 64 else (noise_leg["value"] if noise_leg else None),
            "value_f32_noise": value if args.noise_bits == 32 else (noise_leg["value"] if noise_leg else None),
            "noise_leg": noise_leg,
            # BASELINE configs[1] and configs[4] on this GPU, outside the headline's timed region
            "configs": config_legs,
            "roofline": {
                "bound": "hbm",
                "kernel": "k_sht_gemm (SHT ring-table GEMM, v_mfma_f64_16x16x4_f64)",
                "achieved": achieved,
                "achieved_is": "algorithmic bytes (DESIGN.md section 6) / live kernel time -- an effective rate; the HBM rate "
                               "of the bytes the counters saw is roofline.measured",
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "measured": measured,
                "avg_launch_us": gemm_avg_us,
                "launches": int(nl.value),
                "launch_classes": gemm_classes,
                "alg_bytes_per_launch": nb.value / max(nl.value, 1),
                # the same launches against the matrix pipe (v_mfma_f64_16x16x4_f64: 78.6 TFLOP/s dense spec,
                # 47 TFLOP/s sustained by an MFMA-only loop on this part): the kernel is co-limited
                "mfma_tflops": nf.value / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0,
                "mfma_frac_of_spec": nf.value / (ms.value * 1e-3) / 78.6e12 if ms.value > 0 else 0.0,
                # the rate an MFMA-ONLY loop sustains on this part (scripts/probes/mfma_rate.hip: 8 independent accumulators, 2 waves
                # per SIMD, 44.6 ns per v_mfma_f64_16x16x4_f64 and SIMD -- the part throttles under fp64 matrix load): the
                # two grouped launches (1.42 GFLOP each) run at 0.90-0.95 of it, i.e. they are bound by the matrix pipe
                "mfma_sustained_tflops": MFMA_SUSTAINED_TFLOPS,
                "mfma_frac_of_sustained": nf.value / (ms.value * 1e-3) / (MFMA_SUSTAINED_TFLOPS * 1e12) if ms.value > 0 else 0.0,
            },
        }
        if dnl.value > 0 and dms.value > 0:
            # second kernel of the iteration, the larger share of its time: the grouped phi-DFT + prox + update + Philox
            # kernel is bound by fp64 VALU issue (Bluestein butterflies), not by HBM -- reported for completeness
            out["dft_kernel"] = {
                "kernel": "k_ring2px_group5 (rings -> X' -> rings of every wavelet scale, one grid; the two 511-point scales by the "
                          "exact-length unit 511 = 7 x 73 of csrc/dft_pfa.h, the smaller ones by Bluestein)",
                "bound": "fp64 VALU issue (transforms + Philox / Box-Muller) and LDS transposes, 4 waves per SIMD: work per CU, no tail round",
                "avg_launch_us": dms.value * 1e3 / dnl.value,
                "launches": int(dnl.value),
                "alg_bytes_per_launch": dnb.value / dnl.value,
                "achieved_GBs": dnb.value / (dms.value * 1e-3) / 1e9,
                "hbm_frac": dnb.value / (dms.value * 1e-3) / 1e9 / HBM_PEAK_GBS,
            }
        if cpu is not None:  # the CPU legs ran at N = 1 only, before the GPU was touched
            T = reg.T
            assert abs(s_norm2 / NOMINAL_S_NORM2 - 1.0) < 0.05, "the CPU legs' nominal ||S||^2 is off"
            # free full-size parity evidence on this box: the benchmarked step vs the oracle on the same noise
            if eng["pairs"]:
                out["parity"] = {"what": "3 iterations of the benchmarked step (ring-space + Gram + real pairs, injected noise) "
                                         "vs the oracle's literal loop, chains 0 and 9, error relative to max |X|",
                                 "tolerance": PARITY_TOL}
                try:
                    out["parity"]["max_rel_err_X"] = parity_leg(plan, data, reg.T_dev, T, delta)
                except Exception as exc:  # reported in the line AND in the exit code (leg_failures)
                    out["parity"]["error"] = repr(exc)
            out["cpu_baseline"] = cpu
        failures = leg_failures(out)
        out["legs_ok"] = not failures
        out["leg_failures"] = failures
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if rank == 0 and failures:  # the headline line is out; a broken side leg or parity leg still turns the run red
        print("bench.py: " + "; ".join(failures), file=sys.stderr)
        raise SystemExit(3)


if __name__ == "__main__":
    main()
