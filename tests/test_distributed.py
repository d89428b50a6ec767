"""world_size-2 gloo run of the multi-process host path (chains shard, no data-path collective)."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import ROOT

WORKER = textwrap.dedent(
    """
    import os, sys, time
    import numpy as np
    sys.path.insert(0, os.environ["PXM_ROOT"])
    from pxmcmc_amd import distributed as D
    from oracle import philox

    rank, local_rank, world = D.init(backend="gloo")
    assert world == 2
    TOTAL, N, SEED, IT = 5, 257, 11, 3            # 5 chains over 2 ranks: 3 + 2
    first, count = D.shard_chains(TOTAL, rank, world)
    # every rank draws only its own chains' noise: the stream is keyed by the GLOBAL chain id
    mine = np.stack([philox.randn_real(N, SEED, first + c, IT) for c in range(count)])
    D.barrier()
    t = D.max_over_ranks(1.0 + rank)               # the slowest rank defines the elapsed time
    per_rank = D.all_gather_float(0.5 + rank)      # bench.py's per-rank times, in rank order
    assert per_rank == [0.5, 1.5] and D.count_ranks() == 2
    allc = D.gather_summaries(mine).numpy()
    if rank == 0:
        ref = np.stack([philox.randn_real(N, SEED, c, IT) for c in range(TOTAL)])
        assert allc.shape == ref.shape and np.array_equal(allc, ref), "sharded stream differs from the single-process stream"
        assert t == 2.0
        print("OK", first, count, flush=True)
    D.barrier()
    """
)


def test_shard_chains_arithmetic():
    from pxmcmc_amd.distributed import shard_chains

    for total in (1, 5, 16, 128, 129):
        for world in (1, 2, 3, 8):
            spans = [shard_chains(total, r, world) for r in range(world)]
            assert sum(c for _, c in spans) == total
            pos = 0
            for first, count in spans:
                assert first == pos
                pos += count
    assert shard_chains(128, 3, 8) == (48, 16)


def test_two_process_gloo_run(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PXM_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "OK 0 3" in res.stdout


HIP_WORKER = textwrap.dedent(
    """
    import contextlib, io, os, sys
    import numpy as np
    sys.path.insert(0, os.environ["PXM_ROOT"])
    import torch
    from pxmcmc_amd import distributed as D
    from pxmcmc_amd.forward import SphericalWaveletTransformOperator
    from pxmcmc_amd.mcmc import MYULA, PxMCMCParams
    from pxmcmc_amd.prior import S2_Wavelets_L1

    rank, local_rank, world = D.init(backend="gloo")   # both ranks share the one GPU of the test box (RCCL refuses that)
    torch.cuda.set_device(0)
    L, B, J_min, TOTAL = 16, 2, 2, 6
    data = np.random.default_rng(3).normal(size=L * (2 * L - 1))     # same problem on every rank
    first, count = D.shard_chains(TOTAL, rank, world)
    op = SphericalWaveletTransformOperator(data, 0.2, "synthesis", L, B, J_min, max_chains=count)
    reg = S2_Wavelets_L1("synthesis", None, None, 1e-3, L=L, B=B, J_min=J_min)
    p = PxMCMCParams(lmda=1e-3, delta=5e-4, nsamples=3, nburn=4, ngap=2, verbosity=0)
    s = MYULA(op, reg, p, nchains=count, seed=21, chain_offset=first)   # Philox keyed by the GLOBAL chain id
    with contextlib.redirect_stdout(io.StringIO()):
        s.run(start_point=np.zeros(op.nparams))
    assert s.used_graph
    mine = s.chain if count > 1 else s.chain[None]
    D.barrier()
    allc = D.gather_summaries(mine).numpy()
    t = D.max_over_ranks(1.0 + rank)
    if rank == 0:
        op1 = SphericalWaveletTransformOperator(data, 0.2, "synthesis", L, B, J_min, max_chains=TOTAL)
        one = MYULA(op1, reg, p, nchains=TOTAL, seed=21)
        with contextlib.redirect_stdout(io.StringIO()):
            one.run(start_point=np.zeros(op1.nparams))
        assert allc.shape == one.chain.shape, (allc.shape, one.chain.shape)
        err = np.abs(allc - one.chain).max() / np.abs(one.chain).max()
        assert err < 1e-12, err      # sharded over 2 processes == one batch of 6 chains
        assert t == 2.0
        print("HIP-OK", first, count, flush=True)
    D.barrier()
    """
)


@pytest.mark.gpu
def test_two_rank_hip_chain_sharding(tmp_path):
    """SURVEY.md row (e) on the HIP path: two processes (torchrun, gloo rendezvous, both on the box's one GPU) each
    run the fused HIP MYULA engine on their shard of 6 chains; gathered, the chains equal a single-process batch of
    6 -- the Philox stream is keyed by the global chain id, nothing else crosses the process boundary."""
    script = tmp_path / "hip_worker.py"
    script.write_text(HIP_WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PXM_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    assert "HIP-OK 0 3" in res.stdout


def test_bench_self_launch_command_is_a_child_torchrun(monkeypatch):
    """`python bench.py --gpus N` without a torchrun environment starts its N ranks as a CHILD
    `python -m torch.distributed.run` (no exec, before torch is imported) and returns the child's exit code."""
    import importlib.util
    import subprocess

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    class Done:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return Done()

    monkeypatch.setattr(subprocess, "run", fake_run)
    assert bench.self_launch(4, ["--gpus", "4", "--steps", "20"]) == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-5:] == [os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "20"]
    assert seen["env"]["MASTER_ADDR"] == "127.0.0.1" and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # and main() takes that branch before importing torch when WORLD_SIZE is unset
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3"])
    monkeypatch.setattr(bench, "self_launch", lambda n, argv: 5 if (n, argv) == (2, ["--gpus", "2", "--steps", "3"]) else 99)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 5


RANKS8_WORKER = textwrap.dedent(
    """
    import json, os, sys
    sys.path.insert(0, os.environ["PXM_ROOT"])
    from pxmcmc_amd import distributed as D

    rank, local_rank, world = D.init(backend="gloo")
    assert world == 8 and local_rank == rank
    first, count = D.shard_chains(world * 16, rank, world)   # bench.py: 16 chains per GPU, weak scaling
    assert (first, count) == (16 * rank, 16)
    D.barrier()
    dt = D.max_over_ranks(0.001 * (1 + rank))
    per_rank = D.all_gather_float(0.001 * (1 + rank))
    seen = D.count_ranks()
    D.barrier()
    # BASELINE configs[4] on N ranks: bench.config5_multirank_leg itself (its all-gathers, in its order) with the GPU work of a
    # rank replaced by a stand-in -- every rank "runs" its own chain (chain id = rank) and reports its own figures
    import importlib.util
    import torch
    spec = importlib.util.spec_from_file_location("bench_w", os.path.join(os.environ["PXM_ROOT"], "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)

    class _Op:
        def _wl_plan(self):
            return object()

    bench.config5_problem = lambda: (_Op(), None, None, None, 5e-7)
    bench.config5_tuned = lambda op, reg, tr, lmda, chain_offset=0, cap=0: {
        "ms_per_iteration": 0.5 + 0.01 * chain_offset, "acceptance_in_timed_stretch": 0.4 + 0.01 * chain_offset,
        "delta_at_start_of_stretch": 1e-10 * (1 + chain_offset), "iterations_before_timed_stretch": 700 + 50 * chain_offset,
        "tuned": chain_offset != 3, "finite": True}
    torch.cuda.synchronize = lambda: None
    leg = bench.config5_multirank_leg(rank, world, D, cap=1234)
    D.barrier()
    if rank == 0:
        print(json.dumps({"ranks_seen": seen, "max": dt, "per_rank": per_rank, "config5": leg}), flush=True)
    """
)


def test_eight_rank_rendezvous_through_bench_self_launch(tmp_path, capfd):
    """The driver's N = 8 launch shape on CPU: `bench.self_launch(8, ...)` -- the same child `torch.distributed.run`
    command, free-port choice and environment `python bench.py --gpus 8` uses -- starts eight gloo ranks that run every
    torch.distributed call of the benchmark (init from the torchrun environment, chain sharding, barrier, MAX
    all-reduce, the all-gathered per-rank times, the rank count).  (Eight ranks on the ONE test GPU are not started:
    the GPU box allows six processes on its card.)"""
    import importlib.util
    import json

    spec = importlib.util.spec_from_file_location("bench_mod8", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    script = tmp_path / "ranks8_worker.py"
    script.write_text(RANKS8_WORKER)
    old = {k: os.environ.get(k) for k in ("PXM_ROOT", "RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    os.environ["PXM_ROOT"] = ROOT
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        os.environ.pop(k, None)
    try:
        rc = bench.self_launch(8, [], script=str(script))
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    outerr = capfd.readouterr()
    assert rc == 0, outerr.out[-2000:] + outerr.err[-3000:]
    line = [ln for ln in outerr.out.splitlines() if ln.startswith("{")]
    assert len(line) == 1
    out = json.loads(line[0])
    assert out["ranks_seen"] == 8 and abs(out["max"] - 0.008) < 1e-12
    assert np.allclose(out["per_rank"], 0.001 * np.arange(1, 9))
    c5 = out["config5"]  # one PxMALA chain per rank ("8 chains on 8 GPUs"): per-rank figures in rank order, max / sum rules
    assert c5["ranks_seen"] == 8 and "error" not in c5 and c5["finite"] is True and c5["iterations_run"] == 1234
    assert np.allclose(c5["per_rank_ms_per_iteration"], 0.5 + 0.01 * np.arange(8)) and abs(c5["ms_per_iteration"] - 0.57) < 1e-12
    assert abs(c5["samples_per_s"] - 8e3 / 0.57) < 1e-6 and abs(c5["samples_per_s_sum_of_ranks"] - sum(1e3 / (0.5 + 0.01 * k) for k in range(8))) < 1e-6
    assert np.allclose(c5["per_rank_acceptance"], 0.4 + 0.01 * np.arange(8)) and np.allclose(c5["per_rank_delta"], 1e-10 * np.arange(1, 9))
    assert c5["per_rank_iterations_before_timed_stretch"] == [700 + 50 * k for k in range(8)]
    assert c5["per_rank_tuned"] == [k != 3 for k in range(8)]


def test_bench_leg_failures_turn_the_exit_code_red():
    """bench.leg_failures: a side leg that raised, a non-finite leg or a parity error beyond the tolerance is listed
    (=> `legs_ok: false`, exit code 3 after the JSON line); a clean record is not."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_modf", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    ok = {"configs": {"pmc_source": "x", "configs[1]": {"finite": True}, "configs[4]": {"finite": True}},
          "parity": {"max_rel_err_X": 3e-15}}
    assert bench.leg_failures(ok) == [] and bench.leg_failures({"configs": None}) == []
    bad = {"configs": {"pmc_source": "x", "configs[1]": {"error": "RuntimeError('boom')"}, "configs[4]": {"finite": False}},
           "parity": {"max_rel_err_X": 2e-9}}
    msgs = bench.leg_failures(bad)
    assert len(msgs) == 3 and "boom" in msgs[0] and "configs[4]" in msgs[1] and "parity" in msgs[2]
    assert len(bench.leg_failures({"parity": {"error": "x"}})) == 1
    assert len(bench.leg_failures({"parity": {"max_rel_err_X": float("nan")}})) == 1


def test_bench_timed_region_holds_no_collective():
    """bench.py's clock: `t0` right after the start barrier, `dt_rank` right after the rank's own device synchronise --
    no torch.distributed call (barrier, all-reduce, gather) between the two (the chains never cross ranks,
    experiments/earthtopography/main.py:31-36,169: one process per chain)."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    i0 = src.index("t0 = time.perf_counter()\n        sampler._engine_advance(args.steps)")
    i1 = src.index("dt_r = time.perf_counter() - t0")
    region = [ln.split("#")[0] for ln in src[i0:i1].splitlines()]
    code = "\n".join(region)
    banned_calls = ("barrier(", "dist.", "D.", "all_reduce", "max_over_ranks", "all_gather")
    for banned in banned_calls:
        assert banned not in code, (banned, code)
    assert "torch.cuda.synchronize()" in code
    before = src[:i0].rstrip().splitlines()[-1].strip()
    assert before == "barrier()", before  # the start barrier is the last statement before the clock starts
    # every timed region of the headline (the literal W + K run and the repeats) goes through that one function
    assert src.count("sampler._engine_advance(args.steps)") == 1 and src.count("timed_region()") >= 3
    # BASELINE configs[4] on N ranks: the body a rank times (one PxMALA run with lap stamps) holds no collective either;
    # the all-gathers of config5_multirank_leg come after it has returned
    j0, j1 = src.index("def config5_tuned("), src.index("def config5_leg(")
    body = "\n".join(ln.split("#")[0] for ln in src[j0:j1].splitlines())
    for banned in banned_calls:
        assert banned not in body, banned
    k0, k1 = src.index("def config5_multirank_leg("), src.index("PARITY_TOL =")
    leg = src[k0:k1]
    assert leg.index("config5_tuned(") < leg.index("D.all_gather_float(")


@pytest.mark.gpu
@pytest.mark.parametrize("nranks", [2, 4])
def test_bench_gpus2_self_launch_rehearsal(nranks):
    """The driver's N > 1 command issued plainly: `python bench.py --gpus N` (no torchrun in front).  On the test
    box's single GPU the ranks share cuda:0 and use gloo (PXM_BENCH_REHEARSE=1; RCCL refuses two ranks on one
    device); the JSON line must report every rank."""
    import json

    env = dict(os.environ, PXM_BENCH_REHEARSE="1", OMP_NUM_THREADS="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(nranks), "--steps", "6", "--warmup", "2", "--ramp", "10",
           "--repeats", "3", "--config5-cap", "400"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-3000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == nranks and out["ranks_seen"] == nranks and out["steps"] == 6 and out["warmup"] == 2
    assert out["config"]["untimed_steps_before_headline"] == 2 + 6 + 10 and out["config"]["clock_ramp_steps"] == 10
    # the headline qualifies itself: the literal W + K run without the ramp, and the timed region repeated (value = median)
    rep = out["repeats"]
    assert out["value_no_ramp"] > 0 and rep["n"] == 3 and len(rep["ms_per_step"]) == 3
    assert rep["value_min"] <= out["value"] <= rep["value_max"] and sorted(rep["ms_per_step"])[1] == out["ms_per_step"]
    assert out["config"]["global_chains"] == 16 * nranks and out["value"] > 0 and out["scaling"] == "weak"
    assert "cpu_baseline" not in out  # the CPU legs run at N = 1 only
    # the clock of a rank stops at its own device synchronise (no collective inside): per-rank times are reported, the
    # headline is their maximum, and the cost of one barrier of the group is measured separately
    per_rank = out["per_rank_ms_per_step"]
    assert len(per_rank) == nranks and all(t > 0 for t in per_rank)
    assert abs(max(per_rank) - out["ms_per_step"]) <= 1e-9 * out["ms_per_step"]
    assert out["barrier_us"] > 0
    assert abs(out["value"] - 16 * nranks * 1e3 / out["ms_per_step"]) <= 1e-6 * out["value"]
    assert out["noise_leg"] is None and out["value_f32_noise"] is None  # these side legs: N = 1 only
    # BASELINE configs[4] ("8 chains on 8 GPUs"): every rank ran its own PxMALA chain at L = 512 after the headline
    assert list(out["configs"]) == ["configs[4]"]
    c5 = out["configs"]["configs[4]"]
    assert "error" not in c5 and c5["ranks_seen"] == nranks and c5["finite"] is True
    assert len(c5["per_rank_ms_per_iteration"]) == nranks and all(t > 0 for t in c5["per_rank_ms_per_iteration"])
    assert c5["ms_per_iteration"] == max(c5["per_rank_ms_per_iteration"])
    assert abs(c5["samples_per_s"] - nranks * 1e3 / c5["ms_per_iteration"]) <= 1e-9 * c5["samples_per_s"]
    assert len(c5["per_rank_acceptance"]) == nranks and len(c5["per_rank_delta"]) == nranks
    assert c5["iterations_run"] == 400 and c5["per_rank_tuned"] == [False] * nranks  # (400 iterations: delta_0 still rejected)
    assert out["config"]["noise_bits"] == 64 and out["value_f64_noise"] == out["value"]  # headline = the fp64 noise stream
    assert out["legs_ok"] is True and out["leg_failures"] == []


RCCL_WORKER = textwrap.dedent(
    """
    import os, sys
    import numpy as np
    sys.path.insert(0, os.environ["PXM_ROOT"])
    import torch
    import torch.distributed as dist
    from pxmcmc_amd import distributed as D

    torch.cuda.set_device(0)
    # world size 1 (the test box has one GPU), but the REAL backend: RCCL initialises its communicator on cuda:0 and every
    # collective of the benchmark's multi-GPU path (barrier, MAX / SUM all-reduce, all-gather) runs through it
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    D.barrier()
    assert D.max_over_ranks(1.5) == 1.5
    assert D.all_gather_float(2.5) == [2.5]
    assert D.count_ranks() == 1
    g = D.gather_summaries(np.arange(12.0).reshape(3, 4)).numpy()
    assert g.shape == (3, 4) and np.array_equal(g, np.arange(12.0).reshape(3, 4))
    D.barrier()
    dist.destroy_process_group()
    print("RCCL-OK", flush=True)
    """
)


@pytest.mark.gpu
def test_rccl_backend_single_rank_collectives(tmp_path):
    """`pxmcmc_amd.distributed` on the backend the 8-GPU benchmark uses (`nccl` = RCCL), world size 1: communicator
    initialisation with a device id, barrier, the max-over-ranks and rank-count all-reduces on device tensors and the
    end-of-run all-gather.  (More than one rank per GPU is refused by RCCL; the two-rank tests above use gloo.)"""
    script = tmp_path / "rccl_worker.py"
    script.write_text(RCCL_WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PXM_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    assert "RCCL-OK" in res.stdout
