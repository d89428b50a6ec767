"""world_size-2 gloo run of the multi-process host path (chains shard, no data-path collective)."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

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
