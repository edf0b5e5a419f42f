"""-m gpu: TWO real ranks before the first multi-GPU run.  This pool gives one GPU per call and RCCL will not put two
ranks on one device, so the N > 1 path had only ever run as a world of one, as partitions inside one process, and as gloo
ranks tracing with the oracle.  Here `bench.py --gpus 2` runs as the driver starts it -- torch.distributed.run, two
processes, each with its own contexts, partitions (row tiles t with t % 2 == rank, raythread.cpp:574-588 cut finer),
frames in flight and real kernels on the one device -- with the tiles handed over through pinned host memory and a gloo
gather (--collective host --same-device: the rehearsal mode; on a node with two GPUs the same command without those two
flags issues the RCCL gather).  The assembled frame must be the one-GPU frame bit for bit, the ranks' rays must add up."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import cobbletrace_amd as ca

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("ranks", [2, 3])
def test_real_ranks_on_one_device(ranks, scenes, gpu_ctx, tmp_path):
    dump = str(tmp_path / "frame.npy")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--collective", "host",
           "--same-device", "--scene", "bunny", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--dump-frame", dump]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    proc = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [l for l in proc.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, proc.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == ranks and rec["scaling"] == "strong" and "REHEARSAL" in rec["config"]["collective"]
    detail = rec["scaling_detail"]["ranks"]
    assert [r["rank"] for r in detail] == list(range(ranks))
    # one GPU, one process: the same frame
    scene, ctx = scenes("bunny"), gpu_ctx("bunny")
    frame = scene.frame(1920, 1080, rng_mode=ca.RNG_HASH, rng_seed=0)
    want, st = ctx.render(frame)
    got = np.load(dump)
    assert got.shape == want.shape and (got == want).all(), int((got != want).sum())
    assert sum(r["rays"] for r in detail) == st.rays == rec["config"]["rays_per_frame"]
    assert all(r["rays"] > st.rays // (2 * ranks) for r in detail), "every rank traced its share"
    assert all(r["trace_ms"] > 0 for r in detail)
    print("%d ranks on one device: %.2f ms per frame, rays per rank %s" % (ranks, rec["ms_per_step"], [r["rays"] for r in detail]))
