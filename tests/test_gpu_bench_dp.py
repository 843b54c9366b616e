"""bench.py's N > 1 path rehearsed on ONE GPU: two ranks launched the way the driver launches them
(`python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2`), both on GPU 0 with gloo standing in for
RCCL (SCA_BENCH_REHEARSAL=1; two RCCL ranks cannot share a device).  Checks that the run terminates, that rank 0 prints
exactly one JSON line with the contract's keys, whole-job throughput over both ranks and finite losses - for the
per-rank MMD (the default) and for the MMD over the global minibatch (all-gather of z)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("dp_mmd", ["local", "exact"])
def test_bench_two_ranks_rehearsal(dp_mmd):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, SCA_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4",
           "--warmup", "2", "--batch", "1024", "--cells", "40000", "--no-cpu-baseline", "--no-sweep", "--dp-mmd", dp_mmd]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 2048 and d["scaling"] == "weak"
    assert abs(d["value"] - 2048 * 4 / (d["ms_per_step"] * 4e-3)) <= 1e-6 * d["value"]
    assert all(v == v and abs(v) < 1e9 for v in d["final_losses"].values()), d["final_losses"]
    assert d["config"]["mmd"] == ("per-rank slice" if dp_mmd == "local" else "global minibatch (all-gather of z)")
