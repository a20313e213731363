"""The N > 1 launch path of bench.py on the one-GPU box (runs FIRST in the -m gpu session: the child processes are started before
this process has touched the GPU): `bench.py --gpus 2` spawns two ranks through torch.distributed.run exactly as on a multi-GPU
node; here both ranks share the one device (--share-gpu) and the timing fence runs over gloo instead of RCCL -- frame sharding,
fences, MAX / SUM reductions and the rank-0 JSON are the real ones.  `backend="nccl"` stays the only line that has not run."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_two_ranks_spawned_by_the_launcher():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--share-gpu", "--steps", "5",
           "--warmup", "2", "--repeats", "2", "--no-cpu-baseline", "--no-e2e", "--timing-iters", "5"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]            # rank 0 prints ONE JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["warmup"] == 2
    assert d["config"]["frames_per_step"] == 2 and d["config"]["frames_per_step_per_gpu"] == 1
    assert d["scaling"] == "weak" and d["value"] > 0
    assert d["parity"]["mismatches"] == 0 and d["parity"]["max_abs_diff_int8"] == 0
    assert d["cpu_baseline"] is None and d["e2e"] is None      # rank-0-at-N=1-only legs
