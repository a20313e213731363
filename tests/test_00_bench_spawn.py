"""The N > 1 launch path of bench.py on the one-GPU box (runs FIRST in the -m gpu session: the child processes are started before
this process has touched the GPU): `bench.py --gpus 2` spawns two ranks through torch.distributed.run exactly as on a multi-GPU
node; here both ranks share the one device (--share-gpu) and the timing fence runs over gloo instead of RCCL -- frame sharding,
fences, MAX / SUM reductions and the rank-0 JSON are the real ones.  `backend="nccl"` itself runs as a ONE-rank group (RCCL refuses two
ranks on one device): communicator creation, barrier and the MAX / SUM all-reduces on the device; what stays unrun is xGMI between GPUs."""
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
    assert d["fence"] == "gloo" and d["host"]["ranks_on_node"] == 2


@pytest.mark.gpu
def test_rccl_fence_on_a_one_rank_group():
    """sesrq.dist.Group(backend="nccl") as bench.py builds it (device_id given), forced at world size 1 in a child process: RCCL
    creates its communicator on cuda:0, the barrier and both reductions of run_timed() execute on the device, the group closes."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    code = ("import sys, torch; sys.path.insert(0, %r); from sesrq.dist import Group, run_timed\n"
            "dev = torch.device('cuda:0'); torch.cuda.set_device(dev)\n"
            "g = Group(backend='nccl', device=dev, timeout_s=120, force=True)\n"
            "x = torch.zeros(1 << 20, device=dev)\n"
            "r = run_timed(g, lambda: x.add_(1), steps=4, warmup=1, repeats=2, sync=torch.cuda.synchronize, units_per_step=3)\n"
            "assert r['units_per_step_total'] == 3 and len(r['elapsed']) == 2 and g.max_over_ranks(1.5) == 1.5\n"
            "assert float(x[0]) == 1 + 16 + 8\n"
            "import torch.distributed as d; print('backend', d.get_backend(g.pg), 'fence', g.fence, flush=True); g.close()") % os.path.join(ROOT, "sesr-pytorch-quantize_amd")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "backend nccl fence nccl" in r.stdout


@pytest.mark.gpu
def test_bench_two_ranks_keep_a_fence_when_rccl_cannot_start():
    """Round 5: the DEFAULT backend (nccl) with two ranks on the one device -- RCCL refuses two ranks on one GPU, i.e. a real failure of the
    real library on the real box: the run must still end with rc 0, ONE JSON line, parity 0 and `fence` = "gloo (nccl: ...)"."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--steps", "5",
           "--warmup", "2", "--repeats", "2", "--no-cpu-baseline", "--no-e2e", "--timing-iters", "5"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["parity"]["mismatches"] == 0 and d["value"] > 0
    assert d["fence"].startswith("gloo (nccl: ") or d["fence"] == "nccl", d["fence"]
    if d["fence"] != "nccl":
        assert "the RCCL fence is not available" in r.stderr


@pytest.mark.gpu
def test_soak_every_output_frame_of_the_bench_plan():
    """tools/soak.py: the bench plan (3 streams x 512 slots, sesrq_forward_many) on 24 distinct 1080p frames, EVERY 4K output frame of 25
    rounds compared byte for byte with the one-stream forward (whose frame 0 is checked against the C oracle): a hazard that corrupts one
    store in 10^5 escapes a single whole-frame check (profiles/r04_soak.txt: 96 000 frames, 0 bad)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak.py"), "25"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "SOAK ok 600 frames" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])
