#!/usr/bin/env python3
"""bench.py -- INT8 SESR frames/s on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (sesrq_forward: quantise -> 5 fused conv/requant layers ->
pixel-shuffle) over one batch of synthetic frames already resident in HBM.  Workload at N=1 =
BASELINE.json configs[1]: SESR-x2 INT8 1080p -> 4K, single frame per step.  Frames shard across
ranks with no data-path collective (weak scaling: every rank runs the same per-GPU batch).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "sesr-pytorch-quantize_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK = 8.0e12            # B/s, MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md)

WORKLOADS = {
    # name: (bundle fixture, Cin, H, W, description)
    "sesr_x2_1080p": ("sesr_x2_rand.crop.npz", 3, 1080, 1920, "SESR-x2 INT8 1080p->4K (3->12ch, PixelShuffle 2)"),
    "sesr_x4_540p": ("sesr_x4.crop.npz", 1, 540, 960, "SESR-x4 INT8 540p->4K (1->16ch, PixelShuffle 4)"),
    "nrdm_3_540p": ("nrdm_3.crop.npz", 3, 540, 960, "nrdm_3 INT8 960x540 denoise+demosaic (3->3ch)"),
}


def layer_bytes_per_px(bundle, k, in_f32=True):
    """Algorithmic HBM bytes per input pixel of layer k (each int8 NHWC activation
    written once + read once; fp32 frame in; int8 frame out; shortcut re-read at L-2)."""
    L = bundle.L
    cin = bundle.in_channels
    cout_last = int(bundle.layers[-1].wq.shape[0])
    rd = (4 * cin if in_f32 else cin) if k == 0 else 16
    wr = cout_last if k == L - 1 else 16
    if k == L - 2:
        rd += 16
    return rd + wr


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=1, help="frames per step per GPU")
    ap.add_argument("--streams", type=int, default=2,
                    help="HIP streams the steps are enqueued on round-robin (frames are independent; each stream has "
                         "its own workspace and output buffer) -- fills the launch/prologue/tail gaps of the 5 kernels")
    ap.add_argument("--workload", default="sesr_x2_1080p", choices=sorted(WORKLOADS))
    ap.add_argument("--engine", default="auto", choices=["auto", "dot4", "mfma"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1:
        # not launched by torchrun: start the ranks as children (before anything touches the GPU)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(29500 + os.getpid() % 2000), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd).returncode)

    import numpy as np
    import torch
    import sesrq
    from sesrq import _lib
    from sesrq.bundle import Bundle

    from sesrq.dist import Group, env_world
    rank, local, world = env_world()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device")
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    grp = Group(backend="nccl", device=dev)      # RCCL; only the timing fence uses it

    fixture, cin, H, W, desc = WORKLOADS[args.workload]
    bundle = Bundle.load(os.path.join(ROOT, "tests", "golden", fixture))
    eng = sesrq.Engine(bundle, dev, engine={"auto": _lib.ENGINE_AUTO, "dot4": _lib.ENGINE_DOT4, "mfma": _lib.ENGINE_MFMA}[args.engine])
    B = args.batch
    g = torch.Generator().manual_seed(1 + rank)
    x = torch.rand((B, cin, H, W), generator=g, dtype=torch.float32).to(dev)
    NS = max(1, args.streams)
    streams = [torch.cuda.Stream(device=dev) for _ in range(NS)]
    outs = [torch.empty(eng.out_shape(B, H, W), dtype=torch.int8, device=dev) for _ in range(NS)]
    out_q = outs[0]
    torch.cuda.synchronize()
    counter = [0]

    def step():
        i = counter[0] % NS
        counter[0] += 1
        eng.forward(x, want_q=True, want_f=False, out_q=outs[i], stream=streams[i], slot=i)

    def fence():
        grp.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = grp.max_over_ranks(time.perf_counter() - t0)

    result = None
    if rank == 0:
        frames = args.steps * B * world
        fps = frames / elapsed
        # ---- roofline of the dominant kernel: HIP events on the launch stream, inside this process
        layer_ms, fwd_ms = eng.forward_timed(x, iters=max(10, min(50, args.steps)))
        kdom = int(np.argmax(layer_ms))
        px = B * H * W
        alg = layer_bytes_per_px(bundle, kdom) * px
        ach = alg / (layer_ms[kdom] * 1e-3)
        total_alg = sum(layer_bytes_per_px(bundle, k) for k in range(bundle.L)) * px
        # HBM bytes per launch of that kernel from the committed rocprofv3 PMC passes (profiles/): FETCH_SIZE and
        # WRITE_SIZE in separate runs, FETCH_SIZE doubled for 16-B/lane streaming reads (MI355X_MICROARCH.md, HBM)
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.isfile(tfile):
            tj = json.load(open(tfile))
            traffic = tj.get(args.workload, {}).get(f"layer{kdom}", {}).get("hbm_bytes_per_launch")
        roofline = {"bound": "hbm", "achieved": round(ach / 1e9, 2), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK, 4), "traffic": traffic,
                    "kernel": f"layer{kdom}:{eng.layer_engines()[kdom]}",
                    "kernel_ms": round(layer_ms[kdom], 5), "algorithmic_bytes_per_launch": alg,
                    "layer_ms": [round(v, 5) for v in layer_ms],
                    "layer_frac": [round(layer_bytes_per_px(bundle, k) * px / (layer_ms[k] * 1e-3) / HBM_PEAK, 4) for k in range(bundle.L)],
                    "forward_device_ms": round(fwd_ms, 5),
                    "whole_forward_frac": round(total_alg / (fwd_ms * 1e-3) / HBM_PEAK, 4),
                    "throughput_frac": round(total_alg * fps / world / HBM_PEAK, 4),
                    "note": "layer_ms: HIP events around each launch on one stream (sesrq_forward_timed); throughput_frac = "
                            "algorithmic bytes of a whole forward x frames/s/GPU / peak"}

        # ---- parity spot-check against the oracle (checker only): crop with a 7-px halo
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from oracle import sesrq_oracle as O, c_oracle as CO
        fx = np.load(os.path.join(ROOT, "tests", "golden", fixture), allow_pickle=False)
        onet = O.net_from_fixture(fx)
        r = bundle.pixel_shuffle
        y0, x0, h, w = H // 3, W // 2, 96, 160
        crop = x[0:1, :, y0 - 7:y0 + h + 7, x0 - 7:x0 + w + 7].contiguous()
        want = CO.forward(onet, crop.cpu().numpy(), want_f=False)["q_out"][:, :, 7 * r:(7 + h) * r, 7 * r:(7 + w) * r]
        got = out_q[0:1, :, y0 * r:(y0 + h) * r, x0 * r:(x0 + w) * r].cpu().numpy()
        maxdiff = int(np.abs(got.astype(np.int32) - want.astype(np.int32)).max())
        parity = {"checked": f"{h}x{w} interior crop of frame 0 vs C oracle", "max_abs_diff_int8": maxdiff,
                  "psnr_db": "inf" if maxdiff == 0 else float(10 * np.log10(255.0 ** 2 / np.mean((got.astype(np.float64) - want) ** 2)))}

        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            thr = min(os.cpu_count() or 1, 16)
            xs = x[0:1].cpu().numpy()
            CO.forward(onet, xs[:, :, :64, :64], threads=thr, want_f=False)       # warm the thread pool
            t1 = time.perf_counter()
            CO.forward(onet, xs, threads=thr, want_f=False)
            dt = time.perf_counter() - t1
            cpu = {"value": round(1.0 / dt, 4), "unit": "frames/s", "cores": thr, "kind": "port",
                   "sample": f"1 frame {H}x{W} of the same workload through oracle/sesrq_oracle.c (OpenMP, {thr} threads), {dt:.2f} s"}

        result = {"metric": "INT8 SESR frames/sec (whole job) + PSNR-vs-ref-sim (bit-exact)", "value": round(fps, 2),
                  "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                  "ms_per_step": round(elapsed / args.steps * 1e3, 5), "higher_is_better": True, "scaling": "weak",
                  "vs_baseline": None, "dtype": "i8", "data": "synthetic",
                  "config": {"workload": desc, "name": args.workload, "frames_per_step_per_gpu": B, "streams": NS, "in": [B, cin, H, W],
                             "out": list(eng.out_shape(B, H, W)), "input_dtype": "f32", "output_dtype": "i8",
                             "weights": ("reference random-init net, calibrated by the reference" if "rand" in fixture else
                                         "reference checkpoint, quantised and calibrated by the reference") + f" ({fixture})",
                             "sharding": f"frames x{world}, no collective", "engines": eng.layer_engines()},
                  "roofline": roofline, "cpu_baseline": cpu, "parity": parity}
    grp.close()
    if result is not None:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
