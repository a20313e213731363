#!/usr/bin/env python3
"""Golden-vector generator.  RUNS ONLY IN THE BUILD CONTAINER (needs /root/reference).

It imports the reference's own functions (myQL.quan_func / myQL.graph_modify /
models.*), re-drives the graph construction of the reference's calibration script
(test.py:79-106,185-217) and integer-simulation script (sim.py:82-114,205) on CPU in a
scratch working directory, and harvests the ``output_pt/`` tree the reference writes into
small ``.npz`` fixtures (plain arrays + a JSON string, loadable with allow_pickle=False).

The fixtures define CPU-REFERENCE semantics: the reference as shipped runs on ``.cuda()``, where torch's tensor / Python-scalar
division evaluates ``x * fl(1/s)`` and the input quantiser can differ by one LSB at rint ties from this CPU run (true fp32
division).  No fixture covers a device run of the reference: parity against that variant is unpinned.

Nothing of the reference travels: the fixtures hold inputs, parameters and expected
outputs only.  The reference's two random input tensors (data files) are re-saved as
``.npy``.

Usage:  python tests/golden/make_golden.py            # all cases (spawns one process per net)
        python tests/golden/make_golden.py --case sesr_x4
        python tests/golden/make_golden.py --case time_x2_1080p   # the reference's CPU sim path TIMED on bench.py's headline workload
                                                                   # (SESR-x2 random-init net, 1x3x1080x1920, dump flags off) + SHA-256 of its
                                                                   # int8 / fp32 output -> reference_x2_1080p.json
        python tests/golden/make_golden.py --case anchor          # AnchorOp (sesr_arch.py:171-205) + the eval loop's x2 add (test.py:148-155)
        python tests/golden/make_golden.py --case sesr_x4_nat     # round 5: a natural-ish frame (tests/golden/natural.py), CALIBRATED ON IT by the
                                                                   # reference (zero_0 < -128 with its consistent scale): crop with every stage, 80x960
                                                                   # SHAs, and a BASELINE-size frame through the reference's sim path (*.big.json)
        python tests/golden/make_golden.py --case nrdm_3 --fuzz 30   # numpy oracle vs the reference on 30 random small frames, each calibrated
                                                                   # by the reference; writes nothing (log: profiles/r05_oracle_vs_reference_fuzz.txt)
"""
import argparse
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

# case -> (MFLAG, float model factory, sim model factory, checkpoint, input tensor, qat)
CASES = {
    "sesr_x4":      dict(mflag=5, ckpt="model_params/x4sesr.pth", inp="rand_SR_Input_80x960.pt", qat=False),
    "sesr_x4_qat":  dict(mflag=5, ckpt="model_params/sr_qat_G.pth", inp="rand_SR_Input_80x960.pt", qat=True),
    "nrdm_3":       dict(mflag=3, ckpt="model_params/nrdm_3_raw_G.pth", inp="rand_DM_Input_80x960.pt", qat=False),
    "nrdm_3_qat":   dict(mflag=3, ckpt="model_params/nrdm_3_qat_G.pth", inp="rand_DM_Input_80x960.pt", qat=True),
    # x2sesr.pth.tar is refused by torch.load(weights_only=True) (pickled optimizer object),
    # so the x2 topology is driven with the reference's own random initialisation, seeded.
    "sesr_x2_rand": dict(mflag=6, ckpt=None, inp="rand_DM_Input_80x960.pt", qat=False, seed=1234),
    # Round 5: the same nets on a NATURAL-ISH frame (tests/golden/natural.py: gradients + edges + flat areas + 2 % noise, min > 0) and
    # CALIBRATED ON THAT FRAME by the reference's own mode-0 pass (test.py:185-217 semantics): zero_0 < -128 together with its consistent
    # scale, bias constants that are not saturated -- what a real image gives and no noise fixture has.  big: a BASELINE-size frame of
    # the same kind run through the reference's sim path with the same calibration (dump flags off), pinned by SHA-256.
    "sesr_x4_nat":      dict(mflag=5, ckpt="model_params/x4sesr.pth", inp="natural", qat=False, nat_seed=2024, big=(540, 960)),
    "nrdm_3_nat":       dict(mflag=3, ckpt="model_params/nrdm_3_raw_G.pth", inp="natural", qat=False, nat_seed=2025, big=(540, 960)),
    "sesr_x2_rand_nat": dict(mflag=6, ckpt=None, inp="natural", qat=False, seed=1234, nat_seed=2026, big=(1080, 1920)),
    # ... and the QAT checkpoints BASELINE.json's configs name (config 0: sr_qat_G.pth, config 2: nrdm_3_qat_G.pth at 960x540)
    "sesr_x4_qat_nat":  dict(mflag=5, ckpt="model_params/sr_qat_G.pth", inp="natural", qat=True, nat_seed=2027, big=(540, 960)),
    "nrdm_3_qat_nat":   dict(mflag=3, ckpt="model_params/nrdm_3_qat_G.pth", inp="natural", qat=True, nat_seed=2028, big=(540, 960)),
}
CROP_H, CROP_W = 24, 40


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def run_case(name: str, out_dir: str, time_1080p: bool = False, fuzz: int = 0) -> None:
    cfg = CASES[name]
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    import torch
    from torch import nn
    import define
    define.MFLAG = cfg["mflag"]            # bound by value inside quan_func at import time
    from myQL import quan_func as qf
    from myQL.quan_classes import NodeInsertMapping, FunctionPackage, NodeInsertMappingElement
    from myQL.graph_modify import insert_before, insert_bias_bypass, insert_after
    from models import sesr, sesr_sim, nrdm_3, nrdm_3_sim, sesr_arch, sesr_arch_sim

    torch.manual_seed(0)
    scratch = tempfile.mkdtemp(prefix="golden_", dir=os.path.join(HERE, "..", "..", ".scratch"))
    os.chdir(scratch)

    float_cls, sim_cls = {5: (sesr.sesr, sesr_sim.sesr), 3: (nrdm_3.nr, nrdm_3_sim.nr),
                          6: (sesr_arch.sesr, sesr_arch_sim.sesr)}[cfg["mflag"]]

    def load_into(model):
        model.train()
        if cfg["qat"]:
            from models import quantize_utils_pt as quantize
            quantize.prepare(model, inplace=True, a_bits=8, w_bits=8, q_type=0, q_level="C")
        if cfg["ckpt"] is not None:
            sd = torch.load(os.path.join(REF, cfg["ckpt"]), weights_only=True, map_location="cpu")
            model.load_state_dict(sd, strict=False)
        model = model.float()
        model.collapse()
        return model

    if cfg["ckpt"] is None:
        # random-init: the float and the sim model must share weights -> build once, copy.
        torch.manual_seed(cfg["seed"])
        proto = float_cls()
        proto_sd = {k: v.clone() for k, v in proto.state_dict().items()}

    def make(cls):
        m = cls()
        if cfg["ckpt"] is None:
            m.load_state_dict(proto_sd, strict=False)
        return load_into(m)

    def pack(fn, kw):
        mp = NodeInsertMapping()
        mp.add_config(NodeInsertMappingElement(nn.Conv2d, FunctionPackage(fn, kw)))
        return mp

    def splice(model, qmode):
        model = qf.quantize_model_weight(model, define.QUAN_BIT, qmode)
        mp = NodeInsertMapping()
        fp = FunctionPackage(qf.quantize_asymmetrical_by_tensor, {"width": define.QUAN_BIT, "exe_mode": qmode})
        mp.add_config(NodeInsertMappingElement(nn.Conv2d, fp))
        if qmode == 0:
            mp.add_config(NodeInsertMappingElement(nn.PixelShuffle, fp))
        model = insert_before(model_input=model, insert_mapping=mp, has_func_id=True)
        model = insert_before(model_input=model,
                              insert_mapping=pack(qf.reshape_input_for_hardware_pe, {"pe_num": define.PE}))
        if qmode == 1:
            model = insert_after(model_input=model, insert_mapping=pack(qf.requan_conv2d_output, {"exe_mode": 1}))
        model = insert_bias_bypass(model_input=model, insert_mapping=pack(
            qf.PEs_and_bias_adder, {"pe_add_width": define.PE_ADD_BIT, "pe_acc_width": define.PE_ACC_BIT,
                                    "bias_width": define.BIAS_BIT, "pe_num": define.PE, "exe_mode": qmode}))
        return model

    natural = cfg["inp"] == "natural"
    sys.path.insert(0, HERE)
    from natural import natural_frame, interesting_crop
    if natural:
        x_full = torch.from_numpy(natural_frame(1 if cfg["mflag"] == 5 else 3, 80, 960, cfg["nat_seed"]))
    else:
        x_full = torch.load(os.path.join(REF, cfg["inp"]), weights_only=True, map_location="cpu").float()
    if x_full.shape[1] != (1 if cfg["mflag"] == 5 else 3):
        raise SystemExit("unexpected input shape")

    # ---------------- float collapsed weights (for the weight/bias quantiser fixtures)
    fm = make(sim_cls)
    convs = [fm.conv_first.conv_expand] + [b.conv_expand for b in fm.residual_block] + [fm.conv_last.conv_expand]
    Wf = [c.weight.detach().numpy().copy() for c in convs]
    bf = [c.bias.detach().numpy().copy() for c in convs]

    # ---------------- mode 0: calibration on the full random input (test.py semantics)
    def calibrate(x_cal):
        import glob
        for f in glob.glob("output_pt/input/input.*.m??_val.pt"):      # test.py:108-113 resets the running ranges before its loop
            os.remove(f)
        cal = splice(make(float_cls), 0)
        with torch.no_grad():
            y_c = cal(x_cal)
        QMAX, QMIN = 127, -128
        mins_, maxs_, scales_, zeros_ = [], [], [], []
        for i in range(6):
            mx = torch.load(f"output_pt/input/input.{i}.max_val.pt")
            mn = torch.load(f"output_pt/input/input.{i}.min_val.pt")
            mins_.append(mn); maxs_.append(mx)
            if i == 5:
                mn = 0
            s = (mx - mn) / (QMAX - QMIN)
            z = QMIN - round(mn / s)
            torch.save(s, f"output_pt/input/input.{i}.scale.pt")
            torch.save(z, f"output_pt/input/input.{i}.zero.pt")
            scales_.append(float(s)); zeros_.append(int(z))
        return y_c, mins_, maxs_, scales_, zeros_
    y_cal, mins, maxs, scales, zeros = calibrate(x_full)

    # ---------------- mode 1 runs
    def ld(p):
        return torch.load(p)

    def harvest(x, y, tag, full, save=True):
        d = {}
        L = 5
        for k in range(L):
            d[f"Wq{k}"] = ld(f"output_pt/weight/conv.weight.{k}.pt").numpy().astype(np.int8)
            d[f"add_const{k}"] = ld(f"output_pt/bias/conv.bias.quan{k}.pt").numpy().reshape(-1).astype(np.int32)
        wscale = [float(ld(f"output_pt/weight/conv.weight.{k}.scale.pt")) for k in range(L)]
        sc = [float(ld(f"output_pt/input/input.{k}.scale.pt")) for k in range(6)]
        zr = [int(ld(f"output_pt/input/input.{k}.zero.pt")) for k in range(6)]
        names = ["0_1", "1_2", "2_3", "3_4", "4_5"]
        M = [int(ld(f"output_pt/requan_factor/requan_{n}.pt")) for n in names]
        nn_ = [int(ld(f"output_pt/requan_factor/n_{n}.pt")) for n in names]
        meta = dict(case=name, tag=tag, mflag=cfg["mflag"], wscale=wscale, scale=sc, zero=zr, M=M, n=nn_,
                    M_res=int(ld("output_pt/requan_factor/requan_res.pt")),
                    n_res=int(ld("output_pt/requan_factor/n_res.pt")),
                    H=int(x.shape[2]), W=int(x.shape[3]), out_shape=list(y.shape), sha={})
        if natural:
            meta.update(input="natural", nat_seed=cfg["nat_seed"], x_sha256=sha(x.numpy().astype(np.float32)))
        acts = {f"input{k}": ld(f"output_pt/input/input.{k}.pt").numpy() for k in range(6)}
        acts["input4_special"] = ld("output_pt/input/input.4.spcial.pt").numpy()
        for k, v in acts.items():
            assert np.all(v == np.rint(v)) and v.min() >= -128 and v.max() <= 127, k
            acts[k] = v.astype(np.int8)
        acts["shortcut"] = ld("output_pt/residual/shortcut_tensor.pt").numpy().astype(np.float32)
        for k in range(L):
            v = ld(f"output_pt/pe_add/pe_add_output{k}.pt").numpy()
            assert np.all(v == np.rint(v))
            acts[f"pe_add{k}"] = v.astype(np.int32)
            pes = np.stack([ld(f"output_pt/pe_out/pe_output{k}_{p}.pt").numpy() for p in range(4)])
            assert np.all(pes == np.rint(pes))
            acts[f"pe_out{k}"] = pes.astype(np.int32)
        acts["out"] = y.detach().numpy().astype(np.float32)
        for k, v in acts.items():
            meta["sha"][k] = sha(v)
        if full:
            # full frames: keep only the input, the final int8 tensor; the rest is pinned by SHA-256
            keep = {"input5": acts["input5"], "input1": acts["input1"]}
        else:
            keep = acts
        d.update(keep)
        d["x"] = x.numpy().astype(np.float32) if not full else np.zeros(0, np.float32)
        d["meta"] = np.array(json.dumps(meta))
        if not save:
            return d
        np.savez_compressed(os.path.join(out_dir, f"{name}.{tag}.npz"), **d)
        print(f"[{name}.{tag}] M={M} n={nn_} res=({meta['M_res']},{meta['n_res']}) zero={zr}", flush=True)

    def sim_run(x, mutate_weights=None):
        m = make(sim_cls)
        m = qf.quantize_model_weight(m, define.QUAN_BIT, 1)
        if mutate_weights is not None:
            sd = m.state_dict()
            k = 0
            for pn in sd:
                if pn.endswith("conv_expand.weight"):
                    sd[pn] = mutate_weights(sd[pn], k)
                    torch.save(sd[pn].clone(), f"output_pt/weight/conv.weight.{k}.pt")
                    k += 1
            m.load_state_dict(sd)
        # graph rewrites (weights already quantised above -> call the three splicers directly)
        mp = NodeInsertMapping()
        mp.add_config(NodeInsertMappingElement(nn.Conv2d, FunctionPackage(
            qf.quantize_asymmetrical_by_tensor, {"width": define.QUAN_BIT, "exe_mode": 1})))
        m = insert_before(model_input=m, insert_mapping=mp, has_func_id=True)
        m = insert_before(model_input=m, insert_mapping=pack(qf.reshape_input_for_hardware_pe, {"pe_num": define.PE}))
        m = insert_after(model_input=m, insert_mapping=pack(qf.requan_conv2d_output, {"exe_mode": 1}))
        m = insert_bias_bypass(model_input=m, insert_mapping=pack(
            qf.PEs_and_bias_adder, {"pe_add_width": define.PE_ADD_BIT, "pe_acc_width": define.PE_ACC_BIT,
                                    "bias_width": define.BIAS_BIT, "pe_num": define.PE, "exe_mode": 1}))
        with torch.no_grad():
            return m(x)

    if fuzz:
        # ORACLE vs REFERENCE, many small frames (round 5): every trial draws a frame (natural-ish or noise, random size and level), lets the
        # REFERENCE calibrate on it (mode 0) and run its integer simulation (mode 1), and compares the numpy oracle with every tensor the
        # reference wrote -- no fixture is kept, the log is (profiles/r05_oracle_vs_reference_fuzz.txt).
        sys.path.insert(0, os.path.join(HERE, "..", ".."))
        from oracle import sesrq_oracle as O
        rng = np.random.default_rng(4242 + cfg["mflag"])
        C = x_full.shape[1]
        bad_total = 0
        for t in range(fuzz):
            h, w = int(rng.integers(7, 41)), int(rng.integers(9, 73))
            kind = ("natural", "noise", "dim natural")[t % 3]
            if kind == "noise":
                xs = torch.rand((1, C, h, w), generator=torch.Generator().manual_seed(7000 + t)) * float(rng.uniform(0.3, 1.0)) + float(rng.uniform(0.0, 0.2))
            else:
                xs = torch.from_numpy(natural_frame(C, h, w, 6000 + t))
                if kind == "dim natural":
                    xs = xs * float(rng.uniform(0.2, 0.6))
            xs = xs.float().contiguous()
            _, _, _, sc_t, ze_t = calibrate(xs)
            y = sim_run(xs)
            d = harvest(xs, y, "fuzz", full=False, save=False)
            net = O.net_from_fixture(d)
            st = O.forward(net, d["x"], keep=True)
            bad = 0
            for k_, v_ in d.items():
                if k_ in ("meta", "x") or k_.startswith(("Wq", "add_const")):
                    continue
                got = st["y" if k_ == "out" else k_]
                bad += int((np.asarray(got).astype(v_.dtype) != v_).sum())
            bad_total += bad
            print(f"[fuzz {name} {t:3d}] {kind:11s} {h:2d}x{w:2d} zero={ze_t} mismatches={bad}", flush=True)
        print(f"[fuzz {name}] {fuzz} reference runs, {bad_total} mismatching values in all dumped tensors", flush=True)
        os.chdir(HERE)
        shutil.rmtree(scratch, ignore_errors=True)
        if bad_total:
            raise SystemExit(1)
        return

    # calibration record + float weights
    np.savez_compressed(
        os.path.join(out_dir, f"{name}.params.npz"),
        meta=np.array(json.dumps(dict(case=name, mflag=cfg["mflag"], min=mins, max=maxs, scale=scales, zero=zeros,
                                      cal_out_sha=sha(y_cal.numpy().astype(np.float32))))),
        **{f"Wf{k}": Wf[k] for k in range(5)}, **{f"bf{k}": bf[k] for k in range(5)})

    if time_1080p:
        # bench.py's headline workload through the reference's own sim path: pool frame 0 of rank 0 (torch.Generator().manual_seed(1),
        # torch.rand((1, 3, 1080, 1920))), the seven dump switches off (bound by value inside quan_func at import: set on the module)
        import time
        for flg in ("WEIGHT_W_FLG", "INPUT_W_FLG", "BIAS_W_FLG", "BIAS_QUAN_W_FLG", "OUTPUT_PE_W_FLG", "OUTPUT_PE_ADD_W_FLG", "REQUAN_FACTOR_W_FLG"):
            assert hasattr(qf, flg), flg
            setattr(qf, flg, False)
        g = torch.Generator().manual_seed(1)
        x = torch.rand((1, 3, 1080, 1920), generator=g, dtype=torch.float32)
        sim_run(x[:, :, :64, :64].contiguous())                 # warm-up (thread pool, graph tracing paths)
        secs = []
        for _ in range(3):
            t0 = time.perf_counter()
            y = sim_run(x)
            secs.append(time.perf_counter() - t0)
        # graph construction (quantize_model_weight + four fx rewrites) is inside sim_run like it is inside sim.py's run; the forward alone:
        # dump flags are off, so input.5 is not on disk: the int8 frame is recovered from the float result, y = (q5 - z5) * f32(s5) after PixelShuffle
        s5 = np.float32(torch.load("output_pt/input/input.5.scale.pt")); z5 = int(torch.load("output_pt/input/input.5.zero.pt"))
        yq = np.rint(y.numpy().astype(np.float64) / np.float64(s5) + z5)
        assert yq.min() >= -128 and yq.max() <= 127
        yq = yq.astype(np.int8)
        assert np.array_equal(((yq.astype(np.float32) - np.float32(z5)) * s5).astype(np.float32), y.numpy().astype(np.float32))
        rec = dict(workload="SESR-x2 (sesr_arch_sim.sesr, the reference's seeded random init = tests/golden/sesr_x2_rand.*), 1x3x1080x1920 -> 1x3x2160x3840, "
                            "the reference's own sim path (quantize_model_weight + 4 fx rewrites + forward, sim.py:82-114,205), dump flags off",
                   input="torch.rand((1,3,1080,1920), generator=torch.Generator().manual_seed(1)) = bench.py pool frame 0 of rank 0",
                   seconds=[round(t, 3) for t in secs], seconds_median=round(sorted(secs)[1], 3), frames_per_s=round(1.0 / sorted(secs)[1], 4),
                   cores=torch.get_num_threads(), host="build container: Intel Xeon @ 2.1 GHz, 8 vCPU (nproc %d)" % (os.cpu_count() or 0),
                   torch=torch.__version__, out_shape=list(y.shape), out_q_sha256=sha(yq), out_f_sha256=sha(y.numpy().astype(np.float32)),
                   x_sha256=sha(x.numpy()))
        json.dump(rec, open(os.path.join(out_dir, "reference_x2_1080p.json"), "w"), indent=1)
        print(json.dumps(rec, indent=1), flush=True)
        os.chdir(HERE)
        shutil.rmtree(scratch, ignore_errors=True)
        return

    # (1) full frame
    y = sim_run(x_full)
    harvest(x_full, y, "full", full=True)
    if natural:
        # the crop with every stage stored: where the frame has an edge, a flat area and gradient inside 24 x 40
        cy, cx = interesting_crop(x_full.numpy(), CROP_H, CROP_W)
        x_crop = x_full[:, :, cy:cy + CROP_H, cx:cx + CROP_W].contiguous()
        y = sim_run(x_crop)
        harvest(x_crop, y, "crop", full=False)
        # a BASELINE-size frame of the same kind through the reference's sim path, same calibration, the seven dump switches off
        # (bound by value inside quan_func at import: set on the module); the int8 frame is recovered from the float result
        import time
        for flg in ("WEIGHT_W_FLG", "INPUT_W_FLG", "BIAS_W_FLG", "BIAS_QUAN_W_FLG", "OUTPUT_PE_W_FLG", "OUTPUT_PE_ADD_W_FLG", "REQUAN_FACTOR_W_FLG"):
            assert hasattr(qf, flg), flg
            setattr(qf, flg, False)
        Hb, Wb = cfg["big"]
        xb = torch.from_numpy(natural_frame(x_full.shape[1], Hb, Wb, cfg["nat_seed"] + 100))
        t0 = time.perf_counter()
        yb = sim_run(xb)
        secs = time.perf_counter() - t0
        s5 = np.float32(torch.load("output_pt/input/input.5.scale.pt")); z5 = int(torch.load("output_pt/input/input.5.zero.pt"))
        yq = np.rint(yb.numpy().astype(np.float64) / np.float64(s5) + z5)
        assert yq.min() >= -128 and yq.max() <= 127
        yq = yq.astype(np.int8)
        assert np.array_equal(((yq.astype(np.float32) - np.float32(z5)) * s5).astype(np.float32), yb.numpy().astype(np.float32))
        rec = dict(case=name, bundle=f"{name}.crop.npz", input=f"natural_frame({x_full.shape[1]}, {Hb}, {Wb}, seed={cfg['nat_seed'] + 100}) (tests/golden/natural.py)",
                   nat_seed=cfg["nat_seed"] + 100, in_shape=list(xb.shape), out_shape=list(yb.shape), seconds=round(secs, 3),
                   x_sha256=sha(xb.numpy()), out_q_sha256=sha(yq), out_f_sha256=sha(yb.numpy().astype(np.float32)),
                   cores=torch.get_num_threads(), torch=torch.__version__)
        json.dump(rec, open(os.path.join(out_dir, f"{name}.big.json"), "w"), indent=1)
        print(json.dumps(rec, indent=1), flush=True)
        os.chdir(HERE)
        shutil.rmtree(scratch, ignore_errors=True)
        return
    # (2) crop, all stages kept
    x_crop = x_full[:, :, 8:8 + CROP_H, 100:100 + CROP_W].contiguous()
    y = sim_run(x_crop)
    harvest(x_crop, y, "crop", full=False)

    # (2b) hardware stimulus text: run the reference's own dump scripts on a 40x72 crop (2x3 tiles of 32)
    if name in ("sesr_x4", "nrdm_3"):
        import runpy
        x_st = x_full[:, :, 3:3 + 40, 200:200 + 72].contiguous()
        shutil.rmtree("output_txt", ignore_errors=True)
        y = sim_run(x_st)
        harvest(x_st, y, "stim", full=False)
        txt = {}

        def grab(prefix):
            for root, _, files in os.walk("output_txt"):
                for f in sorted(files):
                    rel = os.path.relpath(os.path.join(root, f), "output_txt")
                    txt[prefix + rel] = np.frombuffer(open(os.path.join(root, f), "rb").read(), dtype=np.uint8)
        runpy.run_path(os.path.join(REF, "output.py"), run_name="__main__")
        grab("output/")
        shutil.rmtree("output_txt/input", ignore_errors=True)
        runpy.run_path(os.path.join(REF, "output_end2end.py"), run_name="__main__")
        for f in sorted(os.listdir("output_txt/input")):
            txt["end2end/input/" + f] = np.frombuffer(open(os.path.join("output_txt/input", f), "rb").read(), dtype=np.uint8)
        np.savez_compressed(os.path.join(out_dir, f"{name}.stimtxt.npz"), **{k.replace("/", "|"): v for k, v in txt.items()})
        print(f"[{name}.stimtxt] {len(txt)} text files, {sum(v.size for v in txt.values())} bytes", flush=True)

    def set_zero(vals):
        for k, v in vals.items():
            torch.save(int(v), f"output_pt/input/input.{k}.zero.pt")

    if not cfg["qat"]:
        # (3) zero-point excursions: z<-128 on layers 0,2,4 ; z>-128 on layer 1 (rc != q1) ; z5 != -128
        set_zero({0: -140, 1: -120, 2: -131, 4: -150, 5: -119})
        y = sim_run(x_crop)
        harvest(x_crop, y, "zeros", full=False)
        set_zero({k: zeros[k] for k in range(6)})

        # (4) saturating weights: +127 / -128 everywhere -> 18-bit PE and 20-bit adder clamps fire
        def sat(w, k):
            return torch.where(w >= 0, torch.full_like(w, 127.0), torch.full_like(w, -128.0))
        y = sim_run(x_crop, mutate_weights=sat)
        harvest(x_crop, y, "satw", full=False)

        # (5) both at once, different zero pattern (z3 > -128 as well)
        set_zero({0: -129, 1: -100, 2: -128, 3: -90, 4: -127, 5: -128})

        def sat2(w, k):   # checkerboard of signs so that PE sums straddle both clamps
            idx = torch.arange(w.numel()).reshape(w.shape)
            return torch.where((idx // 3 + k) % 2 == 0, torch.full_like(w, 127.0), torch.full_like(w, -128.0))
        y = sim_run(x_crop, mutate_weights=sat2)
        harvest(x_crop, y, "satw_zeros", full=False)
        set_zero({k: zeros[k] for k in range(6)})

    os.chdir(HERE)
    shutil.rmtree(scratch, ignore_errors=True)


def run_tables(out_dir: str) -> None:
    """Host-scalar functions: requant-constant encoder, weight quantiser, bias quantiser."""
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    import torch
    scratch = tempfile.mkdtemp(prefix="golden_", dir=os.path.join(HERE, "..", "..", ".scratch"))
    os.chdir(scratch)
    from myQL import quan_func as qf
    rng = np.random.default_rng(7)
    r = np.concatenate([
        np.exp(rng.uniform(np.log(1e-11), np.log(3e4), 4000)),
        2.0 ** np.arange(-36, 15), 2.0 ** np.arange(-36, 15) * (1 - 2.0 ** -40), 2.0 ** np.arange(-36, 15) * (1 + 2.0 ** -40),
        np.array([1.0, 0.999999999, 1.5, 2.0, 3.0, 255.0, 256.0, 65535.0, 65535.9, 0.5, 0.25, 1 / 3, 1e-10, 2.4e-10]),
    ])
    M = np.zeros(len(r), np.int64); n = np.zeros(len(r), np.int64)
    for i, v in enumerate(r):
        M[i], n[i] = qf.quan_layer_between_const(float(v), 16, 32)
    # weight quantiser on assorted tensors
    wq_in, wq_out, wq_scale = [], [], []
    for i in range(6):
        w = torch.from_numpy(rng.standard_normal((5, 3, 3, 3)).astype(np.float32) * (10.0 ** (i - 3)))
        if i == 4:
            w = torch.round(w * 1e3) / 1e3 * 0.5  # ties
        q = qf.quantize_symmetrical_by_tensor(w, 8, 1, func_id=100 + i)
        wq_in.append(w.numpy()); wq_out.append(q.numpy().astype(np.int8))
        wq_scale.append(float(torch.load(f"output_pt/weight/conv.weight.{100 + i}.scale.pt")))
    np.savez_compressed(os.path.join(out_dir, "tables.npz"), r=r, M=M, n=n,
                        wq_in=np.stack(wq_in), wq_out=np.stack(wq_out), wq_scale=np.array(wq_scale))
    print("[tables] n range", n.min(), n.max(), "M max", M.max(), flush=True)
    os.chdir(HERE)
    shutil.rmtree(scratch, ignore_errors=True)


def run_anchor(out_dir: str) -> None:
    """The x2 anchor of the reference: AnchorOp (models/sesr_arch.py:171-205: a frozen 1x1 conv that repeats every input channel r^2
    times) followed by the net's own PixelShuffle(r) = nearest-neighbour upsampling of the input; the eval loop adds exactly that to
    the x2 output (test.py:148-155: inps_x2[:, :, i::2, j::2] = inps; gfake + inps_x2).  Fixture: the reference-made upsampled input and
    the reference-made sum for the sesr_x2_rand crop and full frame (their stored float results)."""
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    import torch
    from models.sesr_arch import AnchorOp
    x_full = torch.load(os.path.join(REF, "rand_DM_Input_80x960.pt"), weights_only=True, map_location="cpu").float()
    x_crop = x_full[:, :, 8:8 + CROP_H, 100:100 + CROP_W].contiguous()
    op = AnchorOp(scaling_factor=2, in_channels=3)
    ps = torch.nn.PixelShuffle(2)
    d = {}
    for tag, x in (("crop", x_crop), ("full", x_full)):
        with torch.no_grad():
            up = ps(op(x))                                     # AnchorOp + depth-to-space
        loop = torch.zeros(x.shape[0], x.shape[1], x.shape[2] * 2, x.shape[3] * 2)      # test.py:148-153 verbatim semantics
        loop[:, :, 0::2, 0::2] = x; loop[:, :, 0::2, 1::2] = x; loop[:, :, 1::2, 0::2] = x; loop[:, :, 1::2, 1::2] = x
        assert torch.equal(up, loop), "AnchorOp + PixelShuffle is the eval loop's nearest upsampling"
        g = np.load(os.path.join(out_dir, f"sesr_x2_rand.{tag}.npz"), allow_pickle=False)
        meta = json.loads(str(g["meta"]))
        if tag == "crop":
            y = torch.from_numpy(g["out"])
        else:                                                  # full frames store input5 (before PixelShuffle) only: y = (q5 - z5) * f32(s5)
            y = ps((torch.from_numpy(g["input5"].astype(np.float32)) - float(meta["zero"][5])) * float(np.float32(meta["scale"][5])))
            assert sha(y.numpy().astype(np.float32)) == meta["sha"]["out"]
        s = (y + up).numpy().astype(np.float32)                # gfake + inps_x2 (test.py:154-155)
        d[f"up_{tag}"] = up.numpy().astype(np.float32) if tag == "crop" else np.zeros(0, np.float32)
        d[f"sum_{tag}"] = s if tag == "crop" else np.zeros(0, np.float32)
        d[f"sha_{tag}"] = np.array(json.dumps(dict(up=sha(up.numpy().astype(np.float32)), sum=sha(s), shape=list(s.shape))))
    np.savez_compressed(os.path.join(out_dir, "sesr_x2_rand.anchor.npz"), **d)
    print("[anchor]", {k: (v.shape if v.ndim else str(v)) for k, v in d.items()}, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default=None)
    ap.add_argument("--fuzz", type=int, default=0, help="with --case <net>: N reference runs on random small frames (each calibrated by the reference) "
                                                        "compared with the numpy oracle; nothing is written")
    args = ap.parse_args()
    os.makedirs(os.path.join(HERE, "..", "..", ".scratch"), exist_ok=True)
    if args.case == "tables":
        run_tables(HERE)
    elif args.case == "time_x2_1080p":
        run_case("sesr_x2_rand", HERE, time_1080p=True)
    elif args.case == "anchor":
        run_anchor(HERE)
    elif args.case:
        run_case(args.case, HERE, fuzz=args.fuzz)
    else:
        # inputs (data files of the reference) as .npy
        import torch
        for f in ("rand_SR_Input_80x960", "rand_DM_Input_80x960"):
            t = torch.load(os.path.join(REF, f + ".pt"), weights_only=True, map_location="cpu")
            np.save(os.path.join(HERE, f + ".npy"), t.numpy().astype(np.float32))
        for c in list(CASES) + ["tables", "anchor", "time_x2_1080p"]:
            subprocess.run([sys.executable, os.path.abspath(__file__), "--case", c], check=True)


if __name__ == "__main__":
    main()
