"""Calibration entry point -- the counterpart of the reference's ``test.py`` (:79-113 graph build with
qmode = 0, :141-183 loop over calibration frames, :185-217 scale/zero derivation).  The reference's
dataset is private; frames come from ``--frames`` (.npy / .pt, (N,C,H,W) float32, one batch per N).

    python test.py --mflag 5 --params tests/golden/sesr_x4.params.npz --frames tests/golden/rand_SR_Input_80x960.npy \\
                   --save-bundle x4.bundle.npz
"""
import argparse

import numpy as np
import torch

import define
from define import QUAN_BIT, PE, BIAS_BIT, PE_ACC_BIT, PE_ADD_BIT
from myQL.quan_func import quantize_model_weight, quantize_asymmetrical_by_tensor, reshape_input_for_hardware_pe, PEs_and_bias_adder
from myQL.quan_classes import NodeInsertMapping, FunctionPackage, NodeInsertMappingElement
from myQL.graph_modify import insert_before, insert_bias_bypass
from sesrq.store import STORE
from sesrq.calibrate import finish_calibration
import sim


def splice_calibration(model):
    """The three graph rewrites of the reference's calibration script, qmode = 0 (test.py:79-106)."""
    qmode = 0
    model = quantize_model_weight(model, QUAN_BIT, qmode)
    mapping = NodeInsertMapping()
    quan = FunctionPackage(quantize_asymmetrical_by_tensor, {"width": QUAN_BIT, "exe_mode": qmode})
    mapping.add_config(NodeInsertMappingElement(torch.nn.Conv2d, quan))
    mapping.add_config(NodeInsertMappingElement(torch.nn.PixelShuffle, quan))
    model = insert_before(model_input=model, insert_mapping=mapping, has_func_id=True)
    m2 = NodeInsertMapping()
    m2.add_config(NodeInsertMappingElement(torch.nn.Conv2d, FunctionPackage(reshape_input_for_hardware_pe, {"pe_num": PE})))
    model = insert_before(model_input=model, insert_mapping=m2)
    m3 = NodeInsertMapping()
    m3.add_config(NodeInsertMappingElement(torch.nn.Conv2d, FunctionPackage(
        PEs_and_bias_adder, {"pe_add_width": PE_ADD_BIT, "pe_acc_width": PE_ACC_BIT, "bias_width": BIAS_BIT, "pe_num": PE,
                             "exe_mode": qmode})))
    return insert_bias_bypass(model_input=model, insert_mapping=m3)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--mflag", type=int, default=define.MFLAG)
    ap.add_argument("--ckpt")
    ap.add_argument("--params")
    ap.add_argument("--frames", required=True)
    ap.add_argument("--method", default="minmax", choices=["minmax", "entropy"],
                    help="minmax: the reference's running min/max (default); entropy: KL-minimising clipping ranges from a second "
                         "pass over the frames (no reference counterpart, parity unpinned)")
    ap.add_argument("--bins", type=int, default=2048, help="histogram bins of --method entropy")
    ap.add_argument("--save-bundle")
    ap.add_argument("--save-output-pt", help="directory to write input.K.{min_val,max_val,scale,zero}.pt like the reference")
    args = ap.parse_args(argv)
    define.check()
    STORE.clear()
    model = splice_calibration(sim.float_model(args.mflag, args.ckpt, args.params))
    frames = torch.load(args.frames, weights_only=True, map_location="cpu") if args.frames.endswith(".pt") else torch.from_numpy(np.load(args.frames))
    if not torch.cuda.is_available():
        raise SystemExit("test.py: calibration runs on a HIP device (no CPU fallback)")
    with torch.no_grad():
        for i in range(frames.shape[0]):
            model(frames[i:i + 1].float().cuda())
    print("calibrate start")
    cal = model._sesrq_cal
    if args.method == "entropy":
        cal.set_method("entropy", args.bins)
        cal.begin_histogram_pass()
        with torch.no_grad():
            for i in range(frames.shape[0]):
                model(frames[i:i + 1].float().cuda())
        scale, zero = cal.finalize()
        STORE.set_activation_domains(scale, zero)
    else:
        scale, zero = finish_calibration(STORE, cal.L)
    for s, z in zip(scale, zero):
        print("scale:", s)
        print("zero:", z)
    print("calibrate end")
    print("bit:", QUAN_BIT)
    if args.save_output_pt:
        STORE.save_output_pt(args.save_output_pt)
    if args.save_bundle:
        model._sesrq_cal.bundle(name=f"mflag{args.mflag}").save(args.save_bundle)
    return scale, zero


if __name__ == "__main__":
    main()
