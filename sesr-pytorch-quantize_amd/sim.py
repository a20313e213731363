"""Integer-inference entry point -- the counterpart of the reference's ``sim.py``.

Behaviour reproduced (reference sim.py:29-114, :197-213): pick the network by ``define.MFLAG``
(3 = nrdm_3, 5 = SESR x4, 6 = SESR x2), load float weights, ``collapse()``, quantise the weights
(mode 1), splice the four stage callables around every conv in the reference's order, run ONE
forward on the frame and print the bit-width banner.  Here the forward is a single fused device
call; activation domains (input.K.scale / input.K.zero) come from a calibration the reference's
test.py produced (an ``output_pt`` directory) or from a bundle/fixture file.

    python sim.py --mflag 5 --ckpt x4sesr.pth --calib output_pt --input rand_SR_Input_80x960.pt
    python sim.py --mflag 5 --params tests/golden/sesr_x4.params.npz --input tests/golden/rand_SR_Input_80x960.npy
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

import define
from define import QUAN_BIT, PE, BIAS_BIT, PE_ACC_BIT, PE_ADD_BIT, REQUAN_BIT, REQUAN_N_MAX
from myQL.quan_func import (quantize_model_weight, quantize_asymmetrical_by_tensor, reshape_input_for_hardware_pe,
                            PEs_and_bias_adder, requan_conv2d_output)
from myQL.quan_classes import NodeInsertMapping, FunctionPackage, NodeInsertMappingElement
from myQL.graph_modify import insert_before, insert_bias_bypass, insert_after
from models import sesr_sim, nrdm_3_sim, sesr_arch_sim, nrdm_6
from models import quantize_utils_pt as quantize
from sesrq.store import STORE

# MFLAG -> net, as the reference's test_float.py:25-48 numbers them.  4 (nrdm_6, 8 convs) has no integer path in the
# reference: roles generalise by position here, parity unpinned.
MODELS = {3: nrdm_3_sim.nr, 4: nrdm_6.nr, 5: sesr_sim.sesr, 6: sesr_arch_sim.sesr}


def float_model(mflag, ckpt=None, params=None):
    """Float net, collapsed.  ckpt: a reference state_dict (.pth, loaded with weights_only=True);
    params: an .npz holding already-collapsed convs Wf{k}/bf{k} (tests/golden/*.params.npz)."""
    if mflag not in MODELS:
        raise ValueError(f"MFLAG {mflag}: only 3 (nrdm_3), 4 (nrdm_6), 5 (SESR x4) and 6 (SESR x2) have an integer path")
    model = MODELS[mflag]()
    model.train()
    if ckpt is not None:
        load_checkpoint(model, ckpt, mflag)
    model.collapse()
    if params is not None:
        z = np.load(params, allow_pickle=False)
        convs = [model.conv_first.conv_expand] + [b.conv_expand for b in model.residual_block] + [model.conv_last.conv_expand]
        with torch.no_grad():
            for k, c in enumerate(convs):
                c.weight.copy_(torch.from_numpy(z[f"Wf{k}"]))
                c.bias.copy_(torch.from_numpy(z[f"bf{k}"]))
        meta = json.loads(str(z["meta"]))
        if "scale" in meta:
            STORE.set_activation_domains(meta["scale"], meta["zero"])
    return model


QAT_SKIP_ADDS = ("add_residual.", "add_upsampled_input.")


def load_checkpoint(model, ckpt, mflag):
    """Load a reference checkpoint into the collapsible net -- strictly.

    A ``*_qat_G.pth`` (it carries ``*_quantizer.*`` entries) is only meaningful after ``quantize.prepare()`` has wrapped
    every conv in a fake-quantising conv (reference sim.py:64-66): ``collapse()`` then folds THROUGH the quantisers, and
    the folded weights differ from a fold of the raw conv weights by tens to hundreds of INT8 weights per net.  The
    reference selects this with its ``qatf`` switch; here the checkpoint itself decides (models/quantize_utils_pt.py).
    Entries of the two long-skip adds' quantisers (``add_residual.*``, ``add_upsampled_input.*``: QuantAdd state) are
    dropped: the integer path merges the skip in the integer domain and never evaluates them.  Any other key mismatch (a checkpoint of another net / --mflag) is refused -- nothing may leave
    random-init weights behind."""
    sd = torch.load(ckpt, weights_only=True, map_location="cpu")
    if isinstance(sd, dict) and "state_dict" in sd:
        sd = sd["state_dict"]
    if not isinstance(sd, dict):
        raise ValueError(f"{ckpt}: not a state_dict")
    if quantize.is_qat_state_dict(sd):
        quantize.prepare(model, inplace=True, a_bits=QUAN_BIT, w_bits=QUAN_BIT, q_type=0, q_level="C")
        sd = {k: v for k, v in sd.items() if not k.startswith(QAT_SKIP_ADDS)}
    try:
        res = model.load_state_dict(sd, strict=False)
    except RuntimeError as e:           # tensor shape mismatch
        raise ValueError(f"{ckpt} does not fit the MFLAG {mflag} net ({type(model).__module__}): {e}") from None
    if res.missing_keys or res.unexpected_keys:
        raise ValueError(f"{ckpt} does not fit the MFLAG {mflag} net ({type(model).__module__}): missing "
                         f"{res.missing_keys[:3]}{'...' if len(res.missing_keys) > 3 else ''}, unexpected "
                         f"{res.unexpected_keys[:3]}{'...' if len(res.unexpected_keys) > 3 else ''}")


def splice(model, qmode=1):
    """The four graph rewrites of the reference, in the reference's order (sim.py:85-114)."""
    model = quantize_model_weight(model, QUAN_BIT, qmode)

    def one(fn, kw):
        m = NodeInsertMapping()
        m.add_config(NodeInsertMappingElement(torch.nn.Conv2d, FunctionPackage(fn, kw)))
        return m
    model = insert_before(model_input=model, insert_mapping=one(quantize_asymmetrical_by_tensor, {"width": QUAN_BIT, "exe_mode": qmode}),
                          has_func_id=True)
    model = insert_before(model_input=model, insert_mapping=one(reshape_input_for_hardware_pe, {"pe_num": PE}))
    model = insert_after(model_input=model, insert_mapping=one(requan_conv2d_output, {"exe_mode": qmode}))
    model = insert_bias_bypass(model_input=model, insert_mapping=one(
        PEs_and_bias_adder, {"pe_add_width": PE_ADD_BIT, "pe_acc_width": PE_ACC_BIT, "bias_width": BIAS_BIT, "pe_num": PE,
                             "exe_mode": qmode}))
    return model


def banner(mflag):
    print("SIM_mflag:", mflag)
    print("QUAN_BIT:", QUAN_BIT)
    print("BIAS_BIT:", BIAS_BIT)
    print("PE_ACC_BIT:", PE_ACC_BIT)
    print("PE_ADD_BIT:", PE_ADD_BIT)
    print("REQUAN_BIT:", REQUAN_BIT)
    print("REQUAN_N_MAX:", REQUAN_N_MAX)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--mflag", type=int, default=define.MFLAG)
    ap.add_argument("--ckpt", help="reference float checkpoint (.pth state_dict)")
    ap.add_argument("--params", help=".npz with collapsed float convs (+ calibrated domains)")
    ap.add_argument("--calib", help="output_pt directory written by the reference's test.py")
    ap.add_argument("--input", required=True, help="frame tensor: .pt (torch) or .npy, shape (N,C,H,W) float32")
    ap.add_argument("--save", help="write the float result here (.npy)")
    ap.add_argument("--dump", help="write the parameter store as an output_pt-compatible tree here (what the define.py *_W_FLG "
                                   "switches select, plus weights and activation domains); all dump switches are turned on")
    args = ap.parse_args(argv)
    define.check()
    if args.dump:
        for n in ("WEIGHT_W_FLG", "INPUT_W_FLG", "BIAS_W_FLG", "BIAS_QUAN_W_FLG", "OUTPUT_PE_W_FLG", "OUTPUT_PE_ADD_W_FLG", "REQUAN_FACTOR_W_FLG"):
            setattr(define, n, True)
    if args.calib:
        STORE.load_output_pt(args.calib)
    model = splice(float_model(args.mflag, args.ckpt, args.params))
    inps = torch.load(args.input, weights_only=True, map_location="cpu") if args.input.endswith(".pt") else \
        torch.from_numpy(np.load(args.input))
    if not torch.cuda.is_available():
        raise SystemExit("sim.py: the integer path needs a HIP device (no CPU fallback)")
    gfake = model(inps.float().cuda())
    torch.cuda.synchronize()
    banner(args.mflag)
    print("output:", tuple(gfake.shape), "engines:", model._sesrq_engine(gfake.device).layer_engines())
    if args.save:
        np.save(args.save, gfake.cpu().numpy())
    if args.dump:
        STORE.save_output_pt(args.dump)
        print("dumped:", args.dump)
    return gfake


if __name__ == "__main__":
    main()
