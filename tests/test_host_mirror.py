"""Host-side mirror of the reference interface (define / myQL / models / sim): the spliced graph
lowers to the integer bundle the reference itself produced (CPU-only: no device call), keeps the
reference's error conventions, and -- on the GPU -- the spliced model's forward is the golden output."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_fixture
import define
import sim
from myQL import quan_func as qf
from myQL.quan_classes import NodeInsertMapping, FunctionPackage, NodeInsertMappingElement
from myQL.graph_modify import insert_before
from models import sesr_sim, model_utils_pt
from sesrq.store import STORE

CASES = {"sesr_x4": 5, "nrdm_3": 3, "sesr_x2_rand": 6, "sesr_x4_qat": 5, "nrdm_3_qat": 3}


@pytest.mark.parametrize("case", sorted(CASES))
def test_spliced_graph_lowers_to_reference_bundle(case):
    STORE.clear()
    model = sim.splice(sim.float_model(CASES[case], params=os.path.join(GOLDEN, f"{case}.params.npz")))
    names = [getattr(n.target, "__name__", str(n.target)) for n in model.graph.nodes]
    # same chain the reference's generated forward shows
    i = names.index("conv_first.conv_expand")
    assert names[i - 2:i + 3] == ["quantize_asymmetrical_by_tensor", "reshape_input_for_hardware_pe",
                                  "conv_first.conv_expand", "PEs_and_bias_adder", "requan_conv2d_output"]
    b = model.sesrq_bundle()
    fx, meta = load_fixture(os.path.join(GOLDEN, f"{case}.crop.npz"))
    for k in range(5):
        np.testing.assert_array_equal(b.layers[k].wq, fx[f"Wq{k}"])
        np.testing.assert_array_equal(b.layers[k].add_const, fx[f"add_const{k}"])
        assert (b.layers[k].M, b.layers[k].n) == (meta["M"][k], meta["n"][k])
    assert (b.M_res, b.n_res) == (meta["M_res"], meta["n_res"])
    assert b.pixel_shuffle == {5: 4, 3: 1, 6: 2}[CASES[case]]
    assert b.zero == meta["zero"] and b.scale == meta["scale"]
    # conv biases were zeroed by the bias bypass and travel as kwargs (graph_modify.py:68-120)
    assert float(model.conv_first.conv_expand.bias.abs().max()) == 0.0


def test_collapse_is_the_same_linear_map():
    torch.manual_seed(0)
    net = sesr_sim.sesr()
    x = torch.rand(1, 1, 12, 14)
    with torch.no_grad():
        init = net.conv_first(x)
        mid = net.residual_block(init)
        want = net.depth_to_space(net.conv_last(mid))
        net.collapse()
        got = net(x)
    assert got.shape == (1, 1, 48, 56)
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-4, atol=1e-5)
    assert isinstance(net.conv_first.conv_squeeze, torch.nn.Identity) and net.conv_first.conv_expand.weight.shape == (16, 1, 5, 5)


@pytest.mark.skipif(not os.path.isfile("/root/reference/model_params/x4sesr.pth"), reason="reference checkpoint not present")
def test_collapse_of_reference_checkpoint_matches_golden_float_weights():
    """Build-container only: my closed-form fold vs the reference's delta-image fold on x4sesr.pth."""
    m = sim.float_model(5, ckpt="/root/reference/model_params/x4sesr.pth")
    z = np.load(os.path.join(GOLDEN, "sesr_x4.params.npz"))
    convs = [m.conv_first.conv_expand] + [b.conv_expand for b in m.residual_block] + [m.conv_last.conv_expand]
    for k, c in enumerate(convs):
        np.testing.assert_allclose(c.weight.detach().numpy(), z[f"Wf{k}"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(c.bias.detach().numpy(), z[f"bf{k}"], rtol=0, atol=1e-7)


def test_error_conventions_of_the_callables():
    with pytest.raises(AssertionError, match="less than shift_max"):
        qf.quan_layer_between_const(0.5, 16, 16)
    assert qf.quan_layer_between_const(0.0078125) == (32768, 22)
    with pytest.raises(AssertionError, match="all zero"):
        qf.quantize_symmetrical_by_tensor(torch.zeros(2, 2, 3, 3), 8, 1, func_id=0)
    with pytest.raises(AssertionError, match="dimension: 4"):
        qf.reshape_input_for_hardware_pe(torch.zeros(3, 4, 5))
    with pytest.raises(RuntimeError, match="stage marker"):
        qf.requan_conv2d_output(torch.zeros(1, 1, 2, 2), func_id=0, exe_mode=1)
    with pytest.raises(RuntimeError, match="stage marker"):
        qf.quantize_asymmetrical_by_tensor(torch.zeros(1, 1, 2, 2), width=8, exe_mode=0, func_id=0)
    assert qf.float_to_hex(-1, 8) == "ff" and qf.float_to_hex(127, 8) == "7f" and qf.float_to_hex(-131072, 18) == "20000"


def test_partial_splice_is_rejected():
    STORE.clear()
    model = qf.quantize_model_weight(sim.float_model(5, params=os.path.join(GOLDEN, "sesr_x4.params.npz")), 8, 1)
    m = NodeInsertMapping()
    m.add_config(NodeInsertMappingElement(torch.nn.Conv2d, FunctionPackage(qf.quantize_asymmetrical_by_tensor, {"width": 8, "exe_mode": 1})))
    gm = insert_before(model_input=model, insert_mapping=m, has_func_id=True)
    with pytest.raises(RuntimeError, match="reshape_input_for_hardware_pe"):
        gm.sesrq_bundle()
    with pytest.raises(RuntimeError, match="GPU only"):
        gm(torch.zeros(1, 1, 8, 8))


def test_missing_calibration_is_reported():
    STORE.clear()
    model = sim.splice(sim.float_model(5, params=os.path.join(GOLDEN, "sesr_x4.params.npz")))
    STORE.clear()
    with pytest.raises(KeyError, match="parameter store"):
        model.sesrq_bundle()


def test_define_surface():
    for name in ("MFLAG", "PE", "QUAN_BIT", "BIAS_BIT", "PE_ACC_BIT", "PE_ADD_BIT", "REQUAN_BIT", "REQUAN_N_MAX",
                 "WEIGHT_W_FLG", "INPUT_W_FLG", "BIAS_W_FLG", "BIAS_QUAN_W_FLG", "OUTPUT_PE_W_FLG", "OUTPUT_PE_ADD_W_FLG",
                 "REQUAN_FACTOR_W_FLG", "WEIGHT_W_HIST_PNG", "INPUT_W_HIST_PNG", "TEST_RAW_ADD_NOISE"):
        assert hasattr(define, name)
    assert (define.PE, define.QUAN_BIT, define.BIAS_BIT, define.PE_ACC_BIT, define.PE_ADD_BIT, define.REQUAN_BIT,
            define.REQUAN_N_MAX) == (4, 8, 16, 18, 20, 16, 32)
    define.check()


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["sesr_x4", "nrdm_3", "sesr_x2_rand"])
def test_spliced_model_forward_is_the_golden_output(case):
    STORE.clear()
    model = sim.splice(sim.float_model(CASES[case], params=os.path.join(GOLDEN, f"{case}.params.npz")))
    fx, meta = load_fixture(os.path.join(GOLDEN, f"{case}.crop.npz"))
    y = model(torch.from_numpy(fx["x"]).cuda())
    np.testing.assert_array_equal(y.cpu().numpy(), fx["out"])
    assert model.last_q.dtype == torch.int8


def test_calibration_graph_is_recognised_on_cpu():
    """mode-0 splice (reference test.py:79-106) is accepted and classified; running it needs the GPU."""
    import importlib
    STORE.clear()
    calib = importlib.import_module("test")            # sesr-pytorch-quantize_amd/test.py, the calibration entry
    from sesrq.lowering import graph_mode
    gm = calib.splice_calibration(sim.float_model(5, params=os.path.join(GOLDEN, "sesr_x4.params.npz")))
    assert graph_mode(gm) == 0
    names = [getattr(n.target, "__name__", str(n.target)) for n in gm.graph.nodes]
    assert names.count("quantize_asymmetrical_by_tensor") == 6          # 5 convs + the one before PixelShuffle
    assert "requan_conv2d_output" not in names
    with pytest.raises(RuntimeError, match="GPU only"):
        gm(torch.zeros(1, 1, 8, 8))


@pytest.mark.gpu
def test_calibrate_then_infer_end_to_end():
    """test.py-equivalent calibration on the reference's random frame, then sim.py-equivalent inference with
    the calibrated domains: the integer output equals the oracle's for the bundle derived from THIS calibration."""
    import importlib
    from oracle import sesrq_oracle as O
    calib = importlib.import_module("test")
    frames = os.path.join(GOLDEN, "rand_SR_Input_80x960.npy")
    scale, zero = calib.main(["--mflag", "5", "--params", os.path.join(GOLDEN, "sesr_x4.params.npz"), "--frames", frames])
    _, pm = load_fixture(os.path.join(GOLDEN, "sesr_x4.params.npz"))
    assert zero == pm["zero"]
    np.testing.assert_allclose(scale, pm["scale"], rtol=2e-4)
    model = sim.splice(sim.float_model(5, params=os.path.join(GOLDEN, "sesr_x4.params.npz")))
    STORE.set_activation_domains(scale, zero)          # float_model() put the reference's domains there; use ours
    b = model.sesrq_bundle()
    x = np.load(frames)[:, :, :32, :64].copy()
    y = model(torch.from_numpy(x).cuda())
    net = O.Net(layers=[O.Layer(l.wq, l.add_const, l.M, l.n, l.relu) for l in b.layers], scale=b.scale, zero=b.zero,
                M_res=b.M_res, n_res=b.n_res, pixel_shuffle=b.pixel_shuffle)
    np.testing.assert_array_equal(y.cpu().numpy(), O.forward(net, x)["y"])


def _plain_state_dict(mflag):
    torch.manual_seed(3)
    return sim.MODELS[mflag]().state_dict()


def test_qat_checkpoint_is_refused_loudly(tmp_path):
    """A *_qat_G.pth carries weight_quantizer / activation_quantizer state that the reference consumes through
    quantize.prepare() (reference sim.py:64-66); folding its raw conv weights would give another INT8 bundle
    (45..1129 differing weights).  float_model must refuse it and point to the golden bundles."""
    sd = _plain_state_dict(5)
    sd["conv_first.conv_expand.weight_quantizer.scale"] = torch.ones(1)
    sd["conv_first.conv_expand.activation_quantizer.observer.min_val"] = torch.zeros(1)
    p = tmp_path / "fake_qat_G.pth"
    torch.save(sd, p)
    with pytest.raises(ValueError, match=r"QAT checkpoint.*--params"):
        sim.float_model(5, ckpt=str(p))
    # the calibration entry goes through the same loader
    import test as calib_entry
    with pytest.raises(ValueError, match="QAT checkpoint"):
        calib_entry.sim.float_model(5, ckpt=str(p))


def test_checkpoint_of_another_net_is_refused(tmp_path):
    p = tmp_path / "nrdm3.pth"
    torch.save(_plain_state_dict(3), p)
    with pytest.raises(ValueError, match="does not fit the MFLAG 5 net"):
        sim.float_model(5, ckpt=str(p))               # 3-channel first conv into the 1-channel x4 net
    p6 = tmp_path / "nrdm6.pth"
    torch.save(_plain_state_dict(4), p6)
    with pytest.raises(ValueError, match="unexpected"):
        sim.float_model(3, ckpt=str(p6))              # 6 residual blocks into the 3-block net
    sim.float_model(3, ckpt=str(p))                   # the right one loads


REF_PARAMS = "/root/reference/model_params"


@pytest.mark.skipif(not os.path.isdir(REF_PARAMS), reason="reference checkpoints not present (build container only)")
@pytest.mark.parametrize("name", ["sr_qat_G.pth", "nrdm_3_qat_G.pth", "nrdm_6_qat_G.pth"])
def test_reference_qat_checkpoints_are_refused(name):
    mflag = {"sr_qat_G.pth": 5, "nrdm_3_qat_G.pth": 3, "nrdm_6_qat_G.pth": 4}[name]
    with pytest.raises(ValueError, match="QAT checkpoint"):
        sim.float_model(mflag, ckpt=os.path.join(REF_PARAMS, name))


@pytest.mark.skipif(not os.path.isdir(REF_PARAMS), reason="reference checkpoints not present (build container only)")
@pytest.mark.parametrize("ckpt,mflag,case", [("x4sesr.pth", 5, "sesr_x4"), ("nrdm_3_raw_G.pth", 3, "nrdm_3")])
def test_reference_float_checkpoint_gives_the_golden_int8_weights(ckpt, mflag, case):
    """--ckpt path end to end on the host: load (strict) -> closed-form collapse -> weight quantiser == the Wq the
    reference produced from the same checkpoint (0 mismatching INT8 weights)."""
    STORE.clear()
    fx, meta = load_fixture(os.path.join(GOLDEN, f"{case}.crop.npz"))
    STORE.set_activation_domains(meta["scale"], meta["zero"])
    model = sim.splice(sim.float_model(mflag, ckpt=os.path.join(REF_PARAMS, ckpt)))
    b = model.sesrq_bundle()
    for k in range(5):
        np.testing.assert_array_equal(b.layers[k].wq, fx[f"Wq{k}"])


def test_registered_torch_op_schema_and_fake_kernel():
    """torch.ops.sesrq.forward is a registered operator: schema, shape inference through the fake kernel, one op
    node in the lowered graph; no CPU kernel (the product path never falls back)."""
    from sesrq import torch_op

    class FakeEngine:                       # only what the fake kernel and the graph builder touch
        def out_shape(self, N, H, W):
            return (N, 3, 2 * H, 2 * W)
    eng = FakeEngine()
    eid = torch_op.register_engine(eng)
    assert str(torch.ops.sesrq.forward.default._schema) == "sesrq::forward(Tensor x, SymInt engine_id) -> (Tensor, Tensor)" or \
        "sesrq::forward(Tensor x, int engine_id) -> (Tensor, Tensor)" in str(torch.ops.sesrq.forward.default._schema)
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        q, y = torch.ops.sesrq.forward(torch.empty(2, 3, 10, 12), eid)
        assert tuple(q.shape) == (2, 3, 20, 24) and q.dtype == torch.int8 and y.dtype == torch.float32
    gm = torch_op.lowered_module(eng)
    ops = [n for n in gm.graph.nodes if n.op == "call_function" and n.target is torch.ops.sesrq.forward.default]
    assert len(ops) == 1 and len([n for n in gm.graph.nodes if n.op == "call_function"]) == 2      # the op + getitem
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.sesrq.forward(torch.zeros(1, 3, 4, 4), eid)                 # CPU tensor: no kernel
    with pytest.raises(RuntimeError, match="not registered"):
        torch_op._engine(10 ** 9)


@pytest.mark.gpu
def test_lowered_one_op_graph_is_the_golden_output():
    STORE.clear()
    model = sim.splice(sim.float_model(5, params=os.path.join(GOLDEN, "sesr_x4.params.npz")))
    fx, meta = load_fixture(os.path.join(GOLDEN, "sesr_x4.crop.npz"))
    x = torch.from_numpy(fx["x"]).cuda()
    gm = model.sesrq_lowered(x.device)
    assert [n.target for n in gm.graph.nodes if n.op == "call_function"][0] is torch.ops.sesrq.forward.default
    np.testing.assert_array_equal(gm(x).cpu().numpy(), fx["out"])
    np.testing.assert_array_equal(model(x).cpu().numpy(), fx["out"])
