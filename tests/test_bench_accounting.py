"""bench.py's roofline bookkeeping on the host: algorithmic bytes per launch and the layer-by-layer yardstick
 for the three reference topologies and the 8-conv net; workload table sanity."""
import importlib.util
import os

import numpy as np

from conftest import ROOT, GOLDEN
from sesrq.bundle import Bundle

spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def test_launch_bytes_match_the_survey_accounting():
    x2 = Bundle.load(os.path.join(GOLDEN, "sesr_x2_rand.crop.npz"))
    # 159 B/px for SESR-x2 with int8 end points (+ 3 * Cin = 9 for the fp32 frame) = 168 layer by layer
    assert bench.layerwise_bytes_per_px(x2, True) == 168 and bench.layerwise_bytes_per_px(x2, False) == 159
    # as launched: first layer, fused trio (in + out: the residual operand IS its input and comes out of the LDS window; NOT its layers'
    # 32 + 32 + 48), last layer
    assert [bench.launch_bytes_per_px(x2, f, c, True) for f, c in ((0, 1), (1, 3), (4, 1))] == [28, 32, 28]
    # one launch per layer (--no-fuse): layer 3 reads the residual operand as a second tensor
    assert [bench.launch_bytes_per_px(x2, k, 1, True) for k in range(5)] == [28, 32, 32, 48, 28]
    # zero[1] != -128: layer 0 writes the residual operand as a tensor of its own and the merging launch reads it besides its input
    import copy
    xz = copy.deepcopy(x2); xz.zero = list(xz.zero); xz.zero[1] = -120
    assert [bench.launch_bytes_per_px(xz, f, c, True) for f, c in ((0, 1), (1, 3), (4, 1))] == [44, 48, 28]
    # fused front: fp32 frame in, one NHWC16 tensor out, the residual operand never leaves the CU
    assert [bench.launch_bytes_per_px(x2, f, c, True) for f, c in ((0, 4), (4, 1))] == [28, 28]
    assert 168 * 1080 * 1920 == 348364800
    x4 = Bundle.load(os.path.join(GOLDEN, "sesr_x4.crop.npz"))
    assert bench.layerwise_bytes_per_px(x4, False) == 161          # SURVEY: SESR-x4 (1 -> 16): 161 B/px
    n3 = Bundle.load(os.path.join(GOLDEN, "nrdm_3.crop.npz"))
    assert bench.layerwise_bytes_per_px(n3, False) == 150          # SURVEY: nrdm_3 (3 -> 3): 150 B/px
    n6 = Bundle.load(os.path.join(GOLDEN, "unpinned", "nrdm_6.bundle.npz"))
    assert bench.layerwise_bytes_per_px(n6, False) == 246          # SURVEY: nrdm_6 (L = 8): 246 B/px
    # two trios: the first moves in + out, the second also reads the residual operand
    assert [bench.launch_bytes_per_px(n6, f, c, False) for f, c in ((0, 1), (1, 3), (4, 3), (7, 1))] == [19, 32, 48, 19]


def test_workload_table_points_at_existing_bundles():
    for name, (fixtures, cin, H, W, (mode, n), desc) in bench.WORKLOADS.items():
        for f in fixtures:
            b = Bundle.load(os.path.join(GOLDEN, f))
            assert b.L >= 5
        assert Bundle.load(os.path.join(GOLDEN, fixtures[0])).in_channels == cin, name
        assert mode in ("per_gpu", "total") and n >= 1
    # config 5: the second net takes the first one's output channels
    f1, f2 = bench.WORKLOADS["nrdm6_sesrx2_540p"][0]
    assert Bundle.load(os.path.join(GOLDEN, f1)).out_channels == Bundle.load(os.path.join(GOLDEN, f2)).in_channels
