"""Configuration surface of the INT8 integer path -- same names and meanings as the reference's
``define.py`` (reference define.py:1-36), so that a ``sim.py``-shaped script keeps working.

Differences by design: values are validated once (``check()``), and the dump switches route through
the device engine's debug taps (sesrq_forward_debug): with a ``*_W_FLG`` on, every forward of the
spliced model also leaves the tensors the reference would have written under ``./output_pt/`` in the
parameter store under the reference's file names (sesrq/lowering.py:_dump_taps), and
``STORE.save_output_pt(dir)`` -- ``sim.py --dump dir`` -- writes the tree.  Nothing is written to the
working directory unless a caller asks for it.
"""

# which network the entry script builds: 3 = nrdm_3 (3->3 ch), 5 = SESR x4 (1->1 ch, PixelShuffle 4),
# 6 = SESR x2 (3->3 ch, PixelShuffle 2)
MFLAG = 3
TEST_RAW_ADD_NOISE = False

# hardware model: four processing elements, channel c is handled by PE c % 4
PE = 4

QUAN_BIT = 8          # activations and weights
BIAS_BIT = 16         # bias constant  clamp16(bias_q - zero * sum(W))
PE_ACC_BIT = 18       # each PE's accumulator saturates here
PE_ADD_BIT = 20       # the 4-input adder tree saturates here

REQUAN_BIT = 16       # requant multiplier M < 2**16
REQUAN_N_MAX = 32     # requant shift n <= 32

# debug taps (reference: "write this intermediate to output_pt/ / output_txt/"); read at every forward, so a script may
# switch them after import (define.INPUT_W_FLG = True)
w_flg_c = False
WEIGHT_W_FLG = w_flg_c and True
INPUT_W_FLG = w_flg_c and True
BIAS_W_FLG = w_flg_c and True
BIAS_QUAN_W_FLG = w_flg_c and True
OUTPUT_PE_W_FLG = w_flg_c and True
OUTPUT_PE_ADD_W_FLG = w_flg_c and True
REQUAN_FACTOR_W_FLG = w_flg_c and True

# NOT in the reference's define.py: one weight scale per OUTPUT channel instead of the reference's one per tensor (myQL/quan_func.py:58-71).
# BASELINE's north star names per-channel weights; nothing of the reference pins them (parity unpinned) and such layers run on the dot4
# kernels.  Read by quantize_model_weight at call time; the default is the reference's behaviour.
WEIGHT_PER_CHANNEL = False

WEIGHT_W_HIST_PNG = False
INPUT_W_HIST_PNG = False


def check():
    """Reject configurations the device engine cannot honour (mirrors sesrq_create's checks)."""
    if PE != 4:
        raise ValueError("only PE == 4 is supported")
    if QUAN_BIT != 8:
        raise ValueError("only QUAN_BIT == 8 is supported")
    if not (REQUAN_BIT < REQUAN_N_MAX <= 32) or REQUAN_BIT > 16:
        raise ValueError("need REQUAN_BIT <= 16 < REQUAN_N_MAX <= 32")
    if not (8 < PE_ACC_BIT <= PE_ADD_BIT < 32):
        raise ValueError("need 8 < PE_ACC_BIT <= PE_ADD_BIT < 32")
    if not (2 <= BIAS_BIT <= 24):
        raise ValueError("BIAS_BIT out of range")
