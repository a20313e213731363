"""Hardware stimulus / expected-response writers: the hex text files the reference hands to
the RTL test bench, produced from a parameter bundle and the device engine's debug taps instead of from
files under ./output_pt/.

Formats restated from the reference (byte-for-byte, including its quirks, pinned by
tests/golden/*.stimtxt.npz which were written by the reference's own scripts):
    weight/conv.weight.K.txt                  myQL/quan_func.py:82-111   4oc x 4ic blocks, line count first
    input/input.K.txt            (tiled)      output.py:41-118           32x32 tiles, overlap shrinking per layer
    bias/param_buf.txt                        output.py:121-141          bias16 | requant16 | requant_res16 per channel
    pe_out/pe_outputK_P.txt                   output.py:143-193          32x32 tiles of 18-bit PE sums
    pe_add/pe_add_outputK.txt                 output.py:195-233          32x32 tiles of 20-bit adder sums
    requan_shift_n/requan_shift_n.txt         output.py:235-246
    end-to-end input.0 / input.5              output_end2end.py:37-101   (row index = block + row, as the reference has it)
Host-side format work only; nothing here is on the hot path.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Sequence

import numpy as np

TILE = 32


def float_to_hex(item, bit_width) -> str:
    """two's-complement hex of int(item) in `bit_width` bits; 3..8 digits as needed, otherwise 2 (the
    reference's table, output.py:13-39)."""
    digits = math.ceil(bit_width / 4)
    v = int(item)
    if v < 0:
        v = int(2 ** bit_width) + v
    return format(v, "0{}x".format(digits if 3 <= digits <= 8 else 2))


def _hex_table(bit_width):
    """vectorised float_to_hex for integer arrays."""
    digits = math.ceil(bit_width / 4)
    fmt = "0{}x".format(digits if 3 <= digits <= 8 else 2)
    mod = int(2 ** bit_width)

    def conv(a: np.ndarray) -> List[str]:
        a = np.asarray(a).astype(np.int64).ravel()
        a = np.where(a < 0, a + mod, a)
        return [format(int(v), fmt) for v in a]
    return conv


def weight_txt(wq: np.ndarray, quan_bit: int = 8) -> str:
    oc_r, ic_r, kh, kw = wq.shape
    oc, ic = -(-oc_r // 4) * 4, -(-ic_r // 4) * 4
    t = np.zeros((oc, ic, kh, kw), np.int64)
    t[:oc_r, :ic_r] = wq
    out = ["{:02x}\n".format(int(oc * ic * kh * kw / 16))]
    for bo in range(0, oc, 4):
        for bi in range(0, ic, 4):
            for y in range(kh):
                for x in range(kw):
                    blk = t[bo:bo + 4, bi:bi + 4, y, x]          # [oc_i][ic_i]; written ic_i-major
                    out.append("".join(float_to_hex(blk[o, i], quan_bit) for i in range(4) for o in range(4)) + "\n")
    return "".join(out)


def input_tiles_txt(acts: Sequence[np.ndarray], kernel_sizes: Sequence[int] = (0, 5, 3, 3, 3, 5), quan_bit: int = 8) -> List[str]:
    """acts[k]: (1, C, H, W) int8 = input.k.pt for k = 0..L (the last one is the un-shuffled output).  The tile
    overlap shrinks by k//2 per layer and carries over from layer to layer, exactly as output.py does."""
    hx = _hex_table(quan_bit)
    zero = float_to_hex(0, quan_bit)
    h_ov = w_ov = TILE
    files = []
    for layer_id, a in enumerate(acts):
        _, C, H, W = a.shape
        EH, EW = (H // TILE + 1) * TILE, (W // TILE + 1) * TILE
        ex = np.zeros((C, H, EW), np.int64)
        ex[:, :, :W] = a[0]
        nwb, nhb = EW // TILE, EH // TILE
        h_ov -= kernel_sizes[layer_id] // 2
        w_ov -= kernel_sizes[layer_id] // 2
        out = []
        bh = 0
        for hb in range(nhb):
            bw = 0
            cur_h = h_ov if hb == 0 else TILE
            for wb in range(nwb):
                cur_w = w_ov if wb == 0 else TILE
                if hb == nhb - 1:
                    cur_h = H - bh
                out.append("{:02x}\n".format(int(cur_h)))
                out.append("{:02x}\n".format(int(C)))
                for c in range(C):
                    out.append("{:02x}\n".format(c))
                    for h in range(cur_h):
                        out.append("".join(hx(ex[c, bh + h, bw:bw + cur_w])) + zero * (TILE - cur_w) + "\n")
                bw += cur_w
            bh += cur_h
        files.append("".join(out))
    return files


def param_buf_txt(add_consts: Sequence[np.ndarray], M: Sequence[int], M_res: int, bias_bit: int = 16, requan_bit: int = 16) -> str:
    res = float_to_hex(M_res, requan_bit)
    out = [float_to_hex(len(add_consts), 8) + "\n"]
    for ac, m in zip(add_consts, M):
        out.append(float_to_hex(len(ac), 8) + "\n")
        for v in ac:
            out.append(float_to_hex(v, bias_bit) + float_to_hex(m, requan_bit) + res + "\n")
    return "".join(out)


def pe_tiles_txt(t: np.ndarray, bits: int) -> str:
    """t: (C, H, W) integer sums -> 32x32 tiles (pe_out: 18-bit, pe_add: 20-bit)."""
    C, H, W = t.shape
    EH = H if H % TILE == 0 else (H // TILE + 1) * TILE
    EW = W if W % TILE == 0 else (W // TILE + 1) * TILE
    ex = np.zeros((C, H, EW), np.int64)
    ex[:, :, :W] = t
    hx = _hex_table(bits)
    out = []
    for hb in range(EH // TILE):
        for wb in range(EW // TILE):
            h0, w0 = hb * TILE, wb * TILE
            out.append("{:02x}\n".format(int(H - h0 if hb == EH // TILE - 1 else TILE)))
            out.append("{:02x}\n".format(int(C)))
            for c in range(C):
                out.append("{:02x}\n".format(c))
                for h in range(TILE):
                    out.append("".join(hx(ex[c, h0 + h, w0:w0 + TILE])) + "\n")
                    if h0 + h == H - 1:
                        break
    return "".join(out)


def requan_shift_n_txt(n: Sequence[int], n_res: int, n_max: int = 32) -> str:
    bits = math.log2(n_max)
    return "".join(float_to_hex(v, bits) + "\n" for v in n) + float_to_hex(n_res, bits)


def end2end_txt(a: np.ndarray, quan_bit: int = 8) -> str:
    """output_end2end.py: whole-width rows, 4 values per line.  NOTE the reference indexes the row as
    `block + row` (not `32*block + row`); reproduced as is."""
    _, C, H, W = a.shape
    EH = H if H % TILE == 0 else (H // TILE + 1) * TILE
    ex = np.zeros((C, EH, W), np.int64)
    ex[:, :H] = a[0]
    hx = _hex_table(quan_bit)
    out = []
    for hb in range(EH // TILE):
        out.append("{:02x}\n".format(hb))
        for c in range(C):
            out.append("{:02x}\n".format(c))
            for bh in range(TILE):
                vals = hx(ex[c, hb + bh])
                for i in range(0, W, 4):
                    out.append("".join(vals[i:i + 4]) + "\n")
    return "".join(out)


def write_all(bundle, taps: Dict[str, np.ndarray], root: str = "output_txt", end2end: bool = False) -> List[str]:
    """bundle: sesrq.Bundle; taps: numpy arrays input{k} (1,C,H,W) k=0..L [input{L} = un-shuffled output],
    pe_out{k} (4,OC,H,W), pe_add{k} (1,OC,H,W) as returned by Engine.forward_debug (frame 0).  Writes the
    reference's output_txt/ tree and returns the list of files."""
    L = bundle.L
    written = []

    def put(rel, text):
        path = os.path.join(root, rel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            f.write(text)
        written.append(path)

    for k, l in enumerate(bundle.layers):
        put(f"weight/conv.weight.{k}.txt", weight_txt(l.wq))
    ks = [0] + [int(l.wq.shape[2]) for l in bundle.layers]
    acts = [np.asarray(taps[f"input{k}"]) for k in range(L + 1)]
    if end2end:
        put("input/input.0.txt", end2end_txt(acts[0]))
        put(f"input/input.{L}.txt", end2end_txt(acts[L]))
    else:
        for k, text in enumerate(input_tiles_txt(acts, ks)):
            put(f"input/input.{k}.txt", text)
    put("bias/param_buf.txt", param_buf_txt([l.add_const for l in bundle.layers], [l.M for l in bundle.layers], bundle.M_res))
    for k in range(L):
        if f"pe_out{k}" in taps:
            for p in range(4):
                put(f"pe_out/pe_output{k}_{p}.txt", pe_tiles_txt(np.asarray(taps[f"pe_out{k}"])[p], bundle.pe_acc_bits))
        if f"pe_add{k}" in taps:
            put(f"pe_add/pe_add_output{k}.txt", pe_tiles_txt(np.asarray(taps[f"pe_add{k}"])[0], bundle.pe_add_bits))
    put("requan_shift_n/requan_shift_n.txt", requan_shift_n_txt([l.n for l in bundle.layers], bundle.n_res))
    return written
