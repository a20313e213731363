"""Shared test helpers (oracle <-> product glue).  The oracle is imported here and only here /
in tests: the product package never sees it."""
import json
import os

import numpy as np

from conftest import GOLDEN, load_fixture, fixture_input
from oracle import sesrq_oracle as O
import sesrq
from sesrq.bundle import Bundle, LayerParams


def bundle_from_oracle(net: O.Net) -> Bundle:
    return Bundle(layers=[LayerParams(wq=l.wq, add_const=l.add_const, M=l.M, n=l.n, relu=l.relu, M_oc=getattr(l, 'M_oc', None),
                                      n_oc=getattr(l, 'n_oc', None)) for l in net.layers],
                  scale=list(net.scale), zero=list(net.zero), M_res=net.M_res, n_res=net.n_res,
                  pixel_shuffle=net.pixel_shuffle, pe_num=net.pe, pe_acc_bits=net.acc_bits, pe_add_bits=net.add_bits,
                  name=net.name)


def fixture_case(path):
    fx, meta = load_fixture(path)
    return fx, meta, O.net_from_fixture(fx), fixture_input(fx, meta)


def rand_frame(shape, seed):
    """U[0,1) float32 frames from a seeded CPU generator."""
    import torch
    g = torch.Generator().manual_seed(seed)
    return torch.rand(shape, generator=g, dtype=torch.float32).numpy()
