"""ParamStore: in-memory replacement of the reference's CWD-relative ``output_pt/`` tree
(SURVEY App. B).  Keys are the reference's relative file names without ``.pt``
("input/input.0.scale", "weight/conv.weight.2", ...).  A directory written by the reference's own
test.py can be imported (scalar/tensor .pt files, loaded with weights_only=True)."""
from __future__ import annotations

import os
from typing import Any, Dict, Iterable


class ParamStore:
    def __init__(self):
        self._d: Dict[str, Any] = {}

    def __contains__(self, key: str) -> bool:
        return key in self._d

    def __getitem__(self, key: str):
        if key not in self._d:
            raise KeyError(f"parameter store has no '{key}' (calibrate first, or load an output_pt directory)")
        return self._d[key]

    def __setitem__(self, key: str, value) -> None:
        self._d[key] = value

    def keys(self) -> Iterable[str]:
        return self._d.keys()

    def clear(self) -> None:
        self._d.clear()

    # ---- calibration results -------------------------------------------------------------
    def set_activation_domains(self, scale, zero) -> None:
        """scale[k], zero[k] of the input of conv k (k = 0..L-1) and of the output domain (k = L)."""
        for k, (s, z) in enumerate(zip(scale, zero)):
            self._d[f"input/input.{k}.scale"] = float(s)
            self._d[f"input/input.{k}.zero"] = int(z)

    def activation_domains(self, L: int):
        return ([self[f"input/input.{k}.scale"] for k in range(L + 1)],
                [self[f"input/input.{k}.zero"] for k in range(L + 1)])

    # ---- compatibility with directories written by the reference ---------------------------
    def load_output_pt(self, root: str = "output_pt") -> int:
        import torch
        n = 0
        for sub in ("input", "weight", "bias", "requan_factor", "residual", "pe_out", "pe_add"):
            d = os.path.join(root, sub)
            if not os.path.isdir(d):
                continue
            for f in sorted(os.listdir(d)):
                if f.endswith(".pt"):
                    self._d[f"{sub}/{f[:-3]}"] = torch.load(os.path.join(d, f), weights_only=True, map_location="cpu")
                    n += 1
        return n

    def save_output_pt(self, root: str = "output_pt") -> None:
        import torch
        for key, val in self._d.items():
            path = os.path.join(root, key + ".pt")
            os.makedirs(os.path.dirname(path), exist_ok=True)
            torch.save(val, path)


STORE = ParamStore()      # the process-wide store the myQL callables read and write
