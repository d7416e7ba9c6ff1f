"""Generates tests/golden/reference_sh4.npz (run ONCE in the build container; never on the GPU box): the reference's python
SH evaluation at degree 4 (utils/sh_utils.py:57-112, the `deg > 3` branch at :102-112) on seeded float64 inputs.  Only data
(inputs + expected outputs) is written.

  python tests/golden/make_reference_fixtures_sh4.py        # needs /root/reference
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_sh4.npz")


def main():
    sys.path.insert(0, REF)
    from utils.sh_utils import eval_sh                       # noqa: E402
    gen = torch.Generator().manual_seed(20261004)
    P = 129
    sh = torch.randn(P, 25, 3, generator=gen, dtype=torch.float64)
    dirs = torch.randn(P, 3, generator=gen, dtype=torch.float64)
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    out = {"sh_coeffs": sh.numpy(), "dirs": dirs.numpy()}
    for deg in (3, 4):
        out[f"sh_raw_deg{deg}"] = eval_sh(deg, sh.transpose(1, 2), dirs).numpy()      # reference layout [P, 3, K]
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
