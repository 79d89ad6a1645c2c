"""Companion of tools/dump_fma_probe.py: compare the CUDA reference's indices (out.npz) with the oracle
in both arithmetic modes and say which one nvcc's build corresponds to.   python tools/check_fma_probe.py out.npz"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle_ext  # noqa: E402


def main():
    g = np.load(sys.argv[1])
    oracle_ext.build()
    for mode in (0, 1):
        oracle_ext.set_fma_mode(mode)
        ok = []
        for i in range(3):
            xyz = torch.from_numpy(g[f"xyz{i}"])
            m = g[f"fps{i}"].shape[1]
            idx = oracle_ext.furthest_point_sampling(xyz, m)
            new = torch.from_numpy(np.take_along_axis(g[f"xyz{i}"], g[f"fps{i}"][..., None].astype(np.int64), 1))
            bq = oracle_ext.ball_query(new, xyz, 0.2, 32)
            ok.append((bool((idx.numpy() == g[f"fps{i}"]).all()), bool((bq.numpy() == g[f"bq{i}"]).all())))
        print(f"mode {mode}: (fps identical, ball_query identical) per cloud = {ok}")
    oracle_ext.set_fma_mode(0)


if __name__ == "__main__":
    main()
