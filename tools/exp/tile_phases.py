#!/usr/bin/env python3
"""Where a tile's time goes in kth_tile_kernel: the KT_CLOCK build (tools/exp/build_variants.sh ktclock "-DKT_CLOCK"
kth_axis0.hip) stamps the shader clock at every phase boundary of a workgroup's first 64 tiles.
    PROBE_SO=tools/exp/var/libcp_pre_hip.ktclock.so python tools/exp/tile_phases.py [n ...]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cp_pre_amd import _lib                      # noqa: E402
_lib.SO_PATH = os.path.abspath(os.environ.get("PROBE_SO", os.path.join(ROOT, "tools/exp/var/libcp_pre_hip.ktclock.so")))
from cp_pre_amd import inductive_cp as icp      # noqa: E402

NAMES = ["window+clear", "hist sweep", "narrow+publish", "map set", "collect+prefetch", "pick+store", "cleanup"]


def main():
    dev = torch.device("cuda:0")
    lib = _lib.load()
    alphas = [float(a) for a in icp.ALPHA_LEVELS]
    ns = [int(a) for a in sys.argv[1:]] or [256, 384, 512, 768, 1000, 1500, 2048]
    for n in ns:
        M = (1 << 32) // (4 * n) // 64 * 64 + 64                 # ~4 GB of scores, no power-of-two pitch
        s = torch.randn(n, M, device=dev).abs_()
        ks = [icp.kth_index(n, n, a) for a in alphas]
        for _ in range(2):
            icp.kth_axis0(s, ks)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        icp.kth_axis0(s, ks)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        buf = np.zeros(1024 * 64 * 8, dtype=np.uint64)
        rc = lib.pre_debug_kt_clock(ctypes.c_void_p(buf.ctypes.data), ctypes.c_size_t(buf.nbytes))
        assert rc == 0, rc
        st = buf.reshape(1024, 64, 8)[:, :, :7].astype(np.int64)
        tiles = (M + (31 if n > 1024 else 63)) // (32 if n > 1024 else 64)
        blocks = min(tiles, 512 if n <= 512 else 256)
        per = min(64, tiles // blocks)
        st = st[:blocks, 1:per - 1]                               # steady state: neither the first nor the last tile
        d = np.diff(st, axis=2)                                   # [blocks, tiles, 6] phase durations (clock ticks)
        nxt = st[:, 1:, 0] - st[:, :-1, 6]                        # end of a tile -> top of the next
        tot = st[:, 1:, 0] - st[:, :-1, 0]
        tick_total = tot.mean()
        print(f"n={n} M={M}: {ms:.3f} ms, {4.0 * n * M / ms / 1e6:.0f} GB/s; per tile {tick_total:.0f} ticks "
              f"({ms * 1e3 / (tiles / blocks):.2f} us per tile per workgroup -> {tick_total / (ms * 1e3 / (tiles / blocks)):.1f} ticks/us)")
        for i, name in enumerate(NAMES[:6]):
            print(f"    {name:18s} {d[:, :, i].mean():8.0f} ticks  {100 * d[:, :, i].mean() / tick_total:5.1f} %   (p10 {np.percentile(d[:, :, i], 10):.0f}, p90 {np.percentile(d[:, :, i], 90):.0f})")
        print(f"    {'loop top':18s} {nxt.mean():8.0f} ticks  {100 * nxt.mean() / tick_total:5.1f} %")
        del s
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
