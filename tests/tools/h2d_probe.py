#!/usr/bin/env python3
"""What the link gives on this box: pinned host -> device and device -> pinned host copies of several sizes through torch (hipMemcpyAsync on one
stream), best of a few — the ceiling the enter -> return rows of bench.py (value_wall: 74 MB of tables in, 5 MB of results out per call) are read against."""
import json
import time

import torch


def main():
    dev = torch.device("cuda:0")
    out = {}
    for mb in (4, 16, 64, 256):
        n = mb << 20
        h = torch.empty(n, dtype=torch.uint8).pin_memory()
        d = torch.empty(n, dtype=torch.uint8, device=dev)
        for name, (src, dst) in (("h2d", (h, d)), ("d2h", (d, h))):
            best = 1e9
            for _ in range(8):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                dst.copy_(src, non_blocking=True)
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            out[f"{name}_{mb}MiB_GBps"] = round(n / best / 1e9, 2)
    # four copies in flight on four streams (what a streamed call does with its parts)
    n = 16 << 20
    hs = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(4)]
    ds = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(4)]
    ss = [torch.cuda.Stream() for _ in range(4)]
    best = 1e9
    for _ in range(8):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for h, d, s in zip(hs, ds, ss):
            with torch.cuda.stream(s):
                d.copy_(h, non_blocking=True)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    out["h2d_4x16MiB_on_4_streams_GBps"] = round(4 * n / best / 1e9, 2)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
