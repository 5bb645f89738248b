#!/usr/bin/env python3
"""Run ON THE GPU BOX: what slows the outlier vote's sweep kernel when it shares the chip?  The sweep (one lane per
match list, a chain of dependent loads) is timed alone and beside synthetic kernels that each load ONE resource
(tools/contention/spin.hip: full-rate VALU issue, v_sad_u8 issue, LDS traffic, dependent HBM gathers) and beside a
streaming copy.

  hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/contention/spin.hip -o tools/contention/libspin.so
  python tools/vote_contention.py [--lists 2048] [--lanes 64] [--waves-per-simd 4]
"""
import argparse
import ctypes as C
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lists", type=int, default=2048)
    ap.add_argument("--lanes", type=int, default=64)
    ap.add_argument("--waves-per-simd", type=int, default=4)
    ap.add_argument("--kinds", default="idle,fma,sad,lds,chase,copy")
    args = ap.parse_args()
    import torch
    pkg = entry.load_package()
    ob = entry.load_oracle()
    o = ob.Oracle()
    p = ob.Params.default()
    spin = C.CDLL(os.path.join(ROOT, "tools", "contention", "libspin.so"))
    spin.spin_launch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint, C.c_void_p]
    W, H = 1241, 376
    dims = [W, H, pkg.synth.bytes_per_line(W)]
    base = []
    for k in range(4):
        _, a = o.compute_features(p, pkg.synth.frame(W, H, (5 * k) % 20, k % 20, 8, 1, 1 + k), dims)
        _, b = o.compute_features(p, pkg.synth.frame(W, H, (5 * k + 5) % 20, (k + 1) % 20, 8, 1, 1 + k), dims)
        base.append(o.matching(p, dims, 0, m1p=a, m1c=b))
    lists = [base[i % len(base)] for i in range(args.lists)]
    want = [o.remove_outliers(pm)[0] for pm in base]
    torch.cuda.init()
    side = torch.cuda.Stream()
    scratch = torch.zeros(1 << 20, dtype=torch.int32, device="cuda")
    table = torch.randint(0, 2 ** 31 - 1, (1 << 31,), dtype=torch.int32, device="cuda")  # 8 GiB: one 64-byte sector per chase step
    big_a = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
    big_b = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
    blocks = 256 * args.waves_per_simd  # 256-thread blocks: 4 waves each, one per SIMD of a CU
    stop = threading.Event()

    def competitor(kind):
        n = 0
        with torch.cuda.stream(side):
            while not stop.is_set():
                for _ in range(4):
                    if kind == "fma":
                        spin.spin_launch(0, blocks, 400000, scratch.data_ptr(), 0, side.cuda_stream)
                    elif kind == "sad":
                        spin.spin_launch(1, blocks, 200000, scratch.data_ptr(), 0, side.cuda_stream)
                    elif kind == "lds":
                        spin.spin_launch(2, blocks, 200000, scratch.data_ptr(), 0, side.cuda_stream)
                    elif kind == "chase":
                        spin.spin_launch(3, blocks * 4, 20000, table.data_ptr(), (1 << 27) - 1, side.cuda_stream)
                    elif kind == "copy":
                        big_b.copy_(big_a, non_blocking=True)
                    n += 1
                side.synchronize()
        return n

    print(f"{args.lists} lists of ~{np.mean([len(b) for b in base]):.0f} matches, {args.lanes} lists per wave; competitors at {args.waves_per_simd} waves per SIMD", flush=True)
    for kind in args.kinds.split(","):
        th = None
        stop.clear()
        if kind != "idle":
            t0 = time.perf_counter()
            th = threading.Thread(target=competitor, args=(kind,))
            th.start()
            time.sleep(0.5)
        ms = []
        for _ in range(3):
            got, ntri, m = pkg.remove_outliers_device(lists, lanes_per_wave=args.lanes)
            ms.append(m)
        stop.set()
        if th:
            th.join()
        ok = all(got[i].tobytes() == want[i % len(base)].tobytes() for i in range(args.lists))
        print(f"{kind:6s} sweep {min(ms):8.2f} ms (runs: {' '.join(f'{m:.1f}' for m in ms)})  parity {ok}", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
