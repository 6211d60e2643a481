#!/usr/bin/env python3
"""Per-round wall time of the FRI fold + commit rounds 2^21 -> 2^4 (diagnostic): where do the 2 ms of the phase go?
Prints, per round: leaves of the folded layer, us for fold+commit enqueue + stream drain, us for the 32-byte root read-back."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import toyni_amd  # noqa: E402
from toyni_amd._lib import lib  # noqa: E402

P = 2013265921


def main():
    dev = torch.device("cuda", 0)
    n = 1 << 21
    ctx = toyni_amd.NttContext(n)
    stream = torch.cuda.current_stream().cuda_stream
    lay = [torch.randint(0, P, (n >> k,), dtype=torch.int32, device=dev) for k in range(18)]
    lv = [torch.empty((lib.toyni_merkle_total_digests(n >> k), 32), dtype=torch.uint8, device=dev) for k in range(18)]
    salts = torch.randint(0, 255, (n, 16), dtype=torch.uint8, device=dev)
    reps = 20
    tot = 0.0
    for k in range(17):
        m = n >> k
        t_run = t_root = 0.0
        for r in range(reps + 2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            toyni_amd.prover.fri_fold_commit_device(ctx, lay[k].data_ptr(), lay[k + 1].data_ptr(), m, 12345, 7, salts.data_ptr(), lv[k + 1].data_ptr(), stream=stream)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            root = lv[k + 1][-1].cpu()
            t2 = time.perf_counter()
            if r >= 2:
                t_run += t1 - t0
                t_root += t2 - t1
        print(f"round {k:2d}: {m // 2:8d} leaves  fold+commit {t_run / reps * 1e6:8.1f} us   root read-back {t_root / reps * 1e6:6.1f} us", flush=True)
        tot += (t_run + t_root) / reps
    print(f"sum {tot * 1e3:.3f} ms")


if __name__ == "__main__":
    main()
