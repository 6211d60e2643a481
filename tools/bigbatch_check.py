#!/usr/bin/env python3
"""Ad-hoc 64-bit-offset check of the entry points tests/test_gpu_big_batch.py does not size up: Ext batches and low-degree extensions
whose OUTPUT passes 2^32 words.  Spot transforms against the oracle on either side of the wrap points."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
import toyni_amd  # noqa: E402

P = 2013265921
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream


def fill(total, seed):
    data = torch.empty(total, dtype=torch.int32, device=dev)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    piece = 1 << 28
    for off in range(0, total, piece):
        m = min(piece, total - off)
        data[off:off + m] = torch.randint(0, P, (m,), dtype=torch.int32, device=dev, generator=g)
    return data


def u64(t):
    return t.cpu().numpy().view(np.uint32).astype(np.uint64)


# Ext: [vector][element][4]
for log_n, vecs in ((20, 1030), (10, 1050000), (24, 66)):
    n = 1 << log_n
    total = vecs * n * 4
    assert total > 1 << 32
    data = fill(total, 7)
    keep = data.clone()
    ctx = toyni_amd.NttContext(n)
    ctx.run_device_ext_batch(data.data_ptr(), data.data_ptr(), vecs, False, shift=7, stream=stream)
    torch.cuda.synchronize()
    bad = []
    wraps = [(1 << 29) // (4 * n), (1 << 30) // (4 * n), (1 << 31) // (4 * n), (1 << 32) // (4 * n)]
    for v in sorted({0, vecs // 2, vecs - 1, *wraps, *[w - 1 for w in wraps if w]}):
        src = u64(keep[v * 4 * n:(v + 1) * 4 * n]).reshape(n, 4)
        got = data[v * 4 * n:(v + 1) * 4 * n].cpu().numpy().view(np.uint32).reshape(n, 4)
        for q in range(4):
            if not (got[:, q] == oracle.domain_fft(np.ascontiguousarray(src[:, q]), n, 7)).all():
                bad.append((v, q))
    ctx.run_device_ext_batch(data.data_ptr(), data.data_ptr(), vecs, True, shift=7, stream=stream)
    torch.cuda.synchronize()
    same = all(torch.equal(data[o:o + (1 << 28)], keep[o:o + (1 << 28)]) for o in range(0, total, 1 << 28))
    print(f"ext n=2^{log_n} vectors={vecs} ({total * 4 / 2**30:.1f} GiB): oracle mismatches {bad}, round trip {'ok' if same else 'BROKEN'}", flush=True)
    ctx.destroy()
    del data, keep
    torch.cuda.empty_cache()

# LDE: batch vectors of n >> lb coefficients -> n evaluations each
for log_n, lb, batch in ((21, 5, 2100), (16, 2, 65800), (24, 3, 260)):
    n = 1 << log_n
    nin = n >> lb
    assert batch * n > 1 << 32
    coeffs = fill(batch * nin, 8)
    out = torch.empty(batch * n, dtype=torch.int32, device=dev)
    ctx = toyni_amd.NttContext(n)
    ctx.lde_device(coeffs.data_ptr(), out.data_ptr(), batch, lb, 7, stream=stream)
    torch.cuda.synchronize()
    bad = []
    wraps = [(1 << 29) // n, (1 << 30) // n, (1 << 31) // n, (1 << 32) // n]
    for b in sorted({0, batch // 2, batch - 1, *wraps, *[w - 1 for w in wraps if w]}):
        want = oracle.domain_fft(u64(coeffs[b * nin:(b + 1) * nin]), n, 7)
        if not (out[b * n:(b + 1) * n].cpu().numpy().view(np.uint32) == want).all():
            bad.append(b)
    print(f"lde n=2^{log_n} blow-up 2^{lb} batch={batch} ({batch * n * 4 / 2**30:.1f} GiB out): oracle mismatches {bad}", flush=True)
    ctx.destroy()
    del coeffs, out
    torch.cuda.empty_cache()
