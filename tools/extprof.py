#!/usr/bin/env python3
"""Per-kernel durations of base and interleaved (Ext) transforms of the same bytes, for `rocprofv3 --kernel-trace --stats`:
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/extprof -- python3 tools/extprof.py
The interleaved kernels carry LQ = 2 as the last template argument of their Pass<...> symbol."""
import os
import sys

import torch

sys.path.insert(0, os.getcwd())
import toyni_amd

P = 2013265921
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
for log_n, vecs in [(int(a), int(b)) for a, b in (s.split(":") for s in os.environ.get("EXTPROF", "12:16384,16:1024,20:64").split(","))]:
    n = 1 << log_n
    ctx = toyni_amd.NttContext(n)
    x = torch.randint(0, P, (vecs * n * 4,), dtype=torch.int32, device=dev)
    p = x.data_ptr()
    for _ in range(6):
        ctx.run_device(p, p, 4 * vecs, False, stream=stream, shift=7)
        ctx.run_device_ext_batch(p, p, vecs, False, shift=7, stream=stream)
        ctx.run_device(p, p, 4 * vecs, True, stream=stream, shift=7)
        ctx.run_device_ext_batch(p, p, vecs, True, shift=7, stream=stream)
    torch.cuda.synchronize()
    ctx.destroy()
    del x
print("done")
