"""Ext-valued FRI fold (fri_fold_ext, src/math/fri.rs:7-25) on large layers, device-resident: 24 B per input Ext element (16 read + 8 written)."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
import toyni_amd
P = 2013265921
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
def t(fn, reps):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for lg in (25, 22):
    m = 1 << lg                                  # Ext elements
    ctx = toyni_amd.NttContext(m)
    e = torch.randint(0, P, (4 * m,), dtype=torch.int32, device=dev)
    o = torch.empty(2 * m, dtype=torch.int32, device=dev)
    ms = t(lambda: toyni_amd.fri_fold_ext_device(ctx, e.data_ptr(), o.data_ptr(), m, [1, 2, 3, 4], 7, stream=stream), 20)
    print(f"fold_ext m=2^{lg} Ext elements: {ms*1e3:.1f} us  {24.0*m/ms/1e6:.0f} GB/s")
