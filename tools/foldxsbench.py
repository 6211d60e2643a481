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
for lg in [int(v) for v in os.environ.get("XS_LOGS", "24,27,20").split(",")]:
    m = 1 << lg
    e = torch.randint(0, P, (m,), dtype=torch.int32, device=dev)
    xs = torch.randint(1, P, (m // 2,), dtype=torch.int32, device=dev)
    o = torch.empty(m // 2, dtype=torch.int32, device=dev)
    ms = t(lambda: toyni_amd.fri_fold_xs_device(e.data_ptr(), xs.data_ptr(), o.data_ptr(), m, 123456789, stream=stream), 20)
    print(f"fold_xs m=2^{lg}: {ms*1e3:.1f} us  {8.0*m/ms/1e6:.0f} GB/s (8 B per input element)")
