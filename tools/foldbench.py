import os, sys, torch
sys.path.insert(0, os.getcwd())
import toyni_amd
P = 2013265921
dev = torch.device("cuda", 0)
c27 = toyni_amd.NttContext(1 << 27)
big = torch.randint(0, P, (1 << 28,), dtype=torch.int32, device=dev)
o2 = torch.empty(1 << 26, dtype=torch.int32, device=dev)
stream = torch.cuda.current_stream().cuda_stream
def t(fn, reps):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for m in [1 << int(v) for v in os.environ.get("FOLD_LOGS", "27,26").split(",")]:
    ms = t(lambda: toyni_amd.fri_fold_device(c27, big.data_ptr(), o2.data_ptr(), m, 123456789, 7, stream=stream), 30)
    print(f"fold m=2^{m.bit_length()-1}: {ms*1e3:.1f} us  {6.0*m/ms/1e6:.0f} GB/s")
