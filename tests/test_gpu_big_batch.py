"""Batches beyond 2^32 ELEMENTS (16 GiB of packed residues) in one call: every tile / row / element offset of the pass kernels has to
be 64-bit clean past the points where a u32 element index (2^32) or a u32 byte offset (2^30 elements; 2^29 for a signed one) wraps.
The reference has no batch at all (one transform per call, cuda/ntt_kernel.cu:246-316); a 288 GB device invites exactly this use of
toyni_ntt_device.  One single-pass size, the headline two-pass size and a three-pass size: the transforms on either side of each wrap
point and at both ends against the oracle (src/ntt.rs:14-66), and forward + inverse as the identity over ALL of the data."""
import numpy as np
import pytest
import torch

import oracle
from oracle import P
from test_gpu_parity import ta  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def fill(total, dev, seed):
    data = torch.empty(total, dtype=torch.int32, device=dev)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    piece = 1 << 28
    for off in range(0, total, piece):
        m = min(piece, total - off)
        data[off:off + m] = torch.randint(0, P, (m,), dtype=torch.int32, device=dev, generator=g)
    return data


@pytest.mark.parametrize("log_n,batch", [(10, (1 << 22) + 5), (20, (1 << 12) + 3), (24, (1 << 8) + 2)])
def test_batch_of_more_than_2_32_elements(ta, log_n, batch):
    free, _ = torch.cuda.mem_get_info()
    n = 1 << log_n
    total = batch * n
    assert total > 1 << 32
    if free < 3 * 4 * total + (4 << 30):                      # data, the untouched copy, the library's intermediate buffer
        pytest.skip("needs ~52 GiB of free device memory")
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    data = fill(total, dev, 4000 + log_n)
    keep = data.clone()
    ctx = ta.NttContext(n)
    try:
        ctx.run_device(data.data_ptr(), data.data_ptr(), batch, False, stream=stream)
        torch.cuda.synchronize()
        wraps = [(1 << 29) // n, (1 << 30) // n, (1 << 31) // n, (1 << 32) // n]
        for b in sorted({0, 1, batch // 2, batch - 2, batch - 1, *[w - 1 for w in wraps], *wraps}):
            want = oracle.ntt(keep[b * n:(b + 1) * n].cpu().numpy().view(np.uint32).astype(np.uint64))
            assert (data[b * n:(b + 1) * n].cpu().numpy().view(np.uint32) == want).all(), f"transform {b} of {batch}"
        ctx.run_device(data.data_ptr(), data.data_ptr(), batch, True, stream=stream)
        torch.cuda.synchronize()
        piece = 1 << 28
        for off in range(0, total, piece):
            assert torch.equal(data[off:off + piece], keep[off:off + piece]), f"round trip differs in elements [{off}, {off + piece})"
    finally:
        ctx.destroy()
        del data, keep
        torch.cuda.empty_cache()


def test_out_of_device_memory_is_a_status_and_the_context_survives(ta):
    # the intermediate buffer of a multi-pass transform is as large as the data: when the device cannot hold it the call returns
    # hipErrorOutOfMemory (2) before any kernel ran, and the same context works once the memory is back
    from toyni_amd._lib import lib
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    n, batch = 1 << 20, 512                                   # 2 GiB of data, 2 GiB of intermediates
    data = fill(batch * n, dev, 77)
    keep = data.clone()
    ctx = ta.NttContext(n)
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    hog = torch.empty(free - (1 << 30), dtype=torch.uint8, device=dev)   # 1 GiB left
    try:
        st = lib.toyni_ntt_device(ctx.handle, data.data_ptr(), data.data_ptr(), batch, 0, stream)
        torch.cuda.synchronize()
        assert st == 2 and b"memory" in lib.toyni_error_string(st)
        assert torch.equal(data, keep), "a refused call must not have touched the data"
        del hog
        torch.cuda.empty_cache()
        assert lib.toyni_ntt_device(ctx.handle, data.data_ptr(), data.data_ptr(), batch, 0, stream) == 0
        torch.cuda.synchronize()
        for b in (0, batch - 1):
            want = oracle.ntt(keep[b * n:(b + 1) * n].cpu().numpy().view(np.uint32).astype(np.uint64))
            assert (data[b * n:(b + 1) * n].cpu().numpy().view(np.uint32) == want).all()
    finally:
        ctx.destroy()
