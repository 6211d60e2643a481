"""GPU tests of round 2's host-facing additions, through the C ABI:
  * toyni_ntt_slab_multi_gpu_{host,device}: ONE transform over G lanes from one process (include/toyni_hip.h 2c).  The box has one
    GPU, so the lanes share device 0 -- the exchange-by-copy path; with RCCL the group has one rank (send/recv to self).
  * per-stream intermediates: one context driven from two streams at once (the async contract; reference src/ntt.rs:128-141
    serialises nothing and shares d_data)
  * context-free host entry points after the staging rewrite (fold, Ext fold, Merkle): repeated calls, zero points."""
import ctypes

import os

import numpy as np
import pytest

import oracle
from oracle import P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ta():
    import __graft_entry__ as entry
    entry.build_hip()
    import torch
    assert torch.cuda.is_available(), "GPU tests need a device"
    torch.cuda.init()
    import toyni_amd
    assert toyni_amd.gpu_available()
    return toyni_amd


@pytest.mark.parametrize("log_n,lanes", [(11, 1), (12, 2), (13, 2), (16, 8), (20, 1), (20, 4), (21, 8), (22, 2), (24, 8)])
def test_slab_multi_gpu_host_natural_order(ta, log_n, lanes):
    n = 1 << log_n
    x = oracle.splitmix(n, 4100 + log_n + lanes)
    v = x.copy()
    ta.ntt_slab_multi_gpu_host(v, [0] * lanes)
    assert (v == oracle.ntt(x)).all(), "forward"
    ta.ntt_slab_multi_gpu_host(v, [0] * lanes, inverse=True)
    assert (v == x).all(), "inverse"


def test_slab_multi_gpu_host_matches_single_device_at_2_27(ta):
    # BASELINE configs[4] at the field's limit (2^28 does not exist), 8 lanes
    n = 1 << 27
    x = oracle.splitmix(n, 2727)
    want = x.copy()
    ta.ntt.get_or_create_ctx(n).run_host(want, False)
    v = x.copy()
    ta.ntt_slab_multi_gpu_host(v, [0] * 8)
    assert (v == want).all()
    ta.ntt_slab_multi_gpu_host(v, [0] * 8, inverse=True)
    assert (v == x).all()
    ta.ntt.get_or_create_ctx(n).trim()


@pytest.mark.parametrize("log_n,lanes", [(16, 2), (22, 8)])
def test_slab_multi_gpu_device_layouts(ta, log_n, lanes):
    import torch
    from toyni_amd import dist as tdist
    dev = torch.device("cuda", 0)
    n = 1 << log_n
    x = oracle.splitmix(n, 4300 + log_n)
    want = oracle.ntt(x)
    slabs = [torch.from_numpy(x[tdist.slab_input_index(log_n, lanes, g).numpy()].astype(np.int32)).to(dev) for g in range(lanes)]
    keep = [t.clone() for t in slabs]
    rows = [torch.empty(n // lanes, dtype=torch.int32, device=dev) for _ in range(lanes)]
    torch.cuda.synchronize()
    ta.ntt_slab_multi_gpu_device(n, [0] * lanes, [t.data_ptr() for t in slabs], [t.data_ptr() for t in rows])
    for h in range(lanes):
        idx = tdist.slab_output_index(log_n, lanes, h).numpy().reshape(-1)
        assert (rows[h].cpu().numpy().view(np.uint32).astype(np.uint64) == want[idx]).all(), f"rows of lane {h}"
    ta.ntt_slab_multi_gpu_device(n, [0] * lanes, [t.data_ptr() for t in slabs], [t.data_ptr() for t in rows], inverse=True)
    for g in range(lanes):
        assert torch.equal(slabs[g], keep[g]), f"slab of lane {g}"


def test_slab_multi_gpu_exchange_in_pieces():
    """TOYNI_SLAB_PIECES = K: the peer-copy exchange issued as K pieces (read once per process, hence a child per value).  8 lanes on
    device 0, n = 2^21 and 2^16, host form in natural order: forward vs the oracle, inverse back."""
    import subprocess
    import sys
    code = (
        "import numpy as np, oracle, toyni_amd\n"
        "for log_n, lanes in ((21, 8), (16, 4), (13, 2)):\n"
        "    n = 1 << log_n\n"
        "    x = oracle.splitmix(n, 8800 + log_n)\n"
        "    v = x.copy()\n"
        "    toyni_amd.ntt_slab_multi_gpu_host(v, [0] * lanes)\n"
        "    assert (v == oracle.ntt(x)).all()\n"
        "    toyni_amd.ntt_slab_multi_gpu_host(v, [0] * lanes, inverse=True)\n"
        "    assert (v == x).all()\n"
        "print('PIECES OK')\n")
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for k in ("2", "4", "16"):
        env = dict(os.environ, TOYNI_SLAB_PIECES=k, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
        res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=root)
        assert res.returncode == 0 and "PIECES OK" in res.stdout, k + ": " + res.stdout[-500:] + res.stderr[-2000:]


def test_slab_multi_gpu_rccl_group_of_one(ta):
    # the RCCL exchange (dlopen'ed librccl, ncclCommInitAll, grouped send/recv) with the one device of the box: one rank
    n = 1 << 18
    x = oracle.splitmix(n, 4400)
    v = x.copy()
    ta.ntt_slab_multi_gpu_host(v, [0], exchange=ta.ntt.EXCHANGE_RCCL)
    assert (v == oracle.ntt(x)).all()
    ta.ntt_slab_multi_gpu_host(v, [0], inverse=True, exchange=ta.ntt.EXCHANGE_RCCL)
    assert (v == x).all()
    with pytest.raises(ta._lib.ToyniError):          # one communicator rank per device: duplicate lanes need the peer-copy form
        ta.ntt_slab_multi_gpu_host(v, [0, 0], exchange=ta.ntt.EXCHANGE_RCCL)


def _tools_child(code, extra_env=None):
    """`code` in a child interpreter bound to libtoyni_hip_tools.so: the fault-injection hook toyni_tools_inject lives there only."""
    import os
    import subprocess
    import sys
    import __graft_entry__ as entry
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""), TOYNI_LIB_OVERRIDE=entry.build_tools(),
               **(extra_env or {}))
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=root)


def test_slab_group_refuses_lanes_without_peer_access():
    """a device pair without direct peer access must surface as TOYNI_E_NO_PEER_ACCESS when the group is built
    (hipDeviceEnablePeerAccess errors used to be swallowed and hipMemcpyPeerAsync silently staged through the host).  One GPU here,
    so the denial is injected (bit 0: every pair of different lanes counts as two devices that cannot reach each other)."""
    code = (
        "import ctypes, numpy as np, oracle, toyni_amd\n"
        "lib = toyni_amd._lib.lib\n"
        "n = 1 << 17\n"
        "x = oracle.splitmix(n, 9100); v = x.copy()\n"
        "devs = (ctypes.c_int * 2)(0, 0)\n"
        "assert lib.toyni_tools_inject(1) == 0\n"
        "rc = lib.toyni_ntt_slab_multi_gpu_host(devs, 2, n, v.ctypes.data, 0, 0)\n"
        "assert rc == 10009, rc\n"
        "assert (v == x).all(), 'a refused call must not touch the data'\n"
        "assert b'peer access' in lib.toyni_error_string(rc)\n"
        "one = (ctypes.c_int * 1)(0)\n"
        "assert lib.toyni_ntt_slab_multi_gpu_host(one, 1, n, v.ctypes.data, 0, 0) == 0   # a single lane has no peers\n"
        "assert (v == oracle.ntt(x)).all()\n"
        "assert lib.toyni_tools_inject(0) == 0\n"
        "v = x.copy()\n"
        "assert lib.toyni_ntt_slab_multi_gpu_host(devs, 2, n, v.ctypes.data, 0, 0) == 0   # the failed group was not cached\n"
        "assert (v == oracle.ntt(x)).all()\n"
        "print('PEER OK')\n")
    res = _tools_child(code)
    assert res.returncode == 0 and "PEER OK" in res.stdout, res.stdout[-500:] + res.stderr[-2000:]
    assert "no peer access" in res.stderr


def test_slab_group_first_use_self_check():
    """on first use a group whose exchange crosses devices runs one n = 2^18 transform through that exchange and compares it
    with the single-device transform.  Forced here for lanes on one device (bit 1), and with one corrupted word (bit 2) it must fail
    with TOYNI_E_SELF_CHECK before the caller's data is touched."""
    code = (
        "import ctypes, numpy as np, oracle, toyni_amd\n"
        "lib = toyni_amd._lib.lib\n"
        "n = 1 << 20\n"
        "x = oracle.splitmix(n, 9200); v = x.copy()\n"
        "devs = (ctypes.c_int * 4)(0, 0, 0, 0)\n"
        "lib.toyni_tools_inject(2)\n"
        "assert lib.toyni_ntt_slab_multi_gpu_host(devs, 4, n, v.ctypes.data, 0, 0) == 0\n"
        "assert (v == oracle.ntt(x)).all()\n"
        "lib.toyni_tools_inject(2 | 4)\n"
        "v = x.copy()\n"
        "rc = lib.toyni_ntt_slab_multi_gpu_host(devs, 4, n, v.ctypes.data, 0, 0)\n"
        "assert rc == 10010, rc\n"
        "assert (v == x).all()\n"
        "lib.toyni_tools_inject(0)\n"
        "print('SELF-CHECK OK')\n")
    res = _tools_child(code, {"TOYNI_VERBOSE": "1"})
    assert res.returncode == 0 and "SELF-CHECK OK" in res.stdout, res.stdout[-500:] + res.stderr[-2000:]
    assert "self-check (4 lanes, peer-copy exchange" in res.stderr and ": ok" in res.stderr and "self-check failed" in res.stderr
    assert "lane 3/4 -> device 0 (PCI" in res.stderr          # the lane -> device table of TOYNI_VERBOSE


def test_multi_device_exchange_on_two_real_devices(ta):
    """Skipped on the one-GPU boxes of this pool: the first place where hipMemcpyPeerAsync, cross-device event waits and an RCCL
    group of more than one rank actually run.  Both exchange kinds, host form, against the oracle."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    ndev = 2 if torch.cuda.device_count() < 4 else 4
    for exchange in (ta.ntt.EXCHANGE_PEER_COPY, ta.ntt.EXCHANGE_RCCL):
        n = 1 << 20
        x = oracle.splitmix(n, 9300 + exchange)
        v = x.copy()
        ta.ntt_slab_multi_gpu_host(v, list(range(ndev)), exchange=exchange)
        assert (v == oracle.ntt(x)).all()
        ta.ntt_slab_multi_gpu_host(v, list(range(ndev)), inverse=True, exchange=exchange)
        assert (v == x).all()


def test_slab_multi_gpu_rejects_bad_arguments(ta):
    lib = ta._lib.lib
    v = np.zeros(1 << 16, dtype=np.uint64)
    devs3 = (ctypes.c_int * 3)(0, 0, 0)
    assert lib.toyni_ntt_slab_multi_gpu_host(devs3, 3, 1 << 16, v.ctypes.data, 0, 0) == 10006       # lanes not a power of two
    devs1 = (ctypes.c_int * 1)(0)
    assert lib.toyni_ntt_slab_multi_gpu_host(devs1, 1, 1 << 10, v.ctypes.data, 0, 0) == 10001       # single-pass size: nothing to split
    assert lib.toyni_ntt_slab_multi_gpu_host(devs1, 1, 1 << 16, v.ctypes.data, 0, 7) == 10006       # unknown exchange
    bad = (ctypes.c_int * 1)(99)
    assert lib.toyni_ntt_slab_multi_gpu_host(bad, 1, 1 << 16, v.ctypes.data, 0, 0) == 10006         # no such device
    devs16 = (ctypes.c_int * 16)(*([0] * 16))
    assert lib.toyni_ntt_slab_multi_gpu_host(devs16, 16, 1 << 11, v.ctypes.data, 0, 0) == 10006     # < 32 columns per lane


def test_two_streams_on_one_context_interleaved(ta):
    """two streams on ONE context at n = 2^20, 50 interleaved launches, bit-exact.  Each stream owns its
    intermediate buffer, so the launches of the two streams may overlap on the device."""
    import torch
    dev = torch.device("cuda", 0)
    n = 1 << 20
    ctx = ta.NttContext(n)
    xa, xb = oracle.splitmix(n, 51), oracle.splitmix(n, 52)
    wa, wb = oracle.ntt(xa), oracle.ntt(xb)
    ta_, tb_ = (torch.from_numpy(v.astype(np.int32)).to(dev) for v in (xa, xb))
    oa, ob = torch.empty_like(ta_), torch.empty_like(tb_)
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    for i in range(50):
        inv = bool(i & 1)                       # forward out of place, then inverse back: both streams keep working sets alive
        ctx.run_device(oa.data_ptr() if inv else ta_.data_ptr(), ta_.data_ptr() if inv else oa.data_ptr(), 1, inv, stream=s1.cuda_stream)
        ctx.run_device(ob.data_ptr() if inv else tb_.data_ptr(), tb_.data_ptr() if inv else ob.data_ptr(), 1, inv, stream=s2.cuda_stream)
    ctx.synchronize(s1.cuda_stream)
    ctx.synchronize(s2.cuda_stream)
    assert (oa.cpu().numpy().view(np.uint32) == wa).all() and (ob.cpu().numpy().view(np.uint32) == wb).all()
    assert (ta_.cpu().numpy().view(np.uint32) == xa).all() and (tb_.cpu().numpy().view(np.uint32) == xb).all()
    # growing the batch on one stream while the other keeps launching: no hipFree on the enqueue path, results stay exact
    big = torch.from_numpy(np.concatenate([xa, xb, xa, xb]).astype(np.int32)).to(dev)
    for i in range(10):
        ctx.run_device(ta_.data_ptr(), oa.data_ptr(), 1, False, stream=s1.cuda_stream)
        if i == 3:
            ctx.run_device(big.data_ptr(), big.data_ptr(), 4, False, stream=s2.cuda_stream)
    ctx.synchronize(s1.cuda_stream)
    ctx.synchronize(s2.cuda_stream)
    got = big.cpu().numpy().view(np.uint32).reshape(4, n)
    assert (got[0] == wa).all() and (got[1] == wb).all() and (got[3] == wb).all() and (oa.cpu().numpy().view(np.uint32) == wa).all()
    ctx.trim()
    ctx.destroy()


def test_many_streams_evict_scratch_sets(ta):
    # more streams than the context keeps buffer sets for (8): the least recently used set is retired, nothing breaks
    import torch
    dev = torch.device("cuda", 0)
    n = 1 << 14
    ctx = ta.NttContext(n)
    x = oracle.splitmix(n, 77)
    want = oracle.ntt(x)
    streams = [torch.cuda.Stream(device=dev) for _ in range(12)]
    bufs = [torch.from_numpy(x.astype(np.int32)).to(dev) for _ in streams]
    torch.cuda.synchronize()
    for rnd in range(3):
        for st, b in zip(streams, bufs):
            ctx.run_device(b.data_ptr(), b.data_ptr(), 1, bool(rnd & 1), stream=st.cuda_stream)
    torch.cuda.synchronize()
    for b in bufs:
        assert (b.cpu().numpy().view(np.uint32) == want).all()
    ctx.destroy()


def test_short_lived_streams_do_not_accumulate_scratch(ta):
    """a caller that cycles through short-lived streams used to grow device memory without bound (an evicted
    scratch set waited for toyni_ntt_ctx_trim: its stream might be gone, so nothing could be recorded on it at eviction time).  Now
    every call of a multi-stream context leaves a fence on its stream and evicted sets are freed by a later call once their fence has
    completed.  64 streams x (256 x 2^20): each set holds a 1 GiB intermediate; without the fix this would pin ~56 GiB."""
    import torch
    lib = ta._lib.lib
    n, batch = 1 << 20, 256
    set_bytes = batch * n * 4
    ctx = ta.NttContext(n)
    data = torch.randint(0, P, (batch * n,), dtype=torch.int32, device="cuda")
    keep = data.clone()

    def one_stream():
        st = ctypes.c_void_p()
        assert lib.toyni_stream_create(ctypes.byref(st), -1) == 0
        ctx.run_device(data.data_ptr(), data.data_ptr(), batch, False, stream=st.value)
        ctx.run_device(data.data_ptr(), data.data_ptr(), batch, True, stream=st.value)
        assert lib.toyni_stream_destroy(st) == 0          # waits, then destroys: the next stream starts on settled data

    torch.cuda.synchronize()
    for _ in range(8):                                     # fill the context's 8 cached sets
        one_stream()
    torch.cuda.synchronize()
    free_start = torch.cuda.mem_get_info()[0]
    low = free_start
    for _ in range(56):
        one_stream()
        low = min(low, torch.cuda.mem_get_info()[0])
    torch.cuda.synchronize()
    one_stream()                                           # any later call frees what has drained (no trim, no API-level sync)
    free_end = torch.cuda.mem_get_info()[0]
    assert torch.equal(data, keep)
    assert free_start - free_end <= set_bytes + (64 << 20), f"{(free_start - free_end) / 2**30:.2f} GiB still held after 64 streams"
    assert free_start - low <= 3 * set_bytes, f"peak growth {(free_start - low) / 2**30:.2f} GiB: evicted sets are not being freed as the streams come and go"
    # growing the batch on ONE stream that is never synchronised through the API: every outgrown buffer is fenced and freed too
    st = ctypes.c_void_p()
    assert lib.toyni_stream_create(ctypes.byref(st), -1) == 0
    ctx2 = ta.NttContext(n)
    free0 = torch.cuda.mem_get_info()[0]
    for b in range(8, 264, 8):
        ctx2.run_device(data.data_ptr(), data.data_ptr(), b, False, stream=st.value)
        ctx2.run_device(data.data_ptr(), data.data_ptr(), b, True, stream=st.value)
    torch.cuda.synchronize()
    ctx2.run_device(data.data_ptr(), data.data_ptr(), 1, False, stream=st.value)
    ctx2.run_device(data.data_ptr(), data.data_ptr(), 1, True, stream=st.value)
    torch.cuda.synchronize()
    held = free0 - torch.cuda.mem_get_info()[0]
    assert held <= (256 + 16) * n * 4, f"{held / 2**30:.2f} GiB held after growing to 256 transforms (sum of all sizes would be 16.5 GiB)"
    assert torch.equal(data, keep)
    assert lib.toyni_stream_destroy(st) == 0
    ctx.destroy()
    ctx2.destroy()


def test_host_fold_staging_repeated_calls_and_zero_points(ta):
    rng = np.random.default_rng(9)
    for m in (2, 6, 64, 1 << 12, 1 << 16, 10, 1 << 12):       # growing and shrinking: the staging set is reused
        e = rng.integers(0, P, m, dtype=np.uint64)
        xs = rng.integers(1, P, m, dtype=np.uint64)
        beta = int(rng.integers(0, P))
        assert (ta.fri_fold(e, xs, beta) == oracle.fri_fold(e, xs, beta)).all(), m
        e4 = rng.integers(0, P, (m, 4), dtype=np.uint64)
        b4 = rng.integers(0, P, 4, dtype=np.uint64)
        assert (ta.fri_fold_ext(e4, xs, b4) == oracle.fri_fold_ext(e4, xs, b4)).all(), m
    # non-canonical inputs reduce like BabyBear::new (src/babybear.rs:26-30)
    e = rng.integers(0, 2**63, 256, dtype=np.uint64)
    xs = rng.integers(1, P, 256, dtype=np.uint64) + np.uint64(P)
    assert (ta.fri_fold(e, xs, P + 5) == oracle.fri_fold(e % P, xs % P, 5)).all()
    # a zero point: "Cannot invert zero" (src/babybear.rs:112) -- also when it only appears after reduction, only within xs[0 .. m/2)
    xs = rng.integers(1, P, 64, dtype=np.uint64)
    e = rng.integers(0, P, 64, dtype=np.uint64)
    xs[40] = 0                                                   # beyond m/2: never read (src/math/fri.rs:31-36)
    assert (ta.fri_fold(e, xs, 3) == oracle.fri_fold(e, np.where(xs == 0, 1, xs), 3)).all()
    xs[13] = P
    with pytest.raises(AssertionError, match="Cannot invert zero"):
        ta.fri_fold(e, xs, 3)
    with pytest.raises(AssertionError, match="Cannot invert zero"):
        ta.fri_fold_ext(rng.integers(0, P, (64, 4), dtype=np.uint64), xs, [1, 2, 3, 4])


def test_device_fold_xs_isolates_a_zero_point(ta):
    """in the explicit-point device kernels four points share one Fermat inversion; a zero among them must not
    zero its neighbours' inverses.  The zero's own inverse is 0 = pow(0, p - 2)."""
    lib = ta._lib.lib
    m = 64
    rng = np.random.default_rng(10)
    e = rng.integers(0, P, m, dtype=np.uint64)
    xs = rng.integers(1, P, m // 2, dtype=np.uint64)
    xs[5] = 0
    beta = 123456
    want = oracle.fri_fold(e, np.concatenate([np.where(xs == 0, 1, xs), np.ones(m // 2, dtype=np.uint64)]), beta)
    a, b = int(e[5]), int(e[5 + m // 2])
    want[5] = oracle.bb_mul(oracle.bb_add(a, b), (P + 1) // 2)  # x^-1 = 0: only the average survives
    bufs = []
    for arr in (e.astype(np.uint32), xs.astype(np.uint32), np.zeros(m // 2, dtype=np.uint32)):
        p = ctypes.c_void_p()
        assert lib.toyni_malloc(ctypes.byref(p), arr.nbytes) == 0
        assert lib.toyni_memcpy_h2d(p, arr.ctypes.data, arr.nbytes) == 0
        bufs.append(p)
    assert lib.toyni_fri_fold_xs_device(bufs[0], bufs[1], bufs[2], m, beta, None) == 0
    out = np.empty(m // 2, dtype=np.uint32)
    assert lib.toyni_memcpy_d2h(out.ctypes.data, bufs[2], out.nbytes) == 0
    assert (out == want).all()
    for p in bufs:
        lib.toyni_free(p)


def test_host_merkle_staging_repeated(ta):
    rng = np.random.default_rng(11)
    for n in (1, 3, 1000, 1 << 14, 5):
        vals = rng.integers(0, P, n, dtype=np.uint64)
        salts = rng.integers(0, 256, (n, 16), dtype=np.uint8)
        for s in (None, salts):
            t = ta.MerkleTree(vals, s)
            assert t.root() == oracle.merkle_commit_values(vals, s)[-1][0].tobytes(), (n, s is None)


def test_pinned_host_memory_takes_the_pipelined_path(ta):
    """Host slices in pinned memory (toyni_host_alloc): chunked, double-buffered, upload / kernels / download on three streams.
    Same values as the plain path on pageable memory, in both directions, with a ragged last chunk and a coset shift."""
    n, batch = 1 << 18, 100                       # 200 MiB of u64: three 64 MiB chunks of 32 transforms and a ragged one of 4
    x = oracle.splitmix(n * batch, 9090)
    ctx = ta.NttContext(n)
    want = x.copy()
    ctx.run_host(want, False, batch=batch)        # pageable numpy memory: the plain path
    for b in (0, 31, 32, 63, 96, 99):
        assert (want[b * n:(b + 1) * n] == oracle.ntt(x[b * n:(b + 1) * n])).all(), b
    pin = ta.PinnedArray(n * batch)
    pin.array[:] = x
    ctx.run_host(pin.array, False, batch=batch)
    assert (pin.array == want).all()
    ctx.run_host(pin.array, True, batch=batch)
    assert (pin.array == x).all()
    ctx.run_host(pin.array, False, batch=batch, shift=7)
    want7 = x.copy()
    ctx.run_host(want7, False, batch=batch, shift=7)
    assert (pin.array == want7).all()
    # and through the multi-GPU batch runner (two lanes on the one device, each lane pipelining its own shard)
    pin.array[:] = x
    ta.ntt_host_multi_gpu(pin.array, n, [0, 0], inverse=False)
    assert (pin.array == want).all()
    pin.free()
    ctx.destroy()


def test_device_calls_can_be_captured_into_a_graph(ta):
    """A warm context enqueues kernels only (no allocation, no synchronisation), so a caller may capture toyni_ntt_device calls
    into a HIP graph and replay them: forward 2^20 and coset round trip 2^16, replayed on fresh contents, bit-exact."""
    import torch
    dev = torch.device("cuda", 0)
    for log_n, batch in ((20, 1), (16, 3), (21, 1)):
        n = 1 << log_n
        ctx = ta.NttContext(n)
        buf = torch.zeros(batch * n, dtype=torch.int32, device=dev)
        out = torch.empty_like(buf)
        s = torch.cuda.Stream(device=dev)
        ctx.run_device(buf.data_ptr(), out.data_ptr(), batch, False, stream=s.cuda_stream)      # warm: buffers of this stream exist
        ctx.run_device(out.data_ptr(), out.data_ptr(), batch, True, stream=s.cuda_stream)
        ctx.synchronize(s.cuda_stream)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            ctx.run_device(buf.data_ptr(), out.data_ptr(), batch, False, stream=s.cuda_stream)
        for rep in range(3):
            x = oracle.splitmix(batch * n, 900 + 10 * log_n + rep)
            buf.copy_(torch.from_numpy(x.astype(np.int32)))
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            got = out.cpu().numpy().view(np.uint32).reshape(batch, n)
            for b in range(batch):
                assert (got[b] == oracle.ntt(x.reshape(batch, n)[b])).all(), (log_n, rep, b)
        del g
        ctx.destroy()


def test_stream_plumbing_for_hosts_without_a_hip_binding(ta):
    """include/toyni_hip.h section 4 (round 3): streams, stream-ordered copies / fills and cross-stream ordering as a Rust caller would
    use them between the library's calls -- here the shape of the C++ prover's first steps: fill, upload, transform on stream A,
    consume on stream B after toyni_stream_wait, download."""
    lib = ta._lib.lib
    n = 1 << 16
    x = oracle.splitmix(n, 7700).astype(np.uint32)
    sa, sb = ctypes.c_void_p(), ctypes.c_void_p()
    assert lib.toyni_stream_create(ctypes.byref(sa), -1) == 0 and lib.toyni_stream_create(ctypes.byref(sb), -1) == 0
    d, e = ctypes.c_void_p(), ctypes.c_void_p()
    assert lib.toyni_malloc(ctypes.byref(d), 8 * n) == 0 and lib.toyni_malloc(ctypes.byref(e), 4 * n) == 0
    pin = ta.PinnedArray(n)                                    # u64 elements of pinned host memory: used as 2 n u32 words
    host = pin.array.view(np.uint32)
    host[:n] = x
    ctx = ta.NttContext(n)
    assert lib.toyni_memset_async(d, 0, 8 * n, sa) == 0        # second half stays zero
    assert lib.toyni_memcpy_h2d_async(d, host.ctypes.data, 4 * n, sa) == 0
    ctx.run_device(d.value, d.value, 1, False, stream=sa.value)
    assert lib.toyni_stream_wait(sb, sa) == 0                  # B's work starts after A's transform
    assert lib.toyni_memcpy_d2d_async(e, d, 4 * n, sb) == 0
    ctx.run_device(e.value, e.value, 1, True, stream=sb.value)
    assert lib.toyni_memcpy_d2h_async(host.ctypes.data + 4 * n, e, 4 * n, sb) == 0
    assert lib.toyni_stream_synchronize(None, sb) == 0
    assert (host[n:2 * n] == x).all()                          # forward on A, inverse on B: the round trip crossed the streams in order
    out = np.empty(2 * n, dtype=np.uint32)
    assert lib.toyni_stream_synchronize(None, sa) == 0
    assert lib.toyni_memcpy_d2h(out.ctypes.data, d, 8 * n) == 0
    assert (out[:n] == oracle.ntt(x.astype(np.uint64))).all() and (out[n:] == 0).all()
    assert lib.toyni_stream_wait(sa, sa) == 0                  # a stream waiting for itself is a no-op
    # the two-step form: mark a point of A now, let B wait for it later
    ev = ctypes.c_void_p()
    assert lib.toyni_event_create(ctypes.byref(ev)) == 0
    ctx.run_device(d.value, d.value, 1, True, stream=sa.value)                      # d[:n] back to x on A
    assert lib.toyni_event_record(ev, sa) == 0
    assert lib.toyni_memset_async(ctypes.c_void_p(d.value + 4 * n), 0xFF, 4 * n, sa) == 0   # later work on A that B does not wait for
    assert lib.toyni_stream_wait_event(sb, ev) == 0
    assert lib.toyni_memcpy_d2h_async(host.ctypes.data, d, 4 * n, sb) == 0
    assert lib.toyni_stream_synchronize(None, sb) == 0 and lib.toyni_stream_synchronize(None, sa) == 0
    assert (host[:n] == x).all()
    assert lib.toyni_event_destroy(ev) == 0 and lib.toyni_event_destroy(None) == 0
    assert lib.toyni_event_record(None, sa) == 10002 and lib.toyni_stream_wait_event(sb, None) == 10002
    ctx.destroy()
    pin.free()
    for p in (d, e):
        assert lib.toyni_free(p) == 0
    assert lib.toyni_stream_destroy(sa) == 0 and lib.toyni_stream_destroy(sb) == 0 and lib.toyni_stream_destroy(None) == 0


def _dispatch_knobs_set():
    return any(k.startswith("TOYNI_") and k not in ("TOYNI_LAUNCH_LOG", "TOYNI_FUZZ_SEED", "TOYNI_FUZZ_CASES", "TOYNI_LIB_OVERRIDE") for k in os.environ)


@pytest.mark.parametrize("log_n,parts,rows,expect_fused", [
    (21, 2, 64, (True, True)),     # rows of 2^14 = 128 x 128: the 128-point column and closing shapes
    (24, 2, 128, (True, True)),    # rows of 2^16 = 256 x 256: the 256-point column and closing shapes
    (22, 2, 128, (True, True)),    # 128-point first pass, rows of 2^15 = 128 x 256: two-step shapes on both sides
    (22, 2, 64, (True, False)),    # half as many rows: the closing 256-point pass of the inverse takes its three-step latency shape -> two-step form
    (26, 8, 32, (True, True)),     # configs[4]'s shapes one size down: rows of 2^18 = 512 x 512 (the (5,4) column and closing shapes)
    (27, 8, 64, (True, True)),     # BASELINE configs[4]: one rank's block of an 8-rank transform of n = 2^27
    (16, 8, 32, (False, False)),   # rows of 2^8: single-pass row transforms -> the two-step form
    (24, 8, 2, (False, False)),    # two rows of 2^16: three-step latency shapes on both sides
])
def test_slab_rows_fused_equals_relayout_plus_transform(ta, log_n, parts, rows, expect_fused):
    """toyni_ntt_slab_rows_device (round 5): the relayout folded into the row transforms' addressing gives, bit for bit, what
    toyni_ntt_slab_relayout_device + toyni_ntt_device give (those two are pinned on the oracle by the tests above), in both directions;
    and the call reports which form ran."""
    import ctypes
    import torch
    lib = ta._lib.lib
    dev = torch.device("cuda", 0)
    n = 1 << log_n
    big = ta.ntt.get_or_create_ctx(n)
    m1 = lib.toyni_ntt_ctx_first_pass_points(big.handle)
    s1 = n // m1
    row = ta.ntt.get_or_create_ctx(s1)
    row0 = m1 - rows                                  # the last block of rows: the largest twiddle exponents
    g = torch.Generator(device="cpu").manual_seed(log_n * 100 + parts)
    src = torch.randint(0, oracle.P, (rows * s1,), dtype=torch.int32, generator=g).to(dev)
    for inverse in (0, 1):
        a, want, got = src.clone(), torch.empty_like(src), torch.empty_like(src)
        if not inverse:
            assert lib.toyni_ntt_slab_relayout_device(big.handle, a.data_ptr(), want.data_ptr(), rows, row0, parts, 0, None) == 0
            assert lib.toyni_ntt_device(row.handle, want.data_ptr(), want.data_ptr(), rows, 0, None) == 0
        else:
            assert lib.toyni_ntt_device(row.handle, a.data_ptr(), a.data_ptr(), rows, 1, None) == 0
            assert lib.toyni_ntt_slab_relayout_device(big.handle, a.data_ptr(), want.data_ptr(), rows, row0, parts, 1, None) == 0
        b = src.clone()
        fused = ctypes.c_int(-1)
        assert lib.toyni_ntt_slab_rows_device(big.handle, row.handle, b.data_ptr(), got.data_ptr(), rows, row0, parts, inverse, ctypes.byref(fused), None) == 0
        torch.cuda.synchronize()
        if not _dispatch_knobs_set():   # which form runs is the DEFAULT dispatch's choice; under a dispatch knob (tools/knob_soak.sh) only the results are judged
            assert bool(fused.value) == expect_fused[inverse], (log_n, parts, rows, inverse, fused.value)
        assert torch.equal(got, want), f"n=2^{log_n} parts={parts} rows={rows} inverse={inverse}: fused rows differ from relayout + transform"
        if fused.value and not inverse:
            assert torch.equal(b, src), "the fused forward form must not touch its input"
    big.trim()
    # bad arguments: same buffer both sides, a row context of the wrong size
    assert lib.toyni_ntt_slab_rows_device(big.handle, row.handle, src.data_ptr(), src.data_ptr(), rows, row0, parts, 0, None, None) == 10006   # TOYNI_E_RANGE
    assert lib.toyni_ntt_slab_rows_device(big.handle, big.handle, src.data_ptr(), got.data_ptr(), rows, row0, parts, 0, None, None) == 10006   # TOYNI_E_RANGE
