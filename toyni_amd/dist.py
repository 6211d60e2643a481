"""Multi-GPU forms of the hot path; one process per GPU, torch.distributed for the plumbing.

1. Batched same-size transforms (the prover's repeated workload, BASELINE configs[3]): independent units, so the
   batch is sharded contiguously over ranks and NOTHING is exchanged -- `shard_batch` + the ordinary batched
   entry point.  `max_over_ranks` is the only collective and it carries a timing scalar, not data.

2. One transform too large / too slow for one GPU (configs[4], n up to the field's limit 2^27 -- SURVEY F1):
   4-step decomposition n = n1 * n2 with exactly ONE all-to-all (RCCL over xGMI on GPUs):
       X[k1 + n1 k2] = sum_j2 w_n2^(j2 k2) * [ w_n^(j2 k1) * sum_j1 x[j1 n2 + j2] w_n1^(j1 k1) ]
   rank g starts with n2/G columns (j2 in its chunk), runs the n1-point column transforms and the twiddle
   locally, the all-to-all turns column ownership into row ownership (k1 in its chunk), and the n2-point row
   transforms finish.  Output layout: rank g holds X[k1 + n1 k2] for k1 in its chunk as a [n1/G, n2] array
   (`fourstep_output_index` gives the natural index).  `fourstep_inverse` is the exact mirror.

3. The same transform WITHOUT local transposes (`slab_forward` / `slab_inverse`, include/toyni_hip.h section 2b): split
   n = M1 * S1 where M1 is the first-pass size of the single-device plan.  That pass only couples elements of one
   column of the [M1][S1] view, so each rank runs it on its own column slab with the ordinary strided pass kernel
   (twiddle fused, column offset = its slab's position); the all-to-all then moves CONTIGUOUS row blocks, one relayout
   makes rows contiguous, and the remaining work is a plain batch of size-S1 transforms.  4 HBM sweeps per direction
   at n = 2^27 against 8 for form 2 (which pays for two transposes, a separate twiddle pass and pack / unpack copies).
   Same input / output layouts as form 2 with (n1, n2) = (M1, S1).

The reference has no counterpart (single device, no collectives: cuda/ntt_kernel.cu:246-248); values are pinned by
the single-device transform / the oracle on the gathered result.

The local work goes through a small `LocalOps` interface so that tests can run the SAME exchange logic over gloo
on CPU tensors (tests/test_dist_cpu.py injects an oracle-backed LocalOps; this package ships only `HipLocalOps`).
"""
import os
from typing import Tuple

import torch
import torch.distributed as dist


def shard_batch(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous shard [start, start+count) of `total` independent transforms for `rank`; remainders go to the low ranks."""
    base, rem = divmod(total, world)
    count = base + (1 if rank < rem else 0)
    start = rank * base + min(rank, rem)
    return start, count


def max_over_ranks(seconds: float, device=None) -> float:
    """The bench contract's timing reduction (max over ranks).  No-op without a process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def fourstep_split(log_n: int, world: int) -> Tuple[int, int]:
    """(log n1, log n2) with n1 >= n2 and both divisible by the world size."""
    l1 = (log_n + 1) // 2
    l2 = log_n - l1
    assert world & (world - 1) == 0, "world size must be a power of two"
    assert (1 << l2) >= world, "transform too small to split over this many ranks"
    return l1, l2


def fourstep_output_index(log_n: int, world: int, rank: int) -> torch.Tensor:
    """Natural index k = k1 + n1*k2 of every element of rank's [n1/G, n2] output block (int64)."""
    l1, l2 = fourstep_split(log_n, world)
    n1, n2 = 1 << l1, 1 << l2
    r = n1 // world
    k1 = torch.arange(rank * r, (rank + 1) * r, dtype=torch.int64).unsqueeze(1)
    k2 = torch.arange(n2, dtype=torch.int64).unsqueeze(0)
    return k1 + n1 * k2


def fourstep_input_index(log_n: int, world: int, rank: int) -> torch.Tensor:
    """Natural index j = j1*n2 + j2 of every element of rank's [n1, n2/G] input block (int64)."""
    l1, l2 = fourstep_split(log_n, world)
    n1, n2 = 1 << l1, 1 << l2
    c = n2 // world
    j1 = torch.arange(n1, dtype=torch.int64).unsqueeze(1)
    j2 = torch.arange(rank * c, (rank + 1) * c, dtype=torch.int64).unsqueeze(0)
    return j1 * n2 + j2


def first_pass_log(log_n: int) -> int:
    """log2 M1 of the single-device plan; 0 when the transform is single-pass.  Asked of the LIBRARY (toyni_first_pass_points:
    split_passes of toyni_amd/csrc/ntt_plan.hpp, no device needed), not re-derived here: the split is the slab layout contract, and
    a second copy of the rule could drift from the one the kernels follow."""
    from ._lib import lib
    m1 = int(lib.toyni_first_pass_points(1 << log_n)) if 0 <= log_n <= 27 else 0
    return m1.bit_length() - 1 if m1 else 0


def slab_split(log_n: int, world: int) -> Tuple[int, int]:
    """(log M1, log S1) of the slab form; every rank needs >= 32 columns and >= 1 row."""
    l1 = first_pass_log(log_n)
    assert l1 > 0, "transform too small to split (n <= 1024 is a single pass)"
    ls = log_n - l1
    assert world & (world - 1) == 0, "world size must be a power of two"
    assert (1 << ls) >= 32 * world and (1 << l1) >= world, "transform too small to split over this many ranks"
    return l1, ls


def slab_input_index(log_n: int, world: int, rank: int) -> torch.Tensor:
    """Natural index j = j1*S1 + j' of every element of rank's [M1, S1/G] input slab (int64)."""
    l1, ls = slab_split(log_n, world)
    w = (1 << ls) // world
    j1 = torch.arange(1 << l1, dtype=torch.int64).unsqueeze(1)
    jc = torch.arange(rank * w, (rank + 1) * w, dtype=torch.int64).unsqueeze(0)
    return j1 * (1 << ls) + jc


def slab_output_index(log_n: int, world: int, rank: int) -> torch.Tensor:
    """Natural index k = k1 + M1*k' of every element of rank's [M1/G, S1] output block (int64)."""
    l1, ls = slab_split(log_n, world)
    r = (1 << l1) // world
    k1 = torch.arange(rank * r, (rank + 1) * r, dtype=torch.int64).unsqueeze(1)
    kp = torch.arange(1 << ls, dtype=torch.int64).unsqueeze(0)
    return k1 + (1 << l1) * kp


class HipLocalOps:
    """Local stages on the GPU through the C ABI (contexts cached per size)."""

    def __init__(self, log_n: int, device: torch.device):
        from . import ntt as _ntt
        self._ntt = _ntt
        self.device = device
        self.big = _ntt.NttContext(1 << log_n, device=device.index)
        self._ctx = {}

    def _ctx_for(self, m: int):
        if m not in self._ctx:
            self._ctx[m] = self._ntt.NttContext(m, device=self.device.index)
        return self._ctx[m]

    def ntt_rows(self, t: torch.Tensor, inverse: bool) -> None:
        assert t.is_contiguous() and t.dtype == torch.int32 and t.dim() == 2
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self._ctx_for(t.shape[1]).run_device(t.data_ptr(), t.data_ptr(), t.shape[0], inverse, stream=stream)

    def twiddle(self, t: torch.Tensor, row0: int, inverse: bool) -> None:
        from ._lib import check, lib
        assert t.is_contiguous() and t.dtype == torch.int32 and t.dim() == 2
        stream = torch.cuda.current_stream(self.device).cuda_stream
        check(lib.toyni_fourstep_twiddle_device(self.big.handle, t.data_ptr(), t.shape[0], t.shape[1], row0, int(inverse), stream or None),
              "4-step twiddle failed")


    # ---- slab form (include/toyni_hip.h section 2b)
    def slab_pass(self, slab: torch.Tensor, col_base: int, inverse: bool) -> None:
        from ._lib import check, lib
        assert slab.is_contiguous() and slab.dtype == torch.int32 and slab.dim() == 2
        assert slab.shape[0] == lib.toyni_ntt_ctx_first_pass_points(self.big.handle), "slab rows must be the plan's first-pass size"
        stream = torch.cuda.current_stream(self.device).cuda_stream
        check(lib.toyni_ntt_slab_pass_device(self.big.handle, slab.data_ptr(), slab.shape[1], col_base, int(inverse), stream or None),
              "slab pass failed")

    def relayout(self, src: torch.Tensor, dst: torch.Tensor, rows: int, row0: int, parts: int, inverse: bool) -> None:
        from ._lib import check, lib
        assert src.is_contiguous() and dst.is_contiguous() and src.dtype == dst.dtype == torch.int32 and src.numel() == dst.numel()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        check(lib.toyni_ntt_slab_relayout_device(self.big.handle, src.data_ptr(), dst.data_ptr(), rows, row0, parts, int(inverse), stream or None),
              "slab relayout failed")

    def slab_rows(self, src: torch.Tensor, dst: torch.Tensor, rows: int, row0: int, parts: int, inverse: bool) -> None:
        """Relayout + size-S1 row transforms as one step (toyni_ntt_slab_rows_device): forward pieces -> transformed rows, inverse rows ->
        inverse-transformed, twiddled pieces.  Where the launch's pass shapes allow it there is no relayout sweep at all (the row
        transforms address the pieces layout directly); `last_rows_fused` says which form ran.  The inverse form may overwrite src."""
        import ctypes
        from ._lib import check, lib
        assert src.is_contiguous() and dst.is_contiguous() and src.dtype == dst.dtype == torch.int32 and src.numel() == dst.numel()
        s1 = src.numel() // rows
        stream = torch.cuda.current_stream(self.device).cuda_stream
        fused = ctypes.c_int(0)
        check(lib.toyni_ntt_slab_rows_device(self.big.handle, self._ctx_for(s1).handle, src.data_ptr(), dst.data_ptr(), rows, row0, parts,
                                             int(inverse), ctypes.byref(fused), stream or None), "slab rows failed")
        self.last_rows_fused = bool(fused.value)


class PhaseClock:
    """Optional per-phase timing of the multi-device forms (bench.py): `mark(name)` records a HIP event on the device's current stream
    and charges the time since the previous mark to `name`.  The collective is stream-ordered against the current stream, so the mark
    behind it fires when the exchange has landed.  `ms()` waits for the device and returns {name: milliseconds summed over all marks}."""

    def __init__(self, device):
        self.device, self.marks = device, []

    def mark(self, name: str) -> None:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream(self.device))
        self.marks.append((name, ev))

    def ms(self) -> dict:
        torch.cuda.synchronize(self.device)
        out = {}
        for (_, e0), (name, e1) in zip(self.marks, self.marks[1:]):
            if name != "start":
                out[name] = out.get(name, 0.0) + e0.elapsed_time(e1)
        return out


def _mark(clock, name: str) -> None:
    if clock is not None:
        clock.mark(name)


def _exchange(send: torch.Tensor, group=None) -> torch.Tensor:
    """The one all-to-all: send[h] goes to rank h; returns recv with recv[g] from rank g."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return send                                # one rank: the exchange is the identity (no copy; callers never reuse `send`)
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send, group=group)
    return recv


def fourstep_forward(cols: torch.Tensor, log_n: int, ops, rank: int = 0, world: int = 1, group=None, clock=None) -> torch.Tensor:
    """cols: [n1, n2/G] int32 (element (j1, jc) = x[j1*n2 + rank*n2/G + jc]).  Returns [n1/G, n2] (see fourstep_output_index)."""
    l1, l2 = fourstep_split(log_n, world)
    n1, n2 = 1 << l1, 1 << l2
    c, r = n2 // world, n1 // world
    assert cols.shape == (n1, c)
    _mark(clock, "start")
    t = cols.t().contiguous()                      # [c, n1]: row = column j2 of the matrix
    if t.data_ptr() == cols.data_ptr():            # a one-column block transposes to a view: do not overwrite the caller's input
        t = t.clone()
    _mark(clock, "transpose_pack")
    ops.ntt_rows(t, False)                         # n1-point transforms over j1
    _mark(clock, "local_transforms")
    ops.twiddle(t, rank * c, False)                # * w_n^(j2 k1)
    _mark(clock, "twiddle")
    send = t.view(c, world, r).permute(1, 0, 2).contiguous()   # [G, c, r]: block h = k1 in rank h's chunk
    _mark(clock, "transpose_pack")
    recv = _exchange(send, group)                  # [G, c, r]: block g = j2 in rank g's chunk
    _mark(clock, "exchange")
    rows = recv.permute(2, 0, 1).reshape(r, n2).contiguous()   # [k1_local, j2]
    _mark(clock, "transpose_pack")
    ops.ntt_rows(rows, False)                      # n2-point transforms over j2
    _mark(clock, "local_transforms")
    return rows


def fourstep_inverse(rows: torch.Tensor, log_n: int, ops, rank: int = 0, world: int = 1, group=None, clock=None) -> torch.Tensor:
    """Mirror of fourstep_forward: [n1/G, n2] block of X -> [n1, n2/G] column block of x."""
    l1, l2 = fourstep_split(log_n, world)
    n1, n2 = 1 << l1, 1 << l2
    c, r = n2 // world, n1 // world
    assert rows.shape == (r, n2)
    _mark(clock, "start")
    b = rows.contiguous().clone()
    _mark(clock, "transpose_pack")
    ops.ntt_rows(b, True)                          # inverse n2-point over k2 -> [k1_local, j2], scaled by n2^-1
    _mark(clock, "local_transforms")
    send = b.view(r, world, c).permute(1, 2, 0).contiguous()   # [G, c, r]: block h = j2 in rank h's chunk
    _mark(clock, "transpose_pack")
    recv = _exchange(send, group)                  # [G, c, r]: block g = k1 in rank g's chunk
    _mark(clock, "exchange")
    t = recv.permute(1, 0, 2).reshape(c, n1).contiguous()      # [jc, k1]
    _mark(clock, "transpose_pack")
    ops.twiddle(t, rank * c, True)                 # * w_n^-(j2 k1)
    _mark(clock, "twiddle")
    ops.ntt_rows(t, True)                          # inverse n1-point over k1, scaled by n1^-1
    _mark(clock, "local_transforms")
    out = t.t().contiguous()
    _mark(clock, "transpose_pack")
    return out


def _exchange_blocks_async(send_blocks, recv_blocks, rank: int, world: int, group=None):
    """Block h of `send_blocks` goes to rank h, block g of `recv_blocks` comes from rank g (contiguous tensors, no
    packing).  Point-to-point sends / receives issued as one batch (one grouped RCCL call on GPUs) and NOT waited for:
    returns the work handles.  The local block is a plain copy."""
    recv_blocks[rank].copy_(send_blocks[rank])
    if world == 1:
        return []
    p2p = []
    for off in range(1, world):
        dst, src = (rank + off) % world, (rank - off) % world
        p2p.append(dist.P2POp(dist.isend, send_blocks[dst], dst, group))
        p2p.append(dist.P2POp(dist.irecv, recv_blocks[src], src, group))
    return dist.batch_isend_irecv(p2p)


def slab_forward(slab: torch.Tensor, log_n: int, ops, rank: int = 0, world: int = 1, group=None, chunks: int = 1, clock=None) -> torch.Tensor:
    """slab: [M1, S1/G] int32, element (j1, c) = x[j1*S1 + rank*S1/G + c]; OVERWRITTEN.  Returns [M1/G, S1] (slab_output_index).

    chunks > 1: the exchange is issued as `chunks` asynchronous pieces (sub-blocks of every destination's row block) and
    the relayout + row transforms of piece q run while pieces q+1.. are still on the wire."""
    l1, ls = slab_split(log_n, world)
    m1, s1 = 1 << l1, 1 << ls
    w, r = s1 // world, m1 // world
    assert slab.shape == (m1, w) and slab.is_contiguous()
    _mark(clock, "start")
    ops.slab_pass(slab, rank * w, False)           # M1-point column transforms * w_n^(j' k1), in place, no transpose
    _mark(clock, "slab_pass")
    rows = torch.empty((r, s1), dtype=slab.dtype, device=slab.device)
    if chunks <= 1:
        recv = _exchange(slab.view(world, r, w), group)  # row block h (k1 in rank h's chunk) is contiguous: no packing
        _mark(clock, "exchange")
        if hasattr(ops, "slab_rows"):              # (round 5) one step: the row transforms read the [G][r][w] pieces directly -- no relayout sweep
            _mark(clock, "relayout")
            ops.slab_rows(recv, rows, r, rank * r, world, False)
        else:
            ops.relayout(recv, rows, r, rank * r, world, False)   # [G][r][w] pieces -> contiguous rows [r][S1]
            _mark(clock, "relayout")
            ops.ntt_rows(rows, False)              # what is left: size-S1 transforms over j'
        _mark(clock, "row_transforms")
        return rows
    assert chunks & (chunks - 1) == 0 and r % chunks == 0, "chunks must be a power of two dividing the rows per rank"
    rq = r // chunks
    sv = slab.view(world, chunks, rq, w)           # [h, q] = rows q*rq.. of rank h's block: contiguous
    pending = []
    for q in range(chunks):                        # everything is put on the wire first, in consumption order
        recv_q = torch.empty((world, rq, w), dtype=slab.dtype, device=slab.device)
        works = _exchange_blocks_async([sv[h, q] for h in range(world)], list(recv_q.unbind(0)), rank, world, group)
        pending.append((recv_q, works))
    for q, (recv_q, works) in enumerate(pending):
        for wk in works:
            wk.wait()
        part = rows[q * rq:(q + 1) * rq]
        if hasattr(ops, "slab_rows"):
            ops.slab_rows(recv_q, part, rq, rank * r + q * rq, world, False)
        else:
            ops.relayout(recv_q, part, rq, rank * r + q * rq, world, False)
            ops.ntt_rows(part, False)
    _mark(clock, "exchange_relayout_rows_pipelined")
    return rows


def slab_inverse(rows: torch.Tensor, log_n: int, ops, rank: int = 0, world: int = 1, group=None, chunks: int = 1, clock=None) -> torch.Tensor:
    """Mirror of slab_forward: [M1/G, S1] block of X (OVERWRITTEN) -> [M1, S1/G] slab of x.  chunks > 1: piece q is on the
    wire while the row transforms of piece q+1 run."""
    l1, ls = slab_split(log_n, world)
    m1, s1 = 1 << l1, 1 << ls
    w, r = s1 // world, m1 // world
    assert rows.shape == (r, s1) and rows.is_contiguous()
    _mark(clock, "start")
    if chunks <= 1:
        send = torch.empty((world, r, w), dtype=rows.dtype, device=rows.device)
        if hasattr(ops, "slab_rows"):              # (round 5) one step: the last pass of the inverse row transforms writes the twiddled pieces
            ops.slab_rows(rows, send, r, rank * r, world, True)
            _mark(clock, "row_transforms")
        else:
            ops.ntt_rows(rows, True)               # inverse size-S1 over k', scaled by 1/S1
            _mark(clock, "row_transforms")
            ops.relayout(rows, send, r, rank * r, world, True)    # rows -> [G][r][w] pieces, times w_n^-(k1 j')
        _mark(clock, "relayout")
        slab = _exchange(send, group).view(m1, w)  # block g = k1 in rank g's chunk: the [M1][w] slab
        _mark(clock, "exchange")
    else:
        assert chunks & (chunks - 1) == 0 and r % chunks == 0, "chunks must be a power of two dividing the rows per rank"
        rq = r // chunks
        slab = torch.empty((m1, w), dtype=rows.dtype, device=rows.device)
        sv = slab.view(world, chunks, rq, w)
        works, keep = [], []
        for q in range(chunks):
            part = rows[q * rq:(q + 1) * rq]
            send_q = torch.empty((world, rq, w), dtype=rows.dtype, device=rows.device)
            if hasattr(ops, "slab_rows"):
                ops.slab_rows(part, send_q, rq, rank * r + q * rq, world, True)
            else:
                ops.ntt_rows(part, True)
                ops.relayout(part, send_q, rq, rank * r + q * rq, world, True)
            works += _exchange_blocks_async(list(send_q.unbind(0)), [sv[g, q] for g in range(world)], rank, world, group)
            keep.append(send_q)                    # alive until the sends have completed
        for wk in works:
            wk.wait()
        _mark(clock, "exchange_relayout_rows_pipelined")
    ops.slab_pass(slab, rank * w, True)            # inverse M1-point column transforms, scaled by 1/M1
    _mark(clock, "slab_pass")
    return slab
