"""Worker for tests/test_dist_cpu.py: world_size-2 (or more) gloo run of toyni_amd.dist on CPU tensors.
The exchange logic is the shipped code; the local stages are an oracle-backed LocalOps (test only)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from toyni_amd import dist as tdist  # noqa: E402

P = oracle.P


class OracleLocalOps:
    def __init__(self, log_n):
        self.n = 1 << log_n
        self.w = oracle.roots_of_unity_domain(self.n)

    def ntt_rows(self, t, inverse):
        a = t.numpy()
        for i in range(a.shape[0]):
            row = a[i].astype(np.uint64)
            a[i] = (oracle.intt(row) if inverse else oracle.ntt(row)).astype(np.int32)

    def twiddle(self, t, row0, inverse):
        a = t.numpy()
        rows, ln = a.shape
        e = (np.arange(row0, row0 + rows, dtype=np.uint64)[:, None] * np.arange(ln, dtype=np.uint64)[None, :]) % np.uint64(self.n)
        if inverse:
            e = (np.uint64(self.n) - e) % np.uint64(self.n)
        a[:] = ((a.astype(np.uint64) * self.w[e]) % np.uint64(P)).astype(np.int32)


    # ---- slab form: numpy restatements of the two device stages (tests only)
    def slab_pass(self, slab, col_base, inverse):
        a = slab.numpy()
        m1, w = a.shape
        for c in range(w):
            col = a[:, c].astype(np.uint64)
            if inverse:
                a[:, c] = oracle.intt(col).astype(np.int32)
            else:
                y = oracle.ntt(col)
                e = (np.uint64(col_base + c) * np.arange(m1, dtype=np.uint64)) % np.uint64(self.n)
                a[:, c] = ((y * self.w[e]) % np.uint64(P)).astype(np.int32)

    def relayout(self, src, dst, rows, row0, parts, inverse):
        s1 = src.numel() // rows
        w = s1 // parts
        if not inverse:
            dst.copy_(src.view(parts, rows, w).permute(1, 0, 2).reshape(dst.shape))
            return
        a = src.view(rows, s1).numpy().astype(np.uint64)
        e = (np.arange(row0, row0 + rows, dtype=np.uint64)[:, None] * np.arange(s1, dtype=np.uint64)[None, :]) % np.uint64(self.n)
        e = (np.uint64(self.n) - e) % np.uint64(self.n)
        tw = torch.from_numpy(((a * self.w[e]) % np.uint64(P)).astype(np.int32))
        dst.copy_(tw.view(rows, parts, w).permute(1, 0, 2).reshape(dst.shape))


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()

    # ---- 1. batch sharding: shards tile the batch exactly; timing reduction is a max
    total = 1024 + 3
    mine = tdist.shard_batch(total, world, rank)
    all_shards = [None] * world
    dist.all_gather_object(all_shards, mine)
    pos = 0
    for s, c in all_shards:
        assert s == pos and c >= total // world
        pos += c
    assert pos == total
    assert tdist.max_over_ranks(1.0 + rank) == float(world)

    # batched transforms, sharded, no data exchange: every rank transforms its own shard and the union equals the batch
    n, batch = 64, 10
    x = oracle.splitmix(n * batch, 77).reshape(batch, n)
    s, c = tdist.shard_batch(batch, world, rank)
    ops = OracleLocalOps(6)
    local = torch.from_numpy(x[s:s + c].astype(np.int32).copy())
    ops.ntt_rows(local, False)
    gathered = [None] * world
    dist.all_gather_object(gathered, local.numpy())
    full = np.concatenate(gathered)
    assert (full.astype(np.uint64) == np.stack([oracle.ntt(r) for r in x])).all()

    # ---- 2. 4-step with one all-to-all
    for log_n in (6, 9, 12):
        if (1 << (log_n // 2)) < world:
            continue
        nn = 1 << log_n
        x = oracle.splitmix(nn, 1000 + log_n)
        want = oracle.ntt(x)
        ops = OracleLocalOps(log_n)
        cols = torch.from_numpy(x[tdist.fourstep_input_index(log_n, world, rank).numpy()].astype(np.int32))
        out = tdist.fourstep_forward(cols, log_n, ops, rank, world)
        idx = tdist.fourstep_output_index(log_n, world, rank).numpy()
        assert (out.numpy().astype(np.uint64) == want[idx]).all(), f"4-step forward log_n={log_n} rank={rank}"
        back = tdist.fourstep_inverse(out, log_n, ops, rank, world)
        assert torch.equal(back, cols), f"4-step inverse log_n={log_n} rank={rank}"

    # ---- 3. slab form (no local transposes), same single all-to-all
    for log_n in (13, 14, 16):
        l1 = tdist.first_pass_log(log_n)
        if (1 << (log_n - l1)) < 32 * world:
            continue
        nn = 1 << log_n
        x = oracle.splitmix(nn, 2000 + log_n)
        want = oracle.ntt(x)
        ops = OracleLocalOps(log_n)
        slab = torch.from_numpy(x[tdist.slab_input_index(log_n, world, rank).numpy()].astype(np.int32))
        keep = slab.clone()
        out = tdist.slab_forward(slab, log_n, ops, rank, world)
        idx = tdist.slab_output_index(log_n, world, rank).numpy()
        assert (out.numpy().astype(np.uint64) == want[idx]).all(), f"slab forward log_n={log_n} rank={rank}"
        back = tdist.slab_inverse(out, log_n, ops, rank, world)
        assert torch.equal(back, keep), f"slab inverse log_n={log_n} rank={rank}"
        # the same with the exchange issued as asynchronous pieces (batched isend / irecv) overlapped with the row work
        for chunks in (2, 4):
            if ((1 << l1) // world) % chunks:
                continue
            out = tdist.slab_forward(keep.clone(), log_n, ops, rank, world, chunks=chunks)
            assert (out.numpy().astype(np.uint64) == want[idx]).all(), f"chunked slab forward log_n={log_n} rank={rank} chunks={chunks}"
            back = tdist.slab_inverse(out, log_n, ops, rank, world, chunks=chunks)
            assert torch.equal(back, keep), f"chunked slab inverse log_n={log_n} rank={rank} chunks={chunks}"
        # every output index is owned exactly once
        owned = [None] * world
        dist.all_gather_object(owned, idx.reshape(-1))
        assert sorted(np.concatenate(owned).tolist()) == list(range(nn))

    dist.barrier()
    if rank == 0:
        print("DIST OK", world)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
