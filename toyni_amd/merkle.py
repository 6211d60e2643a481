"""Merkle commitment of a layer of field elements on the GPU.

Mirrors `MerkleTree` (src/merkle.rs:9-85) as the prover uses it through build_merkle_tree / build_unsalted_tree
(src/fibonacci.rs:340-361): the tree is built on the device; `levels`, `root()` and `get_proof()` expose the same data the
reference holds.  Verification (src/merkle.rs:87-101) is CPU-only in the reference (the verifier never uses the GPU) and is
not part of this package."""
import numpy as np

from ._lib import check, lib


def level_sizes(n: int):
    sizes = []
    while n >= 1:
        sizes.append(n)
        if n == 1:
            break
        n = (n + 1) // 2
    return sizes


class MerkleTree:
    """Tree over leaves salt(16) || value(8, LE) (salts given) or value(8, LE) (salts None)."""

    def __init__(self, values, salts=None):
        v = np.ascontiguousarray(values, dtype=np.uint64)
        n = v.size
        assert n > 0
        s = None
        if salts is not None:
            s = np.ascontiguousarray(salts, dtype=np.uint8).reshape(n, 16)
        flat = np.empty((lib.toyni_merkle_total_digests(n), 32), dtype=np.uint8)
        check(lib.toyni_merkle_commit_host(v.ctypes.data, s.ctypes.data if s is not None else None, n, flat.ctypes.data), "GPU Merkle commit failed")
        self.n = n
        self.levels, off = [], 0
        for m in level_sizes(n):
            self.levels.append(flat[off:off + m])
            off += m

    def root(self) -> bytes:  # src/merkle.rs:82-84
        return self.levels[-1][0].tobytes()

    def get_proof(self, index: int):
        """src/merkle.rs:50-80: (path, position) with position[i] = True when the sibling is on the LEFT of the running hash."""
        if index >= self.n:
            return None
        path, position, cur = [], [], index
        for level in self.levels[:-1]:
            sib = cur + 1 if cur % 2 == 0 else cur - 1
            if sib >= len(level):  # last node of an odd level: paired with itself, treated as right sibling
                path.append(level[cur].tobytes())
                position.append(True)
            else:
                path.append(level[sib].tobytes())
                position.append(cur % 2 == 1)
            cur //= 2
        return path, position


def merkle_commit_device(d_values: int, d_salts: int, n: int, d_levels: int, stream: int = 0) -> None:
    """Device-resident form: packed u32 values, optional 16-byte salts, all levels written to d_levels."""
    check(lib.toyni_merkle_commit_device(d_values, d_salts or None, n, d_levels, stream or None), "GPU Merkle commit failed")
