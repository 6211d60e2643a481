"""Merkle commitment.  CPU part: the oracle's SHA-256 / tree restatement pinned against Python's
hashlib (independent implementation) and the reference's own tests (src/merkle.rs:125-189, restated).  GPU part: the HIP
tree, through the C ABI, byte-identical to the oracle / hashlib."""
import hashlib

import numpy as np
import pytest

import oracle

P = oracle.P


def h(b):
    return hashlib.sha256(b).digest()


def leaf(n):  # src/merkle.rs:129-131
    return int(n).to_bytes(8, "little")


def py_levels(leaves):
    """Independent model of MerkleTree::build_tree (src/merkle.rs:25-48) on hashlib."""
    cur = [h(b"\x00" + l) for l in leaves]
    levels = [cur]
    while len(cur) > 1:
        nxt = [h(b"\x01" + cur[i] + (cur[i + 1] if i + 1 < len(cur) else cur[i])) for i in range(0, len(cur), 2)]
        levels.append(nxt)
        cur = nxt
    return levels


def verify_merkle_proof(leaf_bytes, proof, root):
    """src/merkle.rs:87-101 on hashlib."""
    path, position = proof
    cur = h(b"\x00" + leaf_bytes)
    for sib, is_right in zip(path, position):
        cur = h(b"\x01" + sib + cur) if is_right else h(b"\x01" + cur + sib)
    return cur == root


def oracle_proof(levels, n, index):
    path, position, cur = [], [], index
    for level in levels[:-1]:
        sib = cur + 1 if cur % 2 == 0 else cur - 1
        if sib >= len(level):
            path.append(level[cur].tobytes()); position.append(True)
        else:
            path.append(level[sib].tobytes()); position.append(cur % 2 == 1)
        cur //= 2
    return path, position


# ------------------------------------------------------------------ oracle (CPU)
def test_oracle_sha256_vs_hashlib():
    rng = np.random.default_rng(0)
    for ln in [0, 1, 9, 25, 55, 56, 63, 64, 65, 119, 120, 300]:
        m = rng.integers(0, 256, ln, dtype=np.uint8).tobytes()
        assert oracle.sha256(m) == h(m)
    assert oracle.hash_leaf(b"abc") == h(b"\x00abc")
    assert oracle.hash_node(h(b"l"), h(b"r")) == h(b"\x01" + h(b"l") + h(b"r"))


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 7, 8, 33])
def test_oracle_tree_vs_hashlib_model(n):
    leaves = [leaf(i + 1) for i in range(n)]
    got = oracle.merkle_levels(leaves)
    want = py_levels(leaves)
    assert [len(l) for l in got] == [len(l) for l in want]
    for g, w in zip(got, want):
        assert [x.tobytes() for x in g] == w


def test_reference_merkle_tests_restated():
    # src/merkle.rs:133-189
    for n in (4, 3, 1):
        leaves = [leaf(i) for i in range(1, n + 1)]
        levels = oracle.merkle_levels(leaves)
        root = levels[-1][0].tobytes()
        for i in range(n):
            assert verify_merkle_proof(leaves[i], oracle_proof(levels, n, i), root)
    leaves = [leaf(i) for i in range(1, 5)]
    levels = oracle.merkle_levels(leaves)
    root = levels[-1][0].tobytes()
    assert not verify_merkle_proof(leaf(99), oracle_proof(levels, 4, 0), root)                # wrong leaf rejected
    node_root = oracle.merkle_levels([leaf(1), leaf(2)])[-1][0].tobytes()
    assert oracle.merkle_levels([node_root])[-1][0].tobytes() != node_root                   # leaf/node domain separation


def test_oracle_value_leaf_format():
    # build_merkle_tree: leaf = salt || value.to_bytes(); build_unsalted_tree: leaf = value.to_bytes() (src/fibonacci.rs:340-361)
    vals = oracle.splitmix(6, 77)
    salts = np.arange(96, dtype=np.uint8).reshape(6, 16)
    lv = oracle.merkle_commit_values(vals, salts)
    assert [x.tobytes() for x in lv[0]] == [h(b"\x00" + salts[i].tobytes() + int(vals[i]).to_bytes(8, "little")) for i in range(6)]
    assert lv[-1][0].tobytes() == py_levels([salts[i].tobytes() + int(vals[i]).to_bytes(8, "little") for i in range(6)])[-1][0]
    lu = oracle.merkle_commit_values(vals, None)
    assert lu[-1][0].tobytes() == py_levels([int(v).to_bytes(8, "little") for v in vals])[-1][0]


# ------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def ta():
    import __graft_entry__ as entry
    entry.build_hip()
    import torch  # noqa: F401  (same runtime-order rule as tests/test_gpu_parity.py)
    import toyni_amd
    assert toyni_amd.gpu_available()
    return toyni_amd


@pytest.mark.gpu
@pytest.mark.parametrize("n,salted", [(1, False), (1, True), (2, True), (3, False), (5, True), (8, False), (1000, True), (1 << 12, True), (1 << 16, False),
                                      # ragged sizes above the single-workgroup tail: odd levels inside the two-levels-per-launch kernel
                                      (2049, False), (4097, True), (5001, False), (9999, True), ((1 << 13) + 3, False), (70001, True), (4098, False), (6146, True)])
def test_gpu_tree_vs_oracle(ta, n, salted):
    vals = oracle.splitmix(n, 500 + n)
    salts = np.random.default_rng(n).integers(0, 256, (n, 16), dtype=np.uint8) if salted else None
    tree = ta.MerkleTree(vals, salts)
    want = oracle.merkle_commit_values(vals, salts)
    assert len(tree.levels) == len(want)
    for g, w in zip(tree.levels, want):
        assert (g == w).all()
    assert tree.root() == want[-1][0].tobytes()


@pytest.mark.gpu
def test_gpu_tree_reference_tests(ta):
    # src/merkle.rs:133-189 against the GPU-built tree (leaf(n) = n.to_le_bytes() == the unsalted value leaf)
    for n in (4, 3, 1):
        tree = ta.MerkleTree(np.arange(1, n + 1, dtype=np.uint64))
        for i in range(n):
            assert verify_merkle_proof(leaf(i + 1), tree.get_proof(i), tree.root())
    tree = ta.MerkleTree(np.arange(1, 5, dtype=np.uint64))
    assert not verify_merkle_proof(leaf(99), tree.get_proof(0), tree.root())
    assert tree.get_proof(4) is None
    assert tree.root() == py_levels([leaf(i) for i in range(1, 5)])[-1][0]


@pytest.mark.gpu
def test_gpu_tree_prover_layer_2_21(ta):
    # the prover's largest commitment: a salted tree over an lde_size = 2^21 layer (src/fibonacci.rs:129,153,207)
    n = 1 << 21
    vals = oracle.splitmix(n, 2121)
    salts = np.random.default_rng(21).integers(0, 256, (n, 16), dtype=np.uint8)
    tree = ta.MerkleTree(vals, salts)
    want = oracle.merkle_commit_values(vals, salts)
    assert tree.root() == want[-1][0].tobytes()
    assert (tree.levels[0] == want[0]).all() and (tree.levels[7] == want[7]).all()
    i = 1234567
    assert verify_merkle_proof(salts[i].tobytes() + int(vals[i]).to_bytes(8, "little"), tree.get_proof(i), tree.root())
