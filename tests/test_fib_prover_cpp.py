"""BASELINE configs[0] / [2] with a COMPILED caller: toyni_amd/csrc/host/fib_prover.hpp (the C++ counterpart of
StarkProver::generate_proof, src/fibonacci.rs:99-310, every heavy step a call of the C ABI) run as the program tests/cpp/fib_prove.cpp.
Its serialized proof is checked by the CPU restatement of the reference verifier (tests/harness/fib_verifier.py, src/verifier.rs);
a corrupted trace must be refused like src/fibonacci.rs:430-442."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _exe():
    import __graft_entry__ as entry
    entry.build_hip()
    return entry.build_fib_prove()


def _run(args, timeout=600):
    res = subprocess.run([_exe()] + [str(a) for a in args], capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert lines, res.stdout[-2000:] + res.stderr[-2000:]
    return res.returncode, json.loads(lines[-1])


def _load_proof(path):
    """The program's JSON -> the dict tests/harness/fib_prover.py::expand_proof takes (same serialized opening records)."""
    raw = json.load(open(path))
    proof = dict(raw)
    for k in ("trace_commitment", "quotient_commitment"):
        proof[k] = bytes.fromhex(raw[k])
    proof["fri_commitments"] = [bytes.fromhex(h) for h in raw["fri_commitments"]]
    proof["opening_records"] = np.frombuffer(bytes.fromhex(raw["opening_records"]), dtype=np.uint8)
    proof["opening_groups"] = [(g[0], g[1], g[2]) for g in raw["opening_groups"]]
    return proof


def test_cpp_prover_builds_and_self_skips_without_gpu():
    import toyni_amd
    if toyni_amd.gpu_available():
        pytest.skip("covered by the gpu-marked run")
    rc, out = _run([64, 1, 1])
    assert rc == 0 and out == {"gpu": False}           # src/ntt.rs:265-268: no device, no proof, no CPU fallback


def test_host_sha256_and_transcript_match_hashlib(tmp_path):
    """The prover's host-side SHA-256 (the device kernels' compression function compiled for the host) against hashlib at every
    padding boundary, and its transcript against the Python restatement of src/transcript.rs."""
    src = tmp_path / "t.cpp"
    src.write_text('''#include <cstdio>
#include "%s/toyni_amd/csrc/host/fib_prover.hpp"
int main() {
    using namespace toyni::fib;
    std::vector<uint8_t> m;
    for (int len = 0; len < 200; ++len) {
        auto h = sha256(m.data(), m.size());
        for (int i = 0; i < 32; ++i) std::printf("%%02x", h[i]);
        std::printf("\\n");
        m.push_back((uint8_t)(len * 37 + 11));
    }
    Transcript tr;
    uint8_t root[32];
    for (int i = 0; i < 32; ++i) root[i] = (uint8_t)i;
    tr.absorb(root, 32);
    tr.absorb_field(123456789u);
    std::printf("%%u\\n", tr.squeeze_challenge());
    for (uint32_t v : tr.squeeze_indices(44, 1u << 20)) std::printf("%%u ", v);
    std::printf("\\n%%u %%u\\n", root_of_unity(16), powmod(7, P - 2));
    return 0;
}''' % ROOT)
    exe = tmp_path / "t"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wno-unknown-pragmas", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src),
                           "-L", os.path.join(ROOT, "toyni_amd", "lib"), "-ltoyni_hip", f"-Wl,-rpath,{os.path.join(ROOT, 'toyni_amd', 'lib')}",
                           "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60).stdout.splitlines()
    import hashlib
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from harness import fib_verifier
    m = b""
    for ln in range(200):
        assert out[ln] == hashlib.sha256(m).hexdigest(), ln
        m += bytes([(ln * 37 + 11) & 255])
    tr = fib_verifier.Transcript()
    tr.absorb(bytes(range(32)))
    tr.absorb_field(123456789)
    assert int(out[200]) == tr.squeeze_challenge()
    assert [int(v) for v in out[201].split()] == tr.squeeze_indices(44, 1 << 20)
    assert [int(v) for v in out[202].split()] == [fib_verifier.root_of_unity(16), pow(7, fib_verifier.P - 2, fib_verifier.P)]


@pytest.mark.gpu
@pytest.mark.parametrize("n,lde,folds,final", [(64, 2048, 8, 8), (8, 256, 8, 1), (1 << 16, 1 << 21, 17, 16),
                                               # in between: masks longer than the trace (n < 140), and every plan the LDE sizes 2^9 .. 2^19 take
                                               (16, 512, 8, 2), (128, 4096, 9, 8), (256, 8192, 9, 16), (1024, 1 << 15, 11, 16), (1 << 14, 1 << 19, 15, 16)])
def test_cpp_proof_is_accepted_by_the_verifier_restatement(tmp_path, n, lde, folds, final):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from harness import fib_prover, fib_verifier
    path = tmp_path / "proof.json"
    rc, out = _run([n, 5, 1, path, "--phases"])
    assert rc == 0 and out["gpu"], out
    assert (out["trace_len"], out["lde_size"], out["folds"], out["final_layer_size"]) == (n, lde, folds, final)
    proof = fib_prover.expand_proof(_load_proof(path))
    why = []
    assert fib_verifier.verify(proof, why), why
    assert len(proof["query_proofs"]) == 44 and len(proof["fri_commitments"]) == folds + 1 and len(set(proof["fri_final_layer"])) == 1
    # tampering is caught on the C++ prover's proofs exactly as on the harness' (src/verifier.rs:303-379)
    bad = dict(proof, t_z=(proof["t_z"] + 1) % fib_verifier.P)
    why = []
    assert not fib_verifier.verify(bad, why) and why == ["ood"]
    # two proofs of the same trace under different keys differ in their masked openings (:303-312)
    path2 = tmp_path / "proof2.json"
    rc, _ = _run([n, 6, 1, path2])
    assert rc == 0
    other = fib_prover.expand_proof(_load_proof(path2))
    assert other["t_z"] != proof["t_z"] and fib_verifier.verify(other)


@pytest.mark.gpu
def test_cpp_prover_refuses_a_non_fibonacci_trace():
    rc, out = _run([64, 3, 1, "--corrupt-row", 10])
    assert rc == 1 and out["error"] == "Constraint check at z failed"        # src/fibonacci.rs:430-442


@pytest.mark.gpu
def test_chacha20_keystream_rfc8439_vector():
    """toyni_chacha20_fill_device against RFC 8439 section 2.3.2 (key 00..1f, nonce 00 00 00 09 00 00 00 4a 00 00 00 00, counter 1):
    the library's nonce layout is 0 || nonce_lo || nonce_hi, so that vector's nonce is (0x09000000, 0x4a000000)... its first word is
    non-zero, which the ABI cannot express; the all-zero-nonce vector of section 2.4.2's sibling (A.1 test vector #1 / #2: key 0,
    nonce 0, counters 0 and 1) can."""
    import ctypes
    import torch
    import toyni_amd
    lib = toyni_amd._lib.lib
    out = torch.empty(128, dtype=torch.uint8, device="cuda")
    key = (ctypes.c_uint8 * 32)()
    assert lib.toyni_chacha20_fill_device(out.data_ptr(), 128, key, 0, None) == 0
    torch.cuda.synchronize()
    got = out.cpu().numpy().tobytes().hex()
    # RFC 8439 appendix A.1, test vectors #1 (block counter 0) and #2 (block counter 1), key = 0, nonce = 0
    want0 = ("76b8e0ada0f13d90405d6ae55386bd28bdd219b8a08ded1aa836efcc8b770dc7"
             "da41597c5157488d7724e03fb8d84a376a43b8f41518a11cc387b669b2ee6586")
    want1 = ("9f07e7be5551387a98ba977c732d080dcb0f29a048e3656912c6533e32ee7aed"
             "29b721769ce64e43d57133b074d839d531ed1f28510afb45ace10a1f4b794d6f")
    assert got == want0 + want1
