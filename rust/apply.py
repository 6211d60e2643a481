#!/usr/bin/env python3
"""Apply the MI355X drop-in to a Toyni checkout, mechanically (INTEGRATION.md section 1 as a program):

    python rust/apply.py /path/to/toyni            # edits the checkout in place
    python rust/apply.py /path/to/toyni --check    # only says what it would do

  src/ntt.rs      the `#[cfg(feature = "cuda")] mod cuda { .. }` block and the `pub use cuda::{..}` line behind it are replaced by
                  rust/src/ntt_gpu.rs.  Everything above the block (the CPU transform) and below it (the CPU tests) stays byte for
                  byte.  The block is LOCATED in the checkout (attribute + `mod cuda {` + matching brace, by a small Rust-aware
                  scanner) -- nothing of the reference's text is stored here.  The GPU tests that live inside the old block
                  (`#[cfg(test)] mod tests` nested in `mod cuda`) are carried over verbatim FROM THE CHECKOUT into the new `mod gpu`,
                  where the reference's names resolve through test-only aliases.
  build.rs        replaced by rust/build.rs (hipcc, gfx950 only).
  Cargo.toml      the `[features]` table is replaced by rust/Cargo.features.toml (`hip = []`, `cuda = ["hip"]`).
  cuda/           removed.
  hip/            created and filled with the library's sources (the list INTEGRATION.md section 1 gives).

No Rust toolchain is needed (or available in the build image): the result is checked structurally by tests/test_rust_dropin.py."""
import argparse
import os
import re
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
HIP_SOURCES = ["toyni_hip.hip", "ntt_kernels.hpp", "ntt_plan.hpp", "bb_field.hpp", "merkle_kernels.hpp", "prover_kernels.hpp", "multi_gpu.hpp"]


def matching_brace(src: str, open_at: int) -> int:
    """Index of the `}` that closes the `{` at src[open_at], skipping braces inside string / char literals and comments."""
    assert src[open_at] == "{"
    depth, i, n = 0, open_at, len(src)
    while i < n:
        c = src[i]
        if src.startswith("//", i):
            i = src.find("\n", i)
            if i < 0:
                break
            continue
        if src.startswith("/*", i):
            i = src.find("*/", i) + 2
            continue
        if c == '"':
            i += 1
            while src[i] != '"':
                i += 2 if src[i] == "\\" else 1
            i += 1
            continue
        if c == "r" and re.match(r'r#*"', src[i:]):
            hashes = re.match(r"r(#*)\"", src[i:]).group(1)
            i = src.find('"' + hashes, i + 2 + len(hashes)) + 1 + len(hashes)
            continue
        if c == "'":                                # a char literal ('{', '\n', '\'') or a lifetime ('a)
            m = re.match(r"'(\\.|[^\\'])'", src[i:])
            i += m.end() if m else 1
            continue
        if c == "{":
            depth += 1
        elif c == "}":
            depth -= 1
            if depth == 0:
                return i
        i += 1
    raise ValueError("unbalanced braces")


def locate_cuda_block(src: str):
    """(start, end, tests): src[start:end] is what goes -- an optional banner comment, the cfg attribute, `mod cuda { .. }` and the
    `pub use cuda::{..};` re-export behind it; `tests` is the text of the `#[cfg(test)] mod tests { .. }` nested in the block, or ''."""
    m = re.search(r'^#\[cfg\(feature = "cuda"\)\]\s*\n\s*mod cuda\s*\{', src, flags=re.M)
    if not m:
        raise SystemExit("src/ntt.rs: no `#[cfg(feature = \"cuda\")] mod cuda {` block found (already applied?)")
    start = m.start()
    # a banner comment line directly above the attribute (blank lines between them allowed) belongs to the block
    head = src[:start].rstrip("\n")
    last = head[head.rfind("\n") + 1:]
    if last.lstrip().startswith("//") and "cuda" in last.lower():
        start = head.rfind("\n") + 1
    open_at = m.end() - 1
    close_at = matching_brace(src, open_at)
    body = src[open_at + 1:close_at]
    tests = ""
    mt = re.search(r"^[ \t]*#\[cfg\(test\)\]\s*\n[ \t]*mod tests\s*\{", body, flags=re.M)
    if mt:
        t_close = matching_brace(body, mt.end() - 1)
        tests = body[mt.start():t_close + 1]
    end = close_at + 1
    mu = re.match(r'\s*#\[cfg\(feature = "cuda"\)\]\s*\n\s*pub use cuda::\{[^}]*\};[ \t]*\n?', src[end:])
    if not mu:
        raise SystemExit("src/ntt.rs: the `pub use cuda::{..};` re-export behind `mod cuda` was not found")
    end += mu.end()
    return start, end, tests


def new_gpu_block(tests: str) -> str:
    block = open(os.path.join(HERE, "src", "ntt_gpu.rs")).read()
    if not tests:
        return block
    # the carried tests call the reference's names through `use super::*`: give `mod gpu` test-only aliases for them
    aliases = ("    // the reference's GPU tests (carried over from the old block by rust/apply.py) use its names\n"
               "    #[cfg(test)]\n"
               "    use self::{gpu_available as cuda_available, intt_gpu as intt_cuda, ntt_gpu as ntt_cuda, GpuBuffer as CudaBuffer};\n\n")
    m = re.search(r'^#\[cfg\(all\(feature = "hip", has_hip\)\)\]\s*\nmod gpu\s*\{', block, flags=re.M)
    assert m, "rust/src/ntt_gpu.rs: `mod gpu {` not found"
    close_at = matching_brace(block, m.end() - 1)
    return block[:close_at] + "\n" + aliases + tests.rstrip("\n") + "\n" + block[close_at:]


def replace_features(cargo: str) -> str:
    feats = [l for l in open(os.path.join(HERE, "Cargo.features.toml")).read().splitlines() if l.strip() and not l.lstrip().startswith("#")]
    assert feats and feats[0].strip() == "[features]"
    m = re.search(r"^\[features\][ \t]*\n(?:(?!\[)[^\n]*\n)*", cargo, flags=re.M)
    new = "\n".join(feats) + "\n\n"
    if not m:
        return cargo.rstrip("\n") + "\n\n" + new
    return cargo[:m.start()] + new + cargo[m.end():].lstrip("\n")


def apply(checkout: str, check_only: bool = False) -> dict:
    ntt_rs = os.path.join(checkout, "src", "ntt.rs")
    src = open(ntt_rs).read()
    start, end, tests = locate_cuda_block(src)
    patched = src[:start] + new_gpu_block(tests) + ("\n" if not src[end:].startswith("\n") else "") + src[end:]
    report = {"ntt_rs_block_lines": (src[:start].count("\n") + 1, src[:end].count("\n")), "carried_test_lines": tests.count("\n") + (1 if tests else 0),
              "removed": [], "hip_files": []}
    if check_only:
        return report
    with open(ntt_rs, "w") as f:
        f.write(patched)
    shutil.copy(os.path.join(HERE, "build.rs"), os.path.join(checkout, "build.rs"))
    cargo_toml = os.path.join(checkout, "Cargo.toml")
    with open(cargo_toml) as f:
        cargo = f.read()
    with open(cargo_toml, "w") as f:
        f.write(replace_features(cargo))
    cuda_dir = os.path.join(checkout, "cuda")
    if os.path.isdir(cuda_dir):
        report["removed"] = sorted(os.listdir(cuda_dir))
        shutil.rmtree(cuda_dir)
    hip_dir = os.path.join(checkout, "hip")
    os.makedirs(hip_dir, exist_ok=True)
    for name in HIP_SOURCES:
        shutil.copy(os.path.join(ROOT, "toyni_amd", "csrc", name), hip_dir)
    shutil.copy(os.path.join(ROOT, "include", "toyni_hip.h"), hip_dir)
    report["hip_files"] = sorted(os.listdir(hip_dir))
    return report


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("checkout", help="root of a Toyni checkout (the directory that holds Cargo.toml)")
    ap.add_argument("--check", action="store_true", help="locate everything, change nothing")
    args = ap.parse_args()
    if not os.path.exists(os.path.join(args.checkout, "Cargo.toml")):
        sys.exit(f"{args.checkout}: no Cargo.toml")
    rep = apply(args.checkout, args.check)
    a, b = rep["ntt_rs_block_lines"]
    print(f"src/ntt.rs: lines {a}-{b} (`mod cuda` + re-export) {'would be' if args.check else ''} replaced by rust/src/ntt_gpu.rs; "
          f"{rep['carried_test_lines']} lines of GPU tests carried over from the checkout")
    if not args.check:
        print(f"build.rs, Cargo.toml [features] replaced; cuda/ removed ({', '.join(rep['removed']) or 'absent'}); hip/ = {', '.join(rep['hip_files'])}")


if __name__ == "__main__":
    main()
