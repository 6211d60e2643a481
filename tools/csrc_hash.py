#!/usr/bin/env python3
"""sha256 over the library's sources (toyni_amd/csrc/* without csrc/host/, include/*.h, sorted by path): the identity of the kernels a profile measured.
tools/collect_profiles.sh records it next to the counters ON THE BOX; bench.py compares it with the sources it is running from, so a
profile quoted for roofline.traffic is either from exactly these kernels or flagged STALE -- no git needed (the box has no .git)."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha256():
    h = hashlib.sha256()
    files = []
    for base in ("toyni_amd/csrc", "include"):
        for d, dirs, names in os.walk(os.path.join(ROOT, base)):
            if "host" in dirs:
                dirs.remove("host")   # csrc/host/ = compiled-language CALLERS over the C ABI (C++ mirror, prover): not part of the library
            for n in names:
                if n.endswith((".hip", ".hpp", ".h")):
                    files.append(os.path.relpath(os.path.join(d, n), ROOT))
    for rel in sorted(files):
        h.update(rel.encode() + b"\0")
        h.update(open(os.path.join(ROOT, rel), "rb").read())
    return h.hexdigest()


if __name__ == "__main__":
    print(csrc_sha256())
