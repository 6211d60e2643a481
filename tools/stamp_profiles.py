#!/usr/bin/env python3
"""profiles/PROVENANCE.json: for every committed profile that bench.py quotes, the commit that last touched it and whether
toyni_amd/csrc has changed since (the GPU box receives a snapshot without .git, so bench.py cannot ask git there).  Run before
committing new profiles or kernel changes:  python tools/stamp_profiles.py"""
import glob
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def git(*args):
    return subprocess.run(["git", *args], cwd=ROOT, capture_output=True, text=True, check=True).stdout.strip()


out = {}
for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic*.json")) + glob.glob(os.path.join(ROOT, "profiles", "r*_fold_stats.csv"))):
    rel = os.path.relpath(f, ROOT)
    h = git("log", "-1", "--format=%h", "--", rel)
    if not h:
        out[rel] = {"commit": "uncommitted", "csrc_changed_since": False}
        continue
    newer = git("log", "--format=%h", f"{h}..HEAD", "--", "toyni_amd/csrc").split()
    dirty = bool(git("status", "--porcelain", "--", "toyni_amd/csrc"))
    out[rel] = {"commit": h, "csrc_changed_since": bool(newer) or dirty}
json.dump(out, open(os.path.join(ROOT, "profiles", "PROVENANCE.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1))
