"""sha256 over the kernel sources (popcorn_amd/csrc/*.hip, *.h, its Makefile -- the per-file compiler flags are part of the build --,
include/popcorn_hip.h): the stamp that ties the tracked rocprofv3
summaries in profiles/ to the build they were collected from (tools/profile_round.sh writes it, tests/test_bench_contract.py checks it)."""
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_hash(root=ROOT):
    files = sorted(glob.glob(os.path.join(root, "popcorn_amd", "csrc", "*.hip")) + glob.glob(os.path.join(root, "popcorn_amd", "csrc", "*.h")) +
                   [os.path.join(root, "popcorn_amd", "csrc", "Makefile"), os.path.join(root, "include", "popcorn_hip.h")])
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.relpath(f, root).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest(), [os.path.relpath(f, root) for f in files]


if __name__ == "__main__":
    digest, files = csrc_hash()
    out = {"csrc_sha256": digest, "files": files, "tag": sys.argv[1] if len(sys.argv) > 1 else None}
    print(json.dumps(out, indent=1))
