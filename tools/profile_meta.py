"""Fingerprint of the tower's kernel sources, so that numbers read from a COMMITTED profile cannot outlive the code they
measured.  bench.py puts `roofline.from_profile.gemm_in_situ` and `roofline.traffic` into its line only while the
fingerprint stored beside the profile equals the one of the sources it runs (no git needed: the GPU box has no .git).

    python tools/profile_meta.py stamp <csv or json> [batch]   # writes <file>.meta.json next to a profile (on the GPU box,
                                                               # in the same call that produced it: tools/round_profile.sh)
    python tools/profile_meta.py show
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = ["image_search_amd/csrc/vit_kernels.h", "image_search_amd/csrc/attn32_kernels.h", "image_search_amd/csrc/vit.hip",
           "image_search_amd/csrc/knn_kernels.h", "image_search_amd/csrc/knn.hip", "image_search_amd/csrc/pipeline.hip"]


def git_blob_id(path: str) -> str:
    """what `git hash-object` prints: sha1 of "blob <size>\\0" + content"""
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def fingerprint() -> dict:
    return {p: git_blob_id(os.path.join(ROOT, p)) for p in SOURCES}


def stamp(profile_path: str, batch: int = 256) -> str:
    meta = {"profile": os.path.basename(profile_path), "profile_blob": git_blob_id(profile_path), "batch": batch,
            "model": "ViT-L/14 (24 layers, hidden 1024, ff 4096, 257 tokens)", "sources": fingerprint()}
    out = profile_path + ".meta.json"
    with open(out, "w") as f:
        json.dump(meta, f, indent=1)
    return out


def check(profile_path: str):
    """(meta or None, reason it must not be used or None)"""
    try:
        meta = json.load(open(profile_path + ".meta.json"))
    except (OSError, ValueError):
        return None, "no .meta.json beside it (taken before round 6: the code it measured is not on record)"
    if meta.get("profile_blob") != git_blob_id(profile_path):
        return meta, "the profile changed after it was stamped"
    now = fingerprint()
    changed = [p for p, b in meta.get("sources", {}).items() if now.get(p) != b]
    if changed or set(now) - set(meta.get("sources", {})):
        return meta, "kernel sources changed since the profile was taken: " + ", ".join(changed or sorted(set(now) - set(meta["sources"])))
    return meta, None


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "stamp":
        print(stamp(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 256))
    else:
        print(json.dumps(fingerprint(), indent=1))
