"""What a profile was taken of: sha256 over the kernel sources (csrc/*.hip, *.cpp, *.h, Makefile, include/vfsms.h -- a content hash that is the
same here and on the GPU box, whoever built the .so), sha256 of the built library, and the commit the tree was built at (written by the
Makefile into imagestitch_amd/lib/BUILD_INFO when .git is there).  tools/profile_round.sh stamps it on the PMC summary; bench.py compares."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def src_sha256():
    h = hashlib.sha256()
    c = os.path.join(ROOT, "imagestitch_amd", "csrc")
    files = sorted(glob.glob(os.path.join(c, "*.hip")) + glob.glob(os.path.join(c, "*.cpp")) + glob.glob(os.path.join(c, "*.h")) +
                   [os.path.join(c, "Makefile"), os.path.join(ROOT, "include", "vfsms.h")])
    for f in files:
        h.update(os.path.relpath(f, ROOT).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def lib_sha256(path=None):
    path = path or os.path.join(ROOT, "imagestitch_amd", "lib", "libvfsms.so")
    try:
        return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]
    except OSError:
        return "none"


def build_id(lib_path=None):
    info = {}
    try:
        for tok in open(os.path.join(ROOT, "imagestitch_amd", "lib", "BUILD_INFO")).read().split():
            k, _, v = tok.partition("=")
            info[k] = v
    except OSError:
        pass
    return dict(src_sha256=src_sha256(), lib_sha256=lib_sha256(lib_path), head=info.get("head", "unknown"))


if __name__ == "__main__":
    print(" ".join("%s=%s" % kv for kv in sorted(build_id().items())))
