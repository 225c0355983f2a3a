"""Build libgava_hip.so (gfx950) in-tree with hipcc.  `python -m gava_clip_amd.build`.

The library is rebuilt when the CONTENT of csrc/ + include/gava_hip.h differs from what the existing .so was built from
(sha256 stored next to it in libgava_hip.so.srchash), not by mtimes: a prebuilt .so that travels to the GPU box with its
hash file is reused there, a stale one is never reused.  The ABI version the library reports (gava_abi_version) is the
first 31 bits of sha256(include/gava_hip.h), baked in with -DGAVA_ABI_HASH; gava_clip_amd.hip.load() refuses a library
whose version differs from the header in the tree.
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
HEADER = os.path.join(HERE, "..", "include", "gava_hip.h")
LIB = os.path.join(HERE, "libgava_hip.so")
HASHFILE = LIB + ".srchash"
SOURCES = ["gemm.hip", "attention.hip", "rowops.hip", "forward.hip", "preprocess.hip", "backward.hip", "attention_bwd.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def abi_hash(header=HEADER) -> int:
    """The ABI version: first 31 bits of sha256 of the header text (fits a non-negative C int)."""
    with open(header, "rb") as f:
        return int.from_bytes(hashlib.sha256(f.read()).digest()[:4], "big") >> 1


def source_hash() -> str:
    h = hashlib.sha256()
    h.update(" ".join(FLAGS).encode())
    for name in sorted(os.listdir(CSRC)):
        p = os.path.join(CSRC, name)
        if os.path.isfile(p) and (name in SOURCES or name.endswith(".h")):      # stray editor / backup files do not count
            h.update(name.encode() + b"\0")
            with open(p, "rb") as f:
                h.update(f.read())
    with open(HEADER, "rb") as f:
        h.update(f.read())
    return h.hexdigest()


def needs_build():
    if not os.path.exists(LIB) or not os.path.exists(HASHFILE):
        return True
    with open(HASHFILE) as f:
        return f.read().strip() != source_hash()


def build(force=False, verbose=True, lib=LIB, extra_flags=(), abi=None):
    """`lib`, `extra_flags`, `abi`: experiment / test builds (tools/ab_build.sh, tests/test_host_cpu.py) next to the product one."""
    product = lib == LIB and not extra_flags and abi is None
    if product and not force and not needs_build():
        return lib
    objs = []
    procs = []
    tag = "" if product else "_" + hashlib.sha256((lib + " ".join(extra_flags) + str(abi)).encode()).hexdigest()[:8]
    bdir = os.path.join(HERE, "build" + tag)
    os.makedirs(bdir, exist_ok=True)
    define = f"-DGAVA_ABI_HASH={abi_hash() if abi is None else abi}"
    for src in SOURCES:
        obj = os.path.join(bdir, src.replace(".hip", ".o"))
        cmd = [hipcc()] + FLAGS + [define] + list(extra_flags) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
        objs.append(obj)
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    if product:
        with open(HASHFILE, "w") as f:
            f.write(source_hash() + "\n")
    return lib


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
