"""Builds libofdm_mi355x.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import concurrent.futures as cf
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libofdm_mi355x.so")
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CFLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
          "-Wno-logical-op-parentheses", "-ffp-contract=fast",
          # packed-f32 (v_pk_*) forms cost as much as two plain VALU ops and add register shuffles: measured
          # +10 % on the fused chain with SLP vectorisation off (profiles/round1/slp_ab.txt)
          "-fno-slp-vectorize", "-Wno-pass-failed"]


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    hs.append(os.path.join(os.path.dirname(HERE), "include", "ofdm_mi355x.h"))
    return sorted(hs)


def _digest(paths, extra=""):
    h = hashlib.sha256(extra.encode())
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _compile(src, hdr_digest, verbose, extra=()):
    obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ("_diag" if extra else "") + ".o")
    stamp = obj + ".sha"
    want = _digest([src], hdr_digest + " ".join([*CFLAGS, *extra]))
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == want:
        return obj, False
    cmd = [HIPCC, *CFLAGS, *extra, "-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if verbose and r.stderr.strip():
        print(r.stderr, file=sys.stderr)
    with open(stamp, "w") as f:
        f.write(want)
    return obj, True


def build_library(verbose: bool = False, force: bool = False, diag: bool = False) -> str:
    """diag=True builds libofdm_mi355x_diag.so with -DOFDM_DIAG: the work-skipping ablation switches of tools/ (never
    loaded by the package, bench.py or the tests; the shipped library does not contain them)."""
    os.makedirs(OBJ, exist_ok=True)
    extra = ("-DOFDM_DIAG",) if diag else ()
    lib = LIB[:-3] + "_diag.so" if diag else LIB
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    hdr = _digest(_headers())
    srcs = _sources()
    with cf.ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        res = list(ex.map(lambda s: _compile(s, hdr, verbose, extra), srcs))
    objs = [o for o, _ in res]
    changed = any(c for _, c in res)
    if changed or not os.path.exists(lib):
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", lib]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return lib


if __name__ == "__main__":
    print(build_library(verbose=True, force="--force" in sys.argv, diag="--diag" in sys.argv))
