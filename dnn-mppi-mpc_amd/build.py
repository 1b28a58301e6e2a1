"""Builds libmppi_hip.so in-tree with hipcc for gfx950 (driven by csrc/Makefile)."""
from __future__ import annotations

import os
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "lib", "libmppi_hip.so")


def lib_is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(os.path.dirname(PKG), "include", "mppi_hip.h"))
    return any(os.path.getmtime(s) > t for s in srcs)


def source_id() -> str:
    """Identity of the kernel sources (sha256 over csrc/*.hip, csrc/*.h, include/mppi_hip.h, 16 hex digits): profile
    summaries under profiles/ carry it, and bench.py quotes a counter-derived figure only from a profile of THIS build."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    files.append(os.path.join(os.path.dirname(PKG), "include", "mppi_hip.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def build(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 -shared ... -> lib/libmppi_hip.so; returns its path."""
    if force or lib_is_stale():
        cmd = ["make", "-j4", "-C", CSRC] + (["-B"] if force else [])
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if verbose or res.returncode != 0:
            print(res.stdout)
        if res.returncode != 0:
            raise RuntimeError("building libmppi_hip.so failed (hipcc, gfx950)")
    return LIB
