"""Builds libdvo_amd.so (HIP kernels + host driver + C ABI) in-tree with hipcc for gfx950.

The binary carries a BUILD ID -- a hash of every source file, header and compiler flag it was made from
(`dvo_amd_build_id()`).  `needs_build()` / `capi.lib()` compare it with the hash of the sources next to it: a library that was
not built from exactly these sources is rebuilt (where hipcc exists) or refused (where it does not), so a test or a bench can
never run an edited source tree against a stale kernel.  `python -m dvo_slam_amd._build --print-id` prints the id of the
sources (the Makefile passes it to hipcc the same way)."""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("DVO_AMD_LIB") or os.path.join(_HERE, "libdvo_amd.so")
SOURCES = ["dvo_kernels.hip", "dvo_pyramid.cpp", "dvo_tracker.cpp", "dvo_sharded.cpp", "dvo_probes.cpp", "dvo_validator.cpp",
           "dvo_frontend.cpp", "dvo_tum.cpp"]
HEADERS = ["dvo_types.h", "dvo_internal.h", "se3.h", os.path.join("..", "..", "include", "dvo_amd.h"),
           os.path.join("..", "..", "include", "dvo_amd_debug.h")]
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off",          # the warp/residual stage must round every product and sum separately
    "-fno-fast-math",
    "-fno-slp-vectorize",         # packed fp32 ops buy nothing on gfx950 here and cost shuffle moves (measured: +7 %)
    "-Wall", "-Wno-unused-function",
]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP library cannot be built (there is no CPU fallback)")


def source_id() -> str:
    """sha256 over the compiler flags and the bytes of every source and header, in a fixed order (first 16 hex digits)"""
    h = hashlib.sha256()
    h.update(" ".join(FLAGS).encode())
    for name in SOURCES + HEADERS:
        h.update(b"\0" + os.path.basename(name).encode() + b"\0")
        with open(os.path.join(CSRC, name), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


_MARKER = b"DVO_AMD_BUILD_ID="


def library_id(path: str | None = None) -> str | None:
    """the build id a built library carries (None: no library there, or one from before build ids existed).  Read from the file
    -- the id sits behind a marker string in the binary -- not through dlopen: a stale library loaded into this process could
    not be replaced by the rebuilt one under the same path."""
    path = path or LIB_PATH
    if not os.path.exists(path):
        return None
    with open(path, "rb") as fh:
        blob = fh.read()
    at = blob.find(_MARKER)
    if at < 0:
        return None
    end = blob.find(b";", at)
    return blob[at + len(_MARKER):end].decode(errors="replace") if 0 < end - at < 100 else None


def needs_build() -> bool:
    """True when the library is missing or was not built from exactly the sources next to it.  A library given through
    DVO_AMD_LIB (tuning variants, A/B baselines) is taken as it is."""
    if os.environ.get("DVO_AMD_LIB"):
        return not os.path.exists(LIB_PATH)
    return library_id() != source_id()


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile for gfx950 (cross-compiles without a GPU).  Returns the path of the shared library."""
    if not force and not needs_build():
        return LIB_PATH
    cmd = [_hipcc()] + FLAGS + [f'-DDVO_AMD_BUILD_ID="{source_id()}"', "-x", "hip"] + \
        [os.path.join(CSRC, s) for s in SOURCES] + ["-lz", "-o", LIB_PATH + ".tmp"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
        print(res.stderr)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stderr[-4000:])
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    if "--print-id" in sys.argv:
        print(source_id())
    elif "--print-flags" in sys.argv:
        print(" ".join(FLAGS))
    else:
        print(build(force=True, verbose=True))
