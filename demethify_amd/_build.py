"""In-tree build of libdemethify_hip.so for gfx950 (hipcc cross-compiles without a GPU).

    python -m demethify_amd._build [--force]

The shared object is written next to this file so that it travels with the source tree.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
LIB_PATH = PKG_DIR / "libdemethify_hip.so"
OBJ_DIR = CSRC / "build"

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
EXPERIMENT = ["-DDMF_EXPERIMENT"] if os.environ.get("DMF_EXPERIMENT") == "1" else []
CXXFLAGS = [*EXPERIMENT, "-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result", "-Wno-pass-failed"]


def _sources():
    return sorted(CSRC.glob("*.hip"))


def _headers():
    return sorted(CSRC.glob("*.h")) + [PKG_DIR.parent / "include" / "demethify_hip.h"]


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(d.stat().st_mtime > t for d in deps)


def _compile(src: Path, force: bool) -> Path:
    obj = OBJ_DIR / (src.stem + ".o")
    if force or _stale(obj, [src] + _headers()):
        cmd = [HIPCC, *CXXFLAGS, "-c", str(src), "-o", str(obj)]
        proc = subprocess.run(cmd, capture_output=True, text=True)
        if proc.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src.name}:\n{proc.stdout}\n{proc.stderr}")
        if proc.stderr.strip():
            sys.stderr.write(proc.stderr)
    return obj


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile every HIP translation unit and link the C-ABI shared library."""
    OBJ_DIR.mkdir(parents=True, exist_ok=True)
    srcs = _sources()
    if not srcs:
        raise RuntimeError(f"no HIP sources under {CSRC}")
    with ThreadPoolExecutor(max_workers=min(4, len(srcs))) as pool:
        objs = list(pool.map(lambda s: _compile(s, force), srcs))
    if force or _stale(LIB_PATH, objs):
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", str(LIB_PATH), *map(str, objs)]
        proc = subprocess.run(cmd, capture_output=True, text=True)
        if proc.returncode != 0:
            raise RuntimeError(f"link failed:\n{proc.stdout}\n{proc.stderr}")
    if verbose:
        print(f"built {LIB_PATH}")
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
