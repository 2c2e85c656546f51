"""In-tree build of the native libraries.

Two shared objects are produced under ``nanofed_amd/_lib``:

``libnanofed_host.so``
    C++ host code (OpenSSL-backed crypto, tensor/JSON codec), built with g++.
``libnanofed_hip.so``
    all HIP kernels for gfx950 plus their C launchers, built with hipcc.

Both expose a plain C ABI and are loaded with ctypes, so neither build
touches the PyTorch headers and a full rebuild takes well under a minute.
Sources are recompiled only when they (or a header next to them) are newer
than their object file.  ``python -m nanofed_amd.native_build`` builds both.
"""

from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "csrc"
LIB_DIR = Path(__file__).resolve().parent / "_lib"
OBJ_DIR = ROOT / "build" / "obj"

HOST_LIB = LIB_DIR / "libnanofed_host.so"
HIP_LIB = LIB_DIR / "libnanofed_hip.so"

GPU_ARCH = "gfx950"

HIPCC_FLAGS = [
    f"--offload-arch={GPU_ARCH}",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-ffp-contract=fast",
    # keep hipcc from forming packed-fp32 VALU ops
    "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops",
    "-Wno-unused-command-line-argument",
]

HOST_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-Wall", "-pthread"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (Path(cand).exists() or cand == "hipcc"):
            return cand
    return "hipcc"


def _newest_header(src_dir: Path) -> float:
    stamps = [p.stat().st_mtime for p in src_dir.glob("*.h")]
    stamps += [p.stat().st_mtime for p in src_dir.glob("*.hpp")]
    # headers shared by host and device code
    stamps += [p.stat().st_mtime for p in (CSRC / "common").glob("*.h")]
    return max(stamps, default=0.0)


def _stale(src: Path, obj: Path, header_stamp: float) -> bool:
    if not obj.exists():
        return True
    t = obj.stat().st_mtime
    return src.stat().st_mtime > t or header_stamp > t


def _run(cmd: list[str]) -> None:
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + proc.stdout + proc.stderr)
        raise RuntimeError(f"native build failed: {cmd[0]} ... {cmd[-1]}")
    # surface warnings other than the harmless unknown-feature note
    for line in proc.stderr.splitlines():
        if "warning" in line and "packed-fp32-ops" not in line:
            sys.stderr.write(line + "\n")


def _compile_all(
    compiler: str, flags: list[str], sources: list[Path], tag: str, jobs: int
) -> tuple[list[Path], bool]:
    OBJ_DIR.mkdir(parents=True, exist_ok=True)
    objs, todo = [], []
    for src in sources:
        obj = OBJ_DIR / f"{tag}_{src.stem}.o"
        objs.append(obj)
        if _stale(src, obj, _newest_header(src.parent)):
            todo.append((src, obj))
    with ThreadPoolExecutor(max_workers=max(1, jobs)) as pool:
        list(pool.map(lambda so: _produce(so[1], [compiler, *flags, "-c", str(so[0])]), todo))
    return objs, bool(todo)


def _produce(target: Path, cmd: list[str]) -> None:
    """Run ``cmd -o <temporary>`` and move the result into place: a process that loads or
    links ``target`` meanwhile (a test worker, a second build) sees the old file or the new
    one, never half of one."""
    tmp = target.with_name(f".{target.name}.{os.getpid()}.tmp")
    try:
        _run([*cmd, "-o", str(tmp)])
        os.replace(tmp, target)
    finally:
        if tmp.exists():
            tmp.unlink()


def build_host(force: bool = False, jobs: int = 4) -> Path:
    sources = sorted((CSRC / "host").glob("*.cpp"))
    LIB_DIR.mkdir(parents=True, exist_ok=True)
    if force:
        for o in OBJ_DIR.glob("host_*.o"):
            o.unlink()
    objs, changed = _compile_all("g++", HOST_FLAGS, sources, "host", jobs)
    if changed or not HOST_LIB.exists():
        _produce(HOST_LIB, ["g++", "-shared", *map(str, objs), "-lcrypto", "-pthread"])
    return HOST_LIB


def build_hip(force: bool = False, jobs: int = 4) -> Path:
    sources = sorted((CSRC / "hip").glob("*.hip"))
    LIB_DIR.mkdir(parents=True, exist_ok=True)
    if force:
        for o in OBJ_DIR.glob("hip_*.o"):
            o.unlink()
    hipcc = _hipcc()
    objs, changed = _compile_all(hipcc, HIPCC_FLAGS, sources, "hip", jobs)
    if changed or not HIP_LIB.exists():
        _produce(HIP_LIB, [hipcc, f"--offload-arch={GPU_ARCH}", "-shared", "-fPIC", *map(str, objs)])
    return HIP_LIB


def build_all(force: bool = False, jobs: int = 4) -> dict[str, Path]:
    return {"host": build_host(force, jobs), "hip": build_hip(force, jobs)}


if __name__ == "__main__":
    built = build_all(force="--force" in sys.argv)
    for name, path in built.items():
        print(f"{name}: {path} ({path.stat().st_size} bytes)")
