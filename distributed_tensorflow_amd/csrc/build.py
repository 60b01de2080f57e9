"""In-tree build of the native libraries.

Two shared objects are produced under ``distributed_tensorflow_amd/lib``:

``libdtf_kernels.so``
    every ``csrc/kernels/*.hip`` file, compiled by ``hipcc`` for gfx950 only
    (device code) and exported through a C ABI that takes raw device
    pointers and a ``hipStream_t``.

``libdtf_runtime.so``
    the host runtime in ``csrc/runtime/*.cpp`` (framed TCP transport,
    parameter-server core, data loader, RCCL communicator, IPC peer memory),
    compiled by ``hipcc`` as host code and linked against ``libamdhip64`` and
    ``librccl``.

Objects are rebuilt only when their source (or a header) is newer, so calling
:func:`build_all` repeatedly is cheap. A build holds an exclusive file lock (several ranks that find the
library missing build it once, one after the other finding it up to date) and links to a temporary name
that is renamed into place, so no process ever maps a half-written library. Run ``python -m
distributed_tensorflow_amd.csrc.build`` to build from the shell.
"""
from __future__ import annotations

import concurrent.futures as cf
import contextlib
import fcntl
import os
import shutil
import subprocess
import sys
from pathlib import Path

CSRC = Path(__file__).resolve().parent
PKG = CSRC.parent
LIB_DIR = PKG / "lib"
OBJ_DIR = CSRC / "build"
ARCH = "gfx950"
ROCM = Path(os.environ.get("ROCM_PATH", "/opt/rocm"))


def _hipcc() -> str:
    exe = shutil.which("hipcc") or str(ROCM / "bin" / "hipcc")
    if not Path(exe).exists():
        raise RuntimeError("hipcc not found; a ROCm toolchain is required to build the native libraries")
    return exe


def _newer(src: Path, deps: list[Path], out: Path) -> bool:
    if not out.exists():
        return True
    t = out.stat().st_mtime
    return any(d.stat().st_mtime > t for d in [src, *deps])


def _run(cmd: list[str]) -> None:
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("command failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
    if r.stderr.strip() and os.environ.get("DTF_BUILD_VERBOSE"):
        sys.stderr.write(r.stderr)


def _compile_objects(srcs: list[Path], headers: list[Path], flags: list[str], tag: str, jobs: int) -> list[Path]:
    OBJ_DIR.mkdir(parents=True, exist_ok=True)
    hipcc = _hipcc()
    objs, todo = [], []
    for s in srcs:
        o = OBJ_DIR / f"{tag}_{s.stem}.o"
        objs.append(o)
        if _newer(s, headers + [Path(__file__)], o):
            todo.append([hipcc, *flags, "-c", str(s), "-o", str(o)])
    if todo:
        with cf.ThreadPoolExecutor(max_workers=max(1, jobs)) as ex:
            list(ex.map(_run, todo))
    return objs


@contextlib.contextmanager
def _build_lock():
    OBJ_DIR.mkdir(parents=True, exist_ok=True)
    with open(OBJ_DIR / ".lock", "w") as f:
        fcntl.flock(f, fcntl.LOCK_EX)
        try:
            yield
        finally:
            fcntl.flock(f, fcntl.LOCK_UN)


def _link(cmd_head: list[str], out: Path, tail: list[str]) -> None:
    tmp = out.with_name(out.name + f".tmp{os.getpid()}")
    _run([*cmd_head, "-o", str(tmp), *tail])
    os.replace(tmp, out)


def build_kernels(jobs: int = 8, force: bool = False) -> Path:
    with _build_lock():
        return _build_kernels(jobs, force)


def build_runtime(jobs: int = 8, force: bool = False) -> Path:
    with _build_lock():
        return _build_runtime(jobs, force)


def _build_kernels(jobs: int = 8, force: bool = False) -> Path:
    kdir = CSRC / "kernels"
    srcs = sorted(kdir.glob("*.hip"))
    headers = sorted(kdir.glob("*.h"))
    out = LIB_DIR / "libdtf_kernels.so"
    if force and out.exists():
        out.unlink()
    flags = ["-x", "hip", f"--offload-arch={ARCH}", "-O3", "-fPIC", "-std=c++17", "-fvisibility=hidden",
             "-ffast-math", "-fno-finite-math-only", "-munsafe-fp-atomics", f"-I{kdir}"]
    objs = _compile_objects(srcs, headers, flags, "k", jobs)
    LIB_DIR.mkdir(parents=True, exist_ok=True)
    if force or any(_newer(o, [], out) for o in objs) or _manifest_changed("k", objs):
        _link([_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}"], out, list(map(str, objs)))
    return out


def _manifest_changed(tag: str, objs: list[Path]) -> bool:
    """True (and records the new list) when the set of objects that make up a library changed, e.g. a source
    file was removed: no object is then newer than the library, but it must be linked again."""
    mf = OBJ_DIR / f"{tag}_manifest.txt"
    want = "\n".join(sorted(o.name for o in objs))
    if mf.exists() and mf.read_text() == want:
        return False
    mf.write_text(want)
    return True


def _build_runtime(jobs: int = 8, force: bool = False) -> Path:
    rdir = CSRC / "runtime"
    srcs = sorted(rdir.glob("*.cpp"))
    headers = sorted(rdir.glob("*.h"))
    out = LIB_DIR / "libdtf_runtime.so"
    if force and out.exists():
        out.unlink()
    flags = ["-O2", "-fPIC", "-std=c++17", "-fvisibility=hidden", "-pthread", "-D__HIP_PLATFORM_AMD__",
             f"-I{rdir}", f"-I{ROCM / 'include'}"]
    extra = os.environ.get("DTF_RUNTIME_CXXFLAGS", "").split()
    objs = _compile_objects(srcs, headers, flags + extra, "r", jobs)
    LIB_DIR.mkdir(parents=True, exist_ok=True)
    if force or any(_newer(o, [], out) for o in objs):
        _link([_hipcc(), "-shared", "-fPIC", "-pthread"], out,
              [*map(str, objs), *extra, f"-L{ROCM / 'lib'}", "-lamdhip64", "-lrccl", f"-Wl,-rpath,{ROCM / 'lib'}"])
    return out


def build_all(jobs: int = 8, force: bool = False) -> dict[str, Path]:
    return {"kernels": build_kernels(jobs, force), "runtime": build_runtime(jobs, force)}


if __name__ == "__main__":
    res = build_all(force="--force" in sys.argv)
    for k, v in res.items():
        print(f"{k}: {v} ({v.stat().st_size} bytes)")
