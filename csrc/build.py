"""Build the native libraries in-tree for gfx950.

Produces, under ``pytorch_distributed_training_amd/_lib/``:

* ``libpdt_core.so``    - reducer core + loopback fabric (plain C++, no GPU runtime)
* ``libpdt_rccl.so``    - RCCL transport on a dedicated HIP stream (links amdhip64 + rccl)
* ``libpdt_kernels.so`` - every HIP kernel in ``csrc/kernels`` (``--offload-arch=gfx950``)

The libraries expose a C ABI and are loaded with ctypes, so the build needs no PyTorch
headers and a cold build takes well under a minute.  hipcc cross-compiles without a GPU.
Objects are rebuilt only when a source or header is newer.

``--sanitize address|thread|undefined`` builds, instead, the host-only stress driver of the reducer core and the
loopback fabric under that sanitizer (``build/san/reducer_stress_<kind>``); ``tests/test_sanitizers.py`` runs all three.
"""

from __future__ import annotations

import argparse
import concurrent.futures as cf
import os
import shutil
import subprocess
import sys
from pathlib import Path
from typing import List

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "csrc"
OUT = ROOT / "pytorch_distributed_training_amd" / "_lib"
OBJ = ROOT / "build" / "obj"
ROCM = Path(os.environ.get("ROCM_PATH", "/opt/rocm"))
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), str(ROCM / "bin" / "hipcc"), shutil.which("hipcc")):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found; set HIPCC or ROCM_PATH")


def _newer(target: Path, deps: List[Path]) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(d.stat().st_mtime > t for d in deps)


def _run(cmd: List[str]) -> None:
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + res.stdout + "\n")
        raise RuntimeError(f"build step failed: {cmd[0]} ... {cmd[-1]}")
    if res.stdout.strip() and os.environ.get("PDT_BUILD_VERBOSE"):
        sys.stderr.write(res.stdout)


def build_core(force: bool = False) -> Path:
    srcs = [CSRC / "reducer" / "reducer_core.cpp", CSRC / "reducer" / "loopback.cpp"]
    deps = srcs + [CSRC / "reducer" / "reducer_core.h"]
    out = OUT / "libpdt_core.so"
    if force or _newer(out, deps):
        cxx = os.environ.get("CXX", "g++")
        _run([cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wall", "-o", str(out)] + [str(s) for s in srcs])
    return out


SANITIZERS = ("address", "thread", "undefined")


def build_sanitized(kind: str, force: bool = False) -> Path:
    """Host-only sanitizer build of the reducer core + loopback fabric with their stress driver
    (``csrc/reducer/reducer_stress.cpp``): an executable under ``build/san/``.  GPU code is not involved, so the
    sanitizers run anywhere; run the result directly (``build/san/reducer_stress_thread``)."""
    if kind not in SANITIZERS:
        raise ValueError(f"--sanitize takes one of {SANITIZERS}")
    out_dir = ROOT / "build" / "san"
    out_dir.mkdir(parents=True, exist_ok=True)
    srcs = [CSRC / "reducer" / "reducer_core.cpp", CSRC / "reducer" / "loopback.cpp", CSRC / "reducer" / "reducer_stress.cpp"]
    deps = srcs + [CSRC / "reducer" / "reducer_core.h"]
    out = out_dir / f"reducer_stress_{kind}"
    if force or _newer(out, deps):
        cxx = os.environ.get("CXX", "g++")
        flags = [f"-fsanitize={kind}", "-fno-omit-frame-pointer", "-g", "-O1"]
        if kind == "undefined":
            flags.append("-fno-sanitize-recover=all")
        _run([cxx, "-std=c++17", "-pthread", "-Wall", *flags, "-o", str(out)] + [str(s) for s in srcs])
    return out


def build_rccl(force: bool = False) -> Path:
    src = CSRC / "reducer" / "reducer_rccl.cpp"
    deps = [src, CSRC / "reducer" / "reducer_core.h"]
    out = OUT / "libpdt_rccl.so"
    if force or _newer(out, deps):
        _run(
            [
                _hipcc(), "-O2", "-std=c++17", "-fPIC", "-shared", "-x", "hip", f"--offload-arch={ARCH}",
                "-D__HIP_PLATFORM_AMD__", f"-I{ROCM / 'include'}", "-Wall", "-o", str(out), str(src),
                f"-L{ROCM / 'lib'}", "-lamdhip64", "-lrccl",
            ]
        )
    return out


# Per-file code generation options.  The attention kernels post-process every MFMA result on the vector ALU
# (softmax, dS): with accumulators in AGPRs each value costs a v_accvgpr_read/write pair, so MFMA is asked
# to use VGPR destinations there.
EXTRA_FLAGS = {
    "attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
}

# Floating-point model per file.  -ffast-math tells the compiler that no NaN or infinity ever occurs
# (finite-math-only), which is acceptable where a non-finite value can only pass through (the MFMA / attention
# kernels: their arithmetic is MFMA instructions and explicit intrinsics) but not where the training loop relies on
# seeing one: the loss and the optimizer keep IEEE semantics for NaN / Inf and only allow contraction into fused
# multiply-adds, and so does the BatchNorm finalize kernel, which raises the non-finite flag --check-finite reads
# (under finite-math-only even a bit test of the exponent field is folded away: the compiler may assume that no float
# argument is an Inf or a NaN).  bn.hip's element-wise passes stay on the fast model - they run at the memory roof
# and lose 2 % to the NaN-aware forms of max().
PRECISE_MATH = {"loss.hip", "optim.hip", "bn_finalize.hip"}


def _math_flags(name: str) -> List[str]:
    return ["-ffp-contract=fast", "-fno-math-errno"] if name in PRECISE_MATH else ["-ffast-math"]


def _compile_kernel(src: Path, headers: List[Path], force: bool) -> Path:
    obj = OBJ / (src.stem + ".o")
    if force or _newer(obj, [src] + headers + [Path(__file__)]):
        _run(
            [
                _hipcc(), "-O3", "-std=c++17", "-fPIC", "-c", f"--offload-arch={ARCH}", *_math_flags(src.name),
                *EXTRA_FLAGS.get(src.name, []),
                "-Wall", "-Wno-unused-function", f"-I{CSRC / 'kernels'}", "-o", str(obj), str(src),
            ]
        )
    return obj


def build_kernels(force: bool = False, jobs: int = 0) -> Path:
    kdir = CSRC / "kernels"
    srcs = sorted(kdir.glob("*.hip"))
    headers = sorted(kdir.glob("*.h")) + sorted(kdir.glob("*.hpp"))
    out = OUT / "libpdt_kernels.so"
    if not srcs:
        return out
    jobs = jobs or min(8, int(os.environ.get("MAX_JOBS", "8")), len(srcs))
    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(lambda s: _compile_kernel(s, headers, force), srcs))
    if force or _newer(out, objs):
        _run([_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", str(out)] + [str(o) for o in objs]
             + [f"-L{ROCM / 'lib'}", "-lamdhip64"])
    return out


def build_all(force: bool = False) -> List[Path]:
    OUT.mkdir(parents=True, exist_ok=True)
    OBJ.mkdir(parents=True, exist_ok=True)
    return [build_core(force), build_rccl(force), build_kernels(force)]


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--sanitize", choices=SANITIZERS, default=None,
                    help="build the host-only sanitizer stress driver of the reducer core instead of the libraries")
    args = ap.parse_args()
    if args.sanitize:
        print(build_sanitized(args.sanitize, args.force))
    else:
        for p in build_all(args.force):
            print(p)
