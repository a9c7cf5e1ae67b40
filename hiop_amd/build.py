"""Build the gfx950 shared library `hiop_amd/lib/libhiopamd.so` in-tree with hipcc.

One translation unit per kernel family (compiled in parallel), linked against RCCL.  There is no
CPU build and no fallback: if hipcc is missing this raises.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent
CSRC = ROOT / "csrc"

# HIOPAMD_BUILD_VARIANT selects a TEST build next to the shipped one (own object and library directories; hiop_amd/_lib.py loads the same
# variant).  "poison": every device allocation of the library is filled with 0xFF bytes (NaN / -1) — the GPU suite under it finds reads of
# memory nobody wrote (csrc/common.hpp, HIOPAMD_POISON_ALLOC).
VARIANTS = {"": [], "poison": ["-DHIOPAMD_POISON_ALLOC"]}
VARIANT = os.environ.get("HIOPAMD_BUILD_VARIANT", "")
if VARIANT not in VARIANTS:
    raise RuntimeError(f"unknown HIOPAMD_BUILD_VARIANT {VARIANT!r} (known: {sorted(VARIANTS)})")
LIBDIR = ROOT / ("lib" if not VARIANT else f"lib_{VARIANT}")
LIB = LIBDIR / "libhiopamd.so"
STAMP = LIBDIR / "libhiopamd.stamp"
OBJDIR = ROOT / ("build" if not VARIANT else f"build_{VARIANT}")

SOURCES = [
    "context.hip",
    "vector_kernels.hip",
    "dense_kernels.hip",
    "sparse_kernels.hip",
    "sparse_assembly.hip",
    "csr_condensed.hip",
    "arrow_ldl.hip",
    "sparse_ldl.hip",
    "kkt_sparse.hip",
    "gram.hip",
    "ldlt.hip",
    "ldlt_bk.hip",
    "small_solvers.hip",
    "kkt_mds.hip",
    "lowrank.hip",
    "kkt_xycyd.hip",
    "io.hip",
    "krylov.hip",
    "example_mds.hip",
    "example_dense.hip",
    "ipm_mds.hip",
]

# per-file device-code options.  ldlt.hip / gram.hip: the SI load/store optimizer fuses two ds_read_b64 into one
# ds_read2_b64, which costs 16 LDS cycles on gfx950 instead of 2 + 2 (MI355X_MICROARCH.md, LDS table) — the MFMA operand
# reads of the trailing update went through it; "-load-store-opt" keeps them as ds_read_b64.
# "-amdgpu-mfma-vgpr-form": the chain kernel needs more than 256 registers per lane, and left to itself the allocator then
# puts MFMA accumulators into AGPRs — the AGPR-accumulator form of v_mfma_f64_16x16x4_f64 runs at half rate
# (profiles/r02_probes/README.md) and every read of a result costs two v_accvgpr_read; the serial 16 x 16 factor is built
# from dependent MFMAs.
EXTRA_FLAGS = {
    "ldlt.hip": ["-Xclang", "-target-feature", "-Xclang", "-load-store-opt", "-mllvm", "-amdgpu-mfma-vgpr-form"],
}

ARCH = "gfx950"
CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wno-unused-result", "-ffp-contract=on"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: hiop_amd has no CPU build")
    return exe


def _deps() -> list[Path]:
    hdrs = list(CSRC.glob("*.hpp")) + list(CSRC.glob("*.inc")) + list((ROOT.parent / "include").glob("*.h"))
    return hdrs


def _fingerprint() -> str:
    """Hash of everything the library is made of: sources, headers, this file (flags).  Written next to the library; a library whose
    stamp matches is up to date wherever it was built — the objects do not travel to the GPU box (.gpurunignore) and file times need
    not survive the copy, so neither can decide that."""
    import hashlib
    h = hashlib.sha256()
    for p in sorted([CSRC / s for s in SOURCES] + _deps() + [Path(__file__)]):
        if p.exists():
            h.update(p.name.encode())
            h.update(p.read_bytes())
    h.update(VARIANT.encode())
    return h.hexdigest()


def _needs(obj: Path, src: Path) -> bool:
    if not obj.exists():
        return True
    t = obj.stat().st_mtime
    return any(p.stat().st_mtime > t for p in [src, Path(__file__), *_deps()])


def build(force: bool = False, verbose: bool = True) -> Path:
    fp = _fingerprint()
    if not force and LIB.exists() and STAMP.exists() and STAMP.read_text().strip() == fp:
        return LIB
    hipcc = _hipcc()
    OBJDIR.mkdir(exist_ok=True)
    LIBDIR.mkdir(exist_ok=True)
    srcs = [CSRC / s for s in SOURCES if (CSRC / s).exists()]
    jobs = []
    for src in srcs:
        obj = OBJDIR / (src.stem + ".o")
        if force or _needs(obj, src):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [hipcc, *CXXFLAGS, *VARIANTS[VARIANT], *EXTRA_FLAGS.get(src.name, []), "-c", str(src), "-o", str(obj)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stderr[-4000:]}")
        return src.name

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for name in ex.map(cc, jobs):
                if verbose:
                    print(f"[hiop_amd.build] compiled {name}", file=sys.stderr)
    objs = [OBJDIR / (s.stem + ".o") for s in srcs]
    if force or jobs or not LIB.exists():
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *map(str, objs), "-L/opt/rocm/lib", "-lrccl", "-lrocprofiler-sdk-roctx",
               "-Wl,-rpath,/opt/rocm/lib", "-o", str(LIB)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
        if verbose:
            print(f"[hiop_amd.build] linked {LIB}", file=sys.stderr)
    STAMP.write_text(fp + "\n")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
