"""Build libako.so (HIP kernels for gfx950 + C-ABI + host C drivers) in-tree.

    python -m ako_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU; the resulting ako_amd/libako.so travels to the GPU
box with the repository snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libako_experimental.so" if os.environ.get("AKO_BUILD_EXPERIMENTAL") == "1" else "libako.so")
OBJ = os.path.join(HERE, "csrc", "build_experimental" if os.environ.get("AKO_BUILD_EXPERIMENTAL") == "1" else "build")

# translation units of device code and the headers each one includes (they build in parallel: ako_plan.hip alone takes minutes)
HIP_SOURCES = {
    "ako_plan.hip": ["ako_kernels.hip.h", "ako_stream.hip.h", "ako_tail.hip.h", "ako_tail_params.h", "ako_kagari.hip.h",
                     "ako_requant.hip.h", "ako_fused.h", "ako_u8.h"],
    "ako_u8_rgba.hip": ["ako_kernels.hip.h", "ako_stream.hip.h", "ako_u8_lean.hip.h", "ako_u8.h", "ako_u8_tu.hip.h"],
    "ako_u8_rgb.hip": ["ako_kernels.hip.h", "ako_stream.hip.h", "ako_u8_lean.hip.h", "ako_u8.h", "ako_u8_tu.hip.h"],
    "ako_u8_gray.hip": ["ako_kernels.hip.h", "ako_stream.hip.h", "ako_u8_lean.hip.h", "ako_u8_gray.hip.h", "ako_u8.h"],
    "ako_copy.hip": [],
}
# The routes that lost their measurements (levels 0 + 1 in one workgroup walk, AKO_HIP_FUSE2; level 0 in column groups with
# whole-line stores, AKO_HIP_GROUP) stay in the source and are parity-tested, but only a build with AKO_BUILD_EXPERIMENTAL=1
# holds them (-DAKO_EXPERIMENTAL in every unit; objects and library get their own names so that both builds can coexist).
EXPERIMENTAL = os.environ.get("AKO_BUILD_EXPERIMENTAL") == "1"
if EXPERIMENTAL:
    HIP_SOURCES["ako_u8_group.hip"] = ["ako_kernels.hip.h", "ako_stream.hip.h", "ako_u8.h"]
    HIP_SOURCES["ako_fused.hip"] = ["ako_kernels.hip.h", "ako_stream.hip.h", "ako_fused.h", "ako_fused.hip.h"]
C_SOURCES = ["host/ako_quant.c", "host/ako_head.c", "host/ako_misc.c", "host/ako_kagari.c", "host/ako_codec.c",
             "host/ako_synth.c", "host/ako_batch.c"]
C_HEADERS = ["host/ako_host.h", "../../include/ako.h", "../../include/ako_hip.h"]

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd: list[str]) -> None:
    print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)


def build(force: bool = False, extra_hip_flags: list[str] | None = None) -> str:
    os.makedirs(OBJ, exist_ok=True)
    objs = []
    jobs = []
    for src, headers in HIP_SOURCES.items():
        o = os.path.join(OBJ, os.path.basename(src) + ".o")
        hip_deps = [os.path.join(CSRC, h) for h in headers + C_HEADERS]
        asm = os.path.join(OBJ, os.path.splitext(os.path.basename(src))[0] + f"-hip-amdgcn-amd-amdhsa-{ARCH}.s")
        if force or _stale(o, [os.path.join(CSRC, src)] + hip_deps) or not os.path.exists(asm):
            # -fno-slp-vectorize: keeps hipcc from fusing scalar f32 adds into v_pk_add_f32, which costs
            # about two plain adds on gfx950 (MI355X_MICROARCH.md, cycle constants) and needs register pairs
            cmd = ([HIPCC, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
                    "-fno-slp-vectorize", "-Wall", "-Wno-unused-function",
                    # keeps the unit's device assembly beside the object (build/<unit>-hip-amdgcn-amd-amdhsa-gfx950.s):
                    # scripts/isa_lint.py and tests/test_isa_lint.py read the hot loops out of it
                    "-save-temps=obj"] + (["-DAKO_EXPERIMENTAL"] if EXPERIMENTAL else []) + (extra_hip_flags or []) +
                   os.environ.get("AKO_HIPCC_EXTRA", "").split() +  # experiments only
                   ["-c", os.path.join(CSRC, src), "-o", o])
            print(" ".join(cmd), flush=True)
            jobs.append((cmd, subprocess.Popen(cmd), o, time.time()))
        objs.append(o)
    failed = None
    for cmd, job, o, t0 in jobs:
        if job.wait() != 0:
            failed = failed or subprocess.CalledProcessError(job.returncode, cmd)
        else:
            os.utime(o, (t0, t0))  # dated by the START of its build: a source edited while it compiled is newer
    if failed:
        raise failed
    c_deps = [os.path.join(CSRC, h) for h in C_HEADERS]
    for src in C_SOURCES:
        o = os.path.join(OBJ, os.path.basename(src) + ".o")
        if force or _stale(o, [os.path.join(CSRC, src)] + c_deps):
            _run(["gcc", "-O2", "-std=gnu11", "-pthread", "-fPIC", "-fvisibility=hidden", "-ffp-contract=off", "-Wall", "-Wextra",
                  "-c", os.path.join(CSRC, src), "-o", o])
        objs.append(o)
    if force or _stale(OUT, objs):
        _run([HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", OUT] + objs + ["-lm", "-lpthread"])
    return OUT


TOOLS = os.path.join(os.path.dirname(HERE), "tools")
TOOL_BIN = os.path.join(TOOLS, "bin")


def build_tools(force: bool = False) -> str:
    """akoenc / akodec (link ako_amd/libako.so) and pngcheck (no GPU code) into tools/bin/."""
    build()
    os.makedirs(TOOL_BIN, exist_ok=True)
    inc = os.path.join(os.path.dirname(HERE), "include")
    common = [os.path.join(TOOLS, "cli_common.hpp"), os.path.join(inc, "ako.h")]
    for name, with_lib in (("akoenc", True), ("akodec", True), ("pngcheck", False)):
        src, exe = os.path.join(TOOLS, name + ".cpp"), os.path.join(TOOL_BIN, name)
        if force or _stale(exe, [src] + common + ([OUT] if with_lib else [])):
            cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", f"-I{inc}", f"-I{TOOLS}", src, "-o", exe]
            if with_lib:
                cmd += [f"-L{HERE}", "-lako", "-Wl,-rpath,$ORIGIN/../../ako_amd", "-Wl,-rpath-link,/opt/rocm/lib"]
            _run(cmd + ["-lz"])
    return TOOL_BIN


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    if "--tools" in sys.argv:
        build_tools(force="--force" in sys.argv)
