"""Build the gfx950 shared library in-tree: uvltrack_amd/libuvltrack_hip.so.

    python -m uvltrack_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU; the .so is git-ignored but travels to the GPU box.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libuvltrack_hip.so")
SOURCES = ["gemm.hip", "gemm_fin.hip", "gemm_dr.hip", "attention.hip", "rowops.hip", "head_fin.hip", "prompter.hip", "preprocess.hip", "uvl_api.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result",
         "-mllvm", "-amdgpu-mfma-vgpr-form=1",      # accumulators stay in VGPRs (unified file on gfx950): no v_accvgpr moves around the softmax
         # a*b+c fuses where the SOURCE expression says so, not wherever the optimiser finds a multiply next to an add: the same
         # device function gives the same bits in every kernel it is inlined into (paired / stand-alone launch forms)
         "-ffp-contract=on",
         "-I", INCLUDE, "-I", CSRC]

# translation units whose kernels keep their MFMA accumulators in AGPRs (gemm_dr.hip: the K loop owns the 128 architectural VGPRs)
AGPR_FORM = {"gemm_dr.hip"}


def _flags_for(src):
    """FLAGS, without the `-mllvm <option>` PAIRS an AGPR-form translation unit must not get (dropped as pairs: a lone -mllvm would swallow
    the next argument)."""
    if src not in AGPR_FORM:
        return list(FLAGS)
    out, i = [], 0
    while i < len(FLAGS):
        if FLAGS[i] == "-mllvm" and i + 1 < len(FLAGS) and FLAGS[i + 1] in AGPR_FORM_DROP_OPTIONS:
            i += 2
            continue
        out.append(FLAGS[i])
        i += 1
    return out


AGPR_FORM_DROP_OPTIONS = {"-amdgpu-mfma-vgpr-form=1"}

# The kernels pin their instruction order with sched_barrier and count s_waitcnt by hand around inline-asm LDS reads / LDS-DMA
# (attention.hip::attn_w64_kernel, gemm.hip::gemm_pipe_tile), and two K loops are generated assembly with counted waits (attn_p64_asm.inc,
# gemm_dr_asm.inc): correct for THIS compiler's code generation -- a wrong count is a silent data race, not a crash.  So the toolchain is
# part of the build's identity: build() REFUSES another hipcc unless UVL_ALLOW_UNTESTED_HIPCC=1 is set, the version string is compiled into
# the library (uvl_build_toolchain()), and _native.load() refuses a library stamped with another one (same override).  After a toolchain
# change re-run `pytest tests -m gpu -k "forced or tile_forms or direct_to_register or generated"` and update TESTED_HIPCC.
TESTED_HIPCC = "HIP version: 7.2.26015-fc0010cf6a"
OVERRIDE_ENV = "UVL_ALLOW_UNTESTED_HIPCC"


def same_toolchain(stamp: str, tested: str = None) -> bool:
    """The tested stamp, or the same RELEASE with another build hash ("HIP version: 7.2.26015-<hash>": a repackaged build of the same
    compiler sources generates the same code; a different version number does not count)."""
    tested = TESTED_HIPCC if tested is None else tested
    if not stamp:
        return False
    return stamp == tested or ("-" in tested and stamp.rsplit("-", 1)[0] == tested.rsplit("-", 1)[0] and stamp.startswith("HIP version: "))


def only_hash_differs(stamp: str, tested: str = None) -> bool:
    """same_toolchain() accepted the stamp, but its build hash is not the tested one: the suffix is a source commit, so strictly this is another source tree
    of the same release.  Accepted (repackaged builds are common), but said out loud: callers warn."""
    tested = TESTED_HIPCC if tested is None else tested
    return same_toolchain(stamp, tested) and stamp != tested


def hipcc_version(hipcc: str) -> str:
    try:
        out = subprocess.run([hipcc, "--version"], capture_output=True, text=True).stdout
    except OSError:
        return ""
    for line in out.splitlines():
        if line.startswith("HIP version:"):
            return line.strip()
    return out.splitlines()[0].strip() if out else ""


def check_toolchain(hipcc: str, verbose: bool = True) -> str:
    """The toolchain's version line; raises unless it is the tested one or the override is set."""
    ver = hipcc_version(hipcc)
    if not same_toolchain(ver):
        msg = ("uvltrack_amd.build: hipcc is not the tested toolchain (%s); got: %s.  Hand-counted waits are only known-good there: "
               "set %s=1 to build anyway, then re-run the forced-kernel GPU tests." % (TESTED_HIPCC, ver or "?", OVERRIDE_ENV))
        if os.environ.get(OVERRIDE_ENV) != "1":
            raise RuntimeError(msg)
        if verbose:
            sys.stderr.write(msg + "\n")
    elif only_hash_differs(ver) and verbose:
        sys.stderr.write("uvltrack_amd.build: hipcc '%s' is the tested release with another build hash (tested: %s); accepted -- re-run the forced-kernel "
                         "GPU tests if results look off.\n" % (ver, TESTED_HIPCC))
    return ver


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(INCLUDE, "uvltrack_hip.h")]
    return _newest(deps) > os.path.getmtime(LIB)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    ver = check_toolchain(hipcc, verbose)
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        flags = _flags_for(src)
        if src == "uvl_api.hip":
            flags = flags + ['-DUVL_BUILD_TOOLCHAIN="%s"' % ver.replace('"', "'")]
        cmd = [hipcc] + flags + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            sys.stderr.write(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=5) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
