"""Build libradix_hip.so (gfx950) in-tree with hipcc.  No torch involved: the library is a
plain C-ABI shared object (include/radix_hip.h)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
# RX_LIB_NAME / RX_CFLAGS: developer knobs for A/B-ing kernel variants built side by side
LIB_PATH = os.path.join(HERE, os.environ.get("RX_LIB_NAME", "libradix_hip.so"))
SOURCES = ["rx_misc.hip", "rx_decode.hip", "rx_decode_mla.hip", "rx_extend.hip", "rx_extend32.hip", "rx_extend_nd.hip", "rx_extend_mla.hip", "rx_extend_d256.hip",
           "rx_allreduce.hip", "rx_quick_allreduce.hip", "rx_rope.hip", "rx_pool.hip", "rx_dcp.hip", "rx_radix.cpp"]
# RX_WITH_EXT64=1 (dev): the experimental one-wave-per-SIMD D = 128 extend kernel (tools/probe/rx_extend64.hip, 0.83x of the
# shipped kernel: DESIGN 4.2) is compiled in and option `ext64` routes PLAIN eight-wave calls to it.  The product library
# does not carry it; tools/ext64_check.py and tools/probe/pmc_ext64.sh need such a build.
WITH_EXT64 = bool(os.environ.get("RX_WITH_EXT64"))
PROBE = os.path.join(ROOT, "tools", "probe")
# rx_extend32: without -fno-honor-nans every fmaxf on an MFMA result is preceded by a canonicalising
# v_max_f32 x, x, x (one extra VALU per score in a VALU-issue-bound loop).  The kernel creates no NaN.
# -amdgpu-mfma-vgpr-form: at one wave per SIMD hipcc otherwise puts every MFMA result in AGPRs and
# copies each 16-register accumulator back for the softmax VALU (2000+ v_accvgpr moves in the kernel).
# -fno-slp-vectorize: plain -O3 packs the softmax's adjacent f32 adds into v_pk_add_f32 plus v_mov shuffles;
# beside MFMAs a packed f32 op costs more than the two scalar ones (MI355X_MICROARCH.md cycle constants):
# config-3 extend 758 -> 782 TFLOP/s.
EXTRA_FLAGS = {"rx_extend32.hip": ["-fno-honor-nans", "-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form=1"],
               "rx_extend64.hip": ["-fno-honor-nans", "-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form=1"],
               "rx_extend.hip": ["-fno-honor-nans", "-fno-slp-vectorize"],
               "rx_extend_nd.hip": ["-fno-honor-nans", "-fno-slp-vectorize"],
               "rx_extend_mla.hip": ["-fno-honor-nans", "-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form=1"],
               "rx_extend_d256.hip": ["-fno-honor-nans", "-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form=1"]}
HEADERS = ["rx_common.h", os.path.join(ROOT, "include", "radix_hip.h"), "rx_extend32_kernel.inc"]
if WITH_EXT64:
    SOURCES.append(os.path.join(PROBE, "rx_extend64.hip"))
    HEADERS.append(os.path.join(PROBE, "rx_extend64_kernel.inc"))


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, /opt/rocm/bin/hipcc, PATH)")


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]  # (join keeps an absolute source)
    deps += [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    force = force or bool(os.environ.get("RX_BUILD_FORCE"))
    if not force and not needs_build():
        return LIB_PATH
    hipcc = _hipcc()
    objs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    procs = []
    # RX_VARIANT_SOURCES (dev): with RX_LIB_NAME / RX_CFLAGS set, only these sources are built as the variant; every
    # other object is the default build's (a kernel A/B library then costs one compile, not fourteen)
    only = [x for x in os.environ.get("RX_VARIANT_SOURCES", "").split(",") if x]
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            continue
        variant = not only or src in only
        obj = os.path.join(HERE, "build", os.path.splitext(os.path.basename(src))[0] + (os.environ.get("RX_LIB_NAME", "") if variant else "")
                           + ("_ext64" if WITH_EXT64 else "") + ".o")
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall",
               "-Wno-unused-function", "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-c", sp, "-o", obj]
        cmd[1:1] = (EXTRA_FLAGS.get(os.path.basename(src), []) + (["-DRX_WITH_EXT64"] if WITH_EXT64 else [])
                    + (os.environ.get("RX_CFLAGS", "").split() if variant else []))
        objs.append(obj)
        # per-object incremental build: recompile only what is older than its source, the headers or its flags
        stamp = obj + ".cmd"
        deps = [sp] + [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
        if (not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == " ".join(cmd)
                and all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in deps)):
            continue
        # the stamp is written only AFTER a successful compile, and the stale object goes first: a failed or
        # interrupted compile must not leave an old .o behind a stamp that now matches the new flags
        for stale in (obj, stamp):
            if os.path.exists(stale):
                os.remove(stale)
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, stamp, " ".join(cmd), subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = []
    for src, stamp, line, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            failed.append(f"hipcc failed on {src}:\n{out.decode()}")
            continue
        with open(stamp, "w") as f:
            f.write(line)
        if verbose and out:
            print(out.decode(), file=sys.stderr)
    if failed:
        raise RuntimeError("\n".join(failed))
    tmp = LIB_PATH + ".tmp"
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs
    r = subprocess.run(link, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout.decode())
    os.replace(tmp, LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
