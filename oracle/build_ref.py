"""Build the reference's own native CPU attention kernels into oracle/_ref/ (git-ignored,
travels to the GPU box with the snapshot).  Sources are compiled where they lie under
/root/reference; flags from kernels/aot/csrc/cpu/CMakeLists.txt:118-131 minus the AMX switches
when ``amx=False`` (the GPU box's EPYC hosts have AVX-512/BF16 but no AMX; the kernels pick
their brgemm/AMX paths at run time through ATen's cpublas, the -mamx flags only enable
intrinsics that are not reached without AMX hardware).

TEST INFRASTRUCTURE ONLY.  Loading the built module: ``load()`` (no /root/reference needed).
"""
import glob
import importlib.util
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
REF_CPU = "/root/reference/python/sglang/kernels/aot/csrc/cpu"
NAME = "rx_ref_cpu"


def built_path():
    hits = glob.glob(os.path.join(OUT, NAME + "*.so"))
    return hits[0] if hits else None


def build(force: bool = False) -> str:
    if not force and built_path():
        return built_path()
    if not os.path.isdir(REF_CPU):
        raise RuntimeError("reference sources not present; oracle/_ref can only be built in the build container")
    from torch.utils.cpp_extension import load

    os.makedirs(OUT, exist_ok=True)
    srcs = [os.path.join(REF_CPU, f) for f in ("decode.cpp", "extend.cpp", "flash_attn.cpp", "kvcache.cpp")]
    srcs.append(os.path.join(HERE, "ref_binding.cpp"))
    flags = ["-O3", "-Wno-unknown-pragmas", "-march=x86-64-v4", "-mavx512bf16", "-mavx512vnni",
             "-mamx-tile", "-mamx-bf16", "-mamx-int8", "-fopenmp"]
    csrc = os.path.dirname(REF_CPU)
    load(name=NAME, sources=srcs, extra_cflags=flags, extra_ldflags=["-fopenmp"],
         extra_include_paths=[REF_CPU, csrc, os.path.join(os.path.dirname(csrc), "include")],
         build_directory=OUT, verbose=False)
    return built_path()


def load():
    """Import the prebuilt module (returns None when it was never built)."""
    path = built_path()
    if path is None:
        return None
    import torch  # noqa: F401  (the extension links against libtorch)

    spec = importlib.util.spec_from_file_location(NAME, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
