"""Build libwfstdec.so (HIP kernels + C ABI) for gfx950, in-tree.

hipcc cross-compiles without a GPU.  The result is asr-decoder_amd/lib/libwfstdec.so; it is
git-ignored (history stays source-only) but travels with the repo snapshot to the GPU box.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRCS = [os.path.join(HERE, "csrc", "wfst_kernels.hip"), os.path.join(HERE, "csrc", "wfst_nbest.hip"),
        os.path.join(HERE, "csrc", "wfst_determinize.hip"), os.path.join(HERE, "csrc", "wfst_compose.hip"),
        os.path.join(HERE, "csrc", "wfst_capi.cc"),
        os.path.join(HERE, "csrc", "wfst_openfst.cc")]
HDRS = [os.path.join(HERE, "csrc", "wfst_device.h"), os.path.join(HERE, "csrc", "wfst_determinize.h"), os.path.join(HERE, "csrc", "wfst_determinize_wave.h"), os.path.join(HERE, "csrc", "wfst_openfst.h"),
        os.path.join(HERE, "..", "include", "wfst_decoder.h")]
LIB = os.path.join(HERE, "lib", "libwfstdec.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: the search must round like the reference (no FMA; configure.ac:12-13)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-Wall", "-Wno-unused-function"]


def up_to_date():
    if not os.path.exists(LIB):
        return False
    t = os.path.getmtime(LIB)
    return all(os.path.getmtime(p) <= t for p in SRCS + HDRS + [os.path.abspath(__file__)])


def build(force=False, verbose=False):
    if not force and up_to_date():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    cmd = [HIPCC] + FLAGS + ["-o", LIB] + SRCS
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


def build_variant(name, defines):
    """Kernel experiments: the library built with extra -D flags as lib/libwfstdec_<name>.so (wfstdec.py loads it when
    WFST_LIB_VARIANT=<name>; tools/ab_bench.sh)."""
    out = os.path.join(HERE, "lib", "libwfstdec_%s.so" % name)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call([HIPCC] + FLAGS + list(defines) + ["-o", out] + SRCS)
    return out


if __name__ == "__main__":
    if "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], [a for a in sys.argv[i + 2:] if a.startswith("-D")]))
    else:
        build(force="--force" in sys.argv, verbose=True)
        print(LIB)
