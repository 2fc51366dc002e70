"""Builds libbayesrul_amd.so (the C-ABI library of include/bayesrul_amd.h) for gfx950.

hipcc cross-compiles without a GPU, so this runs in the dev container; the built .so is
git-ignored but travels to the GPU box with the gpurun snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libbayesrul_amd.so")
SOURCES = ["plan.hip"]
NO_PK_F32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
HEADERS = ["common.h", "desc.h", "kernels_core.h", "kernels_group.h", "kernels_misc.h", "kernels_conv_bf.h", "kernels_dense_fwd.h", "kernels_trunk.h", "kernels_trunk_bwd.h", "kernels_trunk_dw.h", "kernels_dense_ks.h", "kernels_mlp.h", "kernels_f32.h",
           os.path.join("..", "..", "include", "bayesrul_amd.h")]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(HERE, f)) > t for f in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # packed fp32 VALU (v_pk_add_f32 / v_pk_fma_f32, which -O3 SLP-forms from adjacent scalar adds) is an anti-lever beside
    # MFMAs on gfx950 (MI355X_MICROARCH.md: +22..26 cycles per MFMA gap) and needs aligned register pairs: off
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC"] + NO_PK_F32 + ["-o", LIB] + SOURCES
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, cwd=HERE, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
