"""Build libtcdiff_gfx950.so in-tree with hipcc (cross-compiles for gfx950 without a GPU).

    python -m tcdiff_amd.build [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = [os.path.join(HERE, "csrc", f) for f in ("gemm.hip", "attention.hip", "ops.hip", "chain.hip", "train.hip", "train_ops.hip",
                                               "attention_train.hip", "gemm_rows.hip", "chain_split.hip")]
HDR = sorted(os.path.join(HERE, "csrc", h) for h in os.listdir(os.path.join(HERE, "csrc")) if h.endswith(".h")) + \
    [os.path.join(ROOT, "include", "tcdiff_hip.h")]
LIB = os.path.join(HERE, "libtcdiff_gfx950.so")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(f) > t for f in SRC + HDR)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for s in SRC:
        o = os.path.join(HERE, "build", os.path.basename(s) + ".o")
        objs.append(o)
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
               "-I" + os.path.join(HERE, "csrc"), "-c", s, "-o", o] + os.environ.get("TCDIFF_EXTRA_HIPCC_FLAGS", "").split()
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append(subprocess.Popen(cmd))
    for p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
