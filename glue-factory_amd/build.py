"""Build libmatchglue.so (all HIP kernels + the C ABI) for gfx950, in-tree.

    python glue-factory_amd/build.py [--force]

hipcc cross-compiles without a GPU; the .so lands in glue-factory_amd/lib/ (git-ignored,
but shipped to the GPU box with the tree).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libmatchglue.so")
SOURCES = ["api.hip", "gemm.hip", "conv.hip", "conv_f16.hip", "gemm_f16.hip", "attention_f16.hip", "sg_misc.hip", "nn_misc.hip", "gs_misc.hip", "sp_post.hip", "attention.hip", "lg_misc.hip", "ffn_f16.hip", "wf_misc.hip", "heads_f16.hip"]
# -packed-fp32-ops: no compiler-generated v_pk_*_f32.  On gfx950 a packed-fp32 instruction that broadcasts an operand half through
# op_sel / op_sel_hi returned wrong low halves, nondeterministically, while MFMAs were executing on the same SIMD (conv_f16.hip:
# conv1ab_dma_kernel, tests/test_gpu_determinism.py); the one place that wants packed FMAs writes them as inline asm on real pairs.
# -fma-mix-insts: no v_fma_mix*.  With scalar multiplies hipcc turned `(_Float16)(x * s)` into v_fma_mixlo_f16 (ONE rounding of the
# exact product) next to a separately rounded fp32 product: the hi half written by split8 and the hi half its lo was computed
# against then differed in the double-rounding cases, an error of one fp16 ulp (2^-11) in a value that should carry 22 bits
# (to_sh8_kernel; caught by tests/test_gpu_f16x3_range.py).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function",
         "-Xclang", "-target-feature", "-Xclang", "-fma-mix-insts"]
NO_PK = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
# conv_f16.hip keeps the instructions (the assembler rejects inline-asm v_pk_fma_f32 once the feature is off); instead its
# device assembly is checked below: no packed-fp32 instruction with an op_sel / op_sel_hi modifier may appear.
PK_CHECKED = {"conv_f16.hip"}
# ffn_f16.hip keeps compiler-generated packed fp32 as well (without it the LayerNorm / GELU phase costs 0.8 ms per step): one
# workgroup per CU, and its VALU phases with operand broadcasts run behind workgroup barriers, after every wave's MFMAs have retired
PK_ALLOWED = {"ffn_f16.hip"}
_NOISE = "is not a recognized feature for this target (ignoring feature)"  # the host pass of every file says this about both features


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "matchglue.h"), __file__]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not _stale():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [hipcc, *FLAGS, *([] if src in PK_CHECKED or src in PK_ALLOWED else NO_PK), "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        out = "\n".join(l for l in out.splitlines() if _NOISE not in l)
        if out.strip() and verbose:
            print(out)
        if p.returncode != 0:
            failed = True
            print(f"hipcc failed on {src}", file=sys.stderr)
    if failed:
        raise RuntimeError("libmatchglue build failed")
    import re
    for src in PK_CHECKED:
        asm = subprocess.run([hipcc, *FLAGS, "-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", "-"], stdout=subprocess.PIPE,
                             stderr=subprocess.DEVNULL, text=True).stdout
        bad = [l.strip() for l in asm.splitlines() if re.search(r"v_pk_[a-z]+_f32", l) and "op_sel" in l]
        if bad:
            raise RuntimeError(f"{src}: packed-fp32 instructions with an op_sel broadcast (unsafe beside MFMAs on gfx950): {bad[:3]} ...")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
