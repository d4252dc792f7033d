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
SOURCES = ["api.hip", "gemm.hip", "conv.hip", "conv_f16.hip", "gemm_f16.hip", "gemm_ws_f16.hip", "attention_f16.hip", "attention_pw_f16.hip", "sg_misc.hip", "nn_misc.hip", "gs_misc.hip", "sp_post.hip", "attention.hip", "lg_misc.hip", "ffn_f16.hip", "wf_misc.hip", "heads_f16.hip"]
# -packed-fp32-ops: no compiler-generated v_pk_*_f32.  On gfx950 a packed-fp32 instruction that broadcasts an operand half through
# op_sel / op_sel_hi returned wrong low halves, nondeterministically, while MFMAs were executing on the same SIMD (conv_f16.hip:
# conv1ab_dma_kernel, tests/test_gpu_determinism.py); the one place that wants packed FMAs writes them as inline asm on real pairs.
# -fma-mix-insts: no v_fma_mix*.  With scalar multiplies hipcc turned `(_Float16)(x * s)` into v_fma_mixlo_f16 (ONE rounding of the
# exact product) next to a separately rounded fp32 product: the hi half written by split8 and the hi half its lo was computed
# against then differed in the double-rounding cases, an error of one fp16 ulp (2^-11) in a value that should carry 22 bits
# (to_sh8_kernel; caught by tests/test_gpu_f16x3_range.py).
# attention_f16.hip is the one file built WITH the mix instructions: its probability split writes v_fma_mixlo/hi_f16 by hand (x - hi read
# straight from the fp16 half: 1.5 instead of 3 VALU instructions per element of a VALU-issue-bound loop); the only product that
# feeds a conversion there (o * 1/l in the epilogue) passes through inline asm first, so it exists as a rounded fp32 register.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]
NO_MIX = ["-Xclang", "-target-feature", "-Xclang", "-fma-mix-insts"]
MIX_ON = {"attention_f16.hip", "attention_pw_f16.hip", "gemm_ws_f16.hip"}
# conv_f16.hip is built with -fno-honor-nans: under IEEE mode every fmaxf(a, b) is preceded by canonicalising v_max_f32 x, x, x of its
# inputs and the quad-DPP moves of the 2x2 max-pool stay separate instructions; without NaNs to honour hipcc emits ONE v_max_f32_dpp per
# pooling step and one v_max per ReLU -- 384 of the ~1200 VALU instructions of a pooled tile epilogue, which no MFMA hides
# (profiles/r03_i_f16x3_sq_counters.json: 5.0 VALU per MFMA in conv1ab).  The file holds convolutions only: ReLU, max-pool, splits --
# no infinities, no NaN-producing arithmetic (attention, the assignment and Sinkhorn kernels use -inf masks and keep IEEE semantics).
NNAN = {"conv_f16.hip"}
NO_PK = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
# One file keeps the feature on: ffn_f16.hip (without it the LayerNorm / GELU phase costs 0.8 ms per step).  What is ENFORCED, on the device
# assembly of every translation unit after every build (pk_scan.py): no kernel that contains MFMAs may contain a packed-fp32
# instruction with an op_sel broadcast, except kernels named in pk_scan.ALLOW, and those only if no s_barrier interval of the kernel
# holds both an MFMA and such an instruction.
PK_ON = {"ffn_f16.hip"}
_NOISE = "is not a recognized feature for this target (ignoring feature)"  # the host pass of every file says this about both features


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "matchglue.h"), __file__, os.path.join(HERE, "pk_scan.py")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    # the filler schedule of attention_pw_f16.hip is generated (and committed): regenerate first, the file only changes when the generator did
    subprocess.check_call([sys.executable, os.path.join(CSRC, "gen_attention_pw_sched.py")], stdout=subprocess.DEVNULL)
    subprocess.check_call([sys.executable, os.path.join(CSRC, "gen_gemm_ws_sched.py")], stdout=subprocess.DEVNULL)
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
        # -save-temps=obj leaves the device assembly of exactly this compilation beside the object (scanned below, then removed)
        cmd = [hipcc, *FLAGS, *([] if src in MIX_ON else NO_MIX), *([] if src in PK_ON else NO_PK), *(["-fno-honor-nans"] if src in NNAN else []),
               "-save-temps=obj", "-c", os.path.join(CSRC, src), "-o", obj]
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
    import importlib.util
    import json

    spec = importlib.util.spec_from_file_location("mg_pk_scan", os.path.join(HERE, "pk_scan.py"))
    pk_scan = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pk_scan)
    violations, report = [], {}
    for src in SOURCES:
        stem = src.replace(".hip", "")
        asm_path = os.path.join(objdir, f"{stem}-hip-amdgcn-amd-amdhsa-gfx950.s")
        with open(asm_path) as f:
            v, rep = pk_scan.scan(f.read(), src)
        violations += v
        report[src] = {k: r for k, r in rep.items() if r["pk_f32"]}
        for f in os.listdir(objdir):  # the temporaries of -save-temps (pre-processed sources, bitcode, host assembly)
            if f.startswith(stem + "-") or f.startswith(stem + ".hip-"):
                os.remove(os.path.join(objdir, f))
    with open(os.path.join(LIBDIR, "pk_scan.json"), "w") as f:
        json.dump(report, f, indent=1)
    if violations:
        raise RuntimeError("packed-fp32 broadcast beside MFMAs (gfx950 hazard, see pk_scan.py):\n" + "\n".join(violations))
    if verbose:
        n = sum(r["pk_f32_broadcast"] for rep in report.values() for r in rep.values())
        print(f"pk_scan: {len(SOURCES)} translation units clean ({n} broadcasting packed-fp32 instructions, all in MFMA-free barrier intervals of allow-listed kernels)")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
