"""Static check of the device assembly for the gfx950 packed-fp32 hazard (hazard 1).

A packed-fp32 VALU instruction (`v_pk_{fma,mul,add}_f32`) that broadcasts one half of a register pair through `op_sel` /
`op_sel_hi` returned wrong low halves while MFMAs were executing on the same SIMD (minimal reproducer and its finding:
tests/micro/diag_conv1ab_det.py, profiles/r03_pk_hazard.md).  The rule enforced on EVERY translation unit of the library:

  * a kernel that contains no MFMA may use any packed-fp32 form;
  * a kernel that contains MFMAs must not contain a packed-fp32 instruction with an op_sel / op_sel_hi modifier that differs from
    the identity, unless the kernel is named in ALLOW below -- and then every such instruction must sit in an `s_barrier` interval
    (the straight-line stretch of the kernel's assembly between two consecutive workgroup barriers) that holds no MFMA, i.e. all
    waves of the workgroup have retired their MFMAs behind a barrier before the first broadcast issues and no MFMA is issued before
    the next barrier.  The allow-list entry names the barriers that do the separating, so a reader can find them in the source.

A second rule rides along: the kernels named in NO_SCRATCH count their VMEM instructions by hand (one `s_waitcnt vmcnt(N)` per stage
with a compile-time N: LDS-DMA pieces and output stores are issued unconditionally so that the count is exact).  A register spill is a
`scratch_*` instruction -- a VMEM operation hipcc adds behind the author's back -- so any scratch access in those kernels fails the build.

build.py runs this after compiling (same flags as the objects) and fails the build on a violation.
"""
import re
import subprocess

# kernel-name substring -> the barriers that separate its broadcasting VALU phase from the MFMA phases (ffn_f16.hip)
ALLOW = {
    "ffn_f16x3_kernel": "phase 2 (bias / LayerNorm / GELU / split) runs between the `__syncthreads()` that ends phase 1's stage loop and the "
                        "one that precedes phase 3's first MFMA; one workgroup per CU (158 KB of LDS), so no other workgroup's MFMAs share the SIMD",
}
NO_SCRATCH = ("conv3x3_dma_kernel", "conv1ab_mfma_kernel", "conv3x3_pair_kernel")  # counted vmcnt waits (conv_f16.hip)
_PK = re.compile(r"\bv_pk_(?:fma|mul|add)_f32\b")
_MFMA = re.compile(r"\bv_mfma_")
_FUNC = re.compile(r"^([A-Za-z_][\w$.]*):\s*(?:;.*)?$")


def _broadcasts(line):
    """True if the instruction's op_sel / op_sel_hi differ from the identity ([0,0,0] / [1,1,1]): some operand half is replicated or swapped."""
    m = re.search(r"op_sel:\[([01,]+)\]", line)
    if m and "1" in m.group(1):
        return True
    m = re.search(r"op_sel_hi:\[([01,]+)\]", line)
    return bool(m and "0" in m.group(1))


def kernels(asm):
    """Yield (name, [instruction lines]) for every function of a device assembly listing."""
    name, body = None, []
    for line in asm.splitlines():
        s = line.strip()
        if name is None:
            m = _FUNC.match(s)
            if m and not s.startswith(".L"):
                name, body = m.group(1), []
            continue
        if s.startswith(".Lfunc_end"):
            yield name, body
            name = None
            continue
        if s and not s.startswith((";", ".")) and not s.endswith(":"):
            body.append(s)


def scan(asm, src="<asm>"):
    """Returns (violations, report): report[kernel] = dict(mfma, pk, pk_broadcast, intervals_with_both)."""
    violations, report = [], {}
    for name, body in kernels(asm):
        n_mfma = sum(1 for l in body if _MFMA.search(l))
        pk = [l for l in body if _PK.search(l)]
        bc = [l for l in pk if _broadcasts(l)]
        mixed = 0
        if n_mfma and bc:
            has_mfma = has_bc = False
            for l in body + ["s_barrier"]:
                if l.startswith("s_barrier"):
                    mixed += int(has_mfma and has_bc)
                    has_mfma = has_bc = False
                elif _MFMA.search(l):
                    has_mfma = True
                elif _PK.search(l) and _broadcasts(l):
                    has_bc = True
            allowed = any(k in name for k in ALLOW)
            if not allowed:
                violations.append(f"{src}: {name}: {len(bc)} packed-fp32 instructions with an op_sel broadcast in a kernel with {n_mfma} MFMAs "
                                  f"(not in pk_scan.ALLOW), e.g. `{bc[0]}`")
            elif mixed:
                violations.append(f"{src}: {name}: {mixed} s_barrier interval(s) hold both an MFMA and a broadcasting packed-fp32 instruction")
        n_scratch = sum(1 for l in body if l.startswith("scratch_"))
        if n_scratch and any(k in name for k in NO_SCRATCH):
            violations.append(f"{src}: {name}: {n_scratch} scratch instructions in a kernel whose vmcnt waits are counted by hand")
        if n_mfma or pk:
            report[name] = {"mfma": n_mfma, "pk_f32": len(pk), "pk_f32_broadcast": len(bc), "barrier_intervals_with_both": mixed}
    return violations, report


def device_asm(hipcc, flags, path):
    return subprocess.Popen([hipcc, *flags, "-S", "--cuda-device-only", path, "-o", "-"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)


if __name__ == "__main__":
    import sys

    v, rep = scan(open(sys.argv[1]).read(), sys.argv[1])
    for k, r in rep.items():
        print(k, r)
    print("\n".join(v) or "clean")
