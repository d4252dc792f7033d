#!/usr/bin/env python3
"""Generates attention_pw_sched.inc: which vector / LDS / DMA instructions of one key-tile iteration of attention_pw_f16.hip sit in
the shadow of which MFMA.

    python glue-factory_amd/csrc/gen_attention_pw_sched.py [--report]

One wave per SIMD issues everything itself, in order.  A v_mfma_f32_32x32x16_f16 keeps the matrix pipe busy for 32 cycles and the
issue port for 8: a gap between two MFMAs hides ~24 cycles of other instructions (tests/micro/mfma_gap.hip, profiles/r04_f_mfma_gap.txt:
five v_add / cvt per gap are free, three v_exp or v_fma_mix are free, the fourth costs 8), but ONLY if they do not wait for each
other: a dependent v_add costs ~8 cycles, a dependent v_exp ~12, because no second wave fills the bubble.  The first version of the
kernel put whole soft-max steps (sub, exp, add, cvt, mix, mix -- a dependent chain) behind single MFMAs and ran at the SUM of its
matrix time and its vector time (profiles/r04_e_variants.txt).  Hence this table: every instruction is its own micro-op, and no
micro-op shares a gap with anything it depends on (producer at least one gap = one MFMA issue = >= 8 cycles earlier).

Iteration kt of the key loop (48 gaps): gaps 0-23 follow the MFMAs of S(kt+1) = K(kt+1) Q^T, gaps 24-47 those of O += V(kt) P(kt).
Micro-ops (cost in issue cycles):
  SB qt r   s[r] -= m_off                   4     EX qt r   s[r] = exp2(s[r])               8
  AD qt r   l[qt][r & 3] += s[r]            4     CV qt p   hi pair p = cvt_pkrtz(s[2p], s[2p+1])   4
  ML qt p   lo pair p, low half (fma_mix)   8     MH qt p   lo pair p, high half            8
  M1 qt i   t_i = max3 of S(kt+1) regs      4     M2 qt j / M3 qt: combine                  4
  SW qt     other lane half + max           14    VR i      transposed V(kt) read i of 16   4
  AV i      V read base address i of 4      4     AK i      K read address i of 8           4
  KR i      K(kt+2) fragment read i of 8    4     DM i      LDS-DMA piece i of 4 (K(kt+4), V(kt+2))   20
Constraints: P(qt, hf) (pairs 4 hf .. 4 hf + 3) complete before PV MFMA 12 hf + qt (gap 24 + 12 hf + qt); the V fragment (s2, dt, hl)
before its first PV MFMA, with >= 3 gaps of LDS latency; K fragment reads only after the last S MFMA that reads the register they
overwrite, and by the end of the iteration; row maxima only once the S(kt+1) chains have retired (>= 3 gaps after their last MFMA:
the asm MFMAs get no hazard padding from hipcc).
"""
import sys

NG = 48
CAP = 22          # issue cycles of fillers a gap may hold (24 hide; keep a little slack)
W = 6             # micro-ops per gap in the table

KINDS = ["NOP", "SB", "EX", "AD", "CV", "ML", "MH", "M1", "M2", "M3", "SW", "VR", "AV", "AK", "KR", "DM"]
COST = {"SB": 4, "EX": 8, "AD": 4, "CV": 4, "ML": 8, "MH": 8, "M1": 4, "M2": 4, "M3": 4, "SW": 14, "VR": 4, "AV": 4, "AK": 4, "KR": 4, "DM": 20}


class Op:
    def __init__(self, kind, a, b, earliest=0, deadline=NG - 1):
        self.kind, self.a, self.b = kind, a, b
        self.earliest, self.deadline = earliest, deadline
        self.deps = []      # (op, min gap distance)
        self.gap = None

    @property
    def code(self):
        return KINDS.index(self.kind) << 8 | self.a << 4 | self.b

    def __repr__(self):
        return f"{self.kind}({self.a},{self.b})"


def build_ops():
    ops = []

    def add(*args, **kw):
        o = Op(*args, **kw)
        ops.append(o)
        return o

    # soft-max finish of S(kt)
    for qt in range(2):
        ex = {}
        for r in range(16):
            sb = add("SB", qt, r)
            e = add("EX", qt, r)
            e.deps.append((sb, 1))
            ad = add("AD", qt, r)
            ad.deps.append((e, 1))
            ex[r] = e
        last_ad = {}
        for o in [o for o in ops if o.kind == "AD" and o.a == qt]:
            c = o.b & 3
            if c in last_ad:
                o.deps.append((last_ad[c], 1))   # same partial sum register
            last_ad[c] = o
        for p in range(8):
            hf = p >> 2
            cv = add("CV", qt, p)
            cv.deps += [(ex[2 * p], 1), (ex[2 * p + 1], 1)]
            ml = add("ML", qt, p)
            ml.deps.append((cv, 1))
            mh = add("MH", qt, p, deadline=24 + 12 * hf + qt - 1)
            mh.deps.append((ml, 1))
    # row maxima of S(kt+1): its chains end at gap 22 (qt 0) / 23 (qt 1)
    for qt in range(2):
        ready = 22 + qt + 3
        m1 = [add("M1", qt, i, earliest=ready) for i in range(5)]
        m2 = [add("M2", qt, j, earliest=ready) for j in range(2)]
        m2[0].deps += [(m1[0], 1), (m1[1], 1), (m1[2], 1)]
        m2[1].deps += [(m1[3], 1), (m1[4], 1)]
        m3 = add("M3", qt, 0)
        m3.deps += [(m2[0], 1), (m2[1], 1)]
        sw = add("SW", qt, 0)
        sw.deps.append((m3, 1))
    # V(kt) reads: i = (s2, dt, hl, jj); first use of fragment (s2, dt, hl): PV MFMA 12 s2 + 6 dt + (4 if hl else 0)
    av = [add("AV", 0, i) for i in range(4)]   # base (hl, dt) = (i >> 1, i & 1)
    for i in range(16):
        s2, dt, hl, jj = i >> 3, (i >> 2) & 1, (i >> 1) & 1, i & 1
        first = 24 + 12 * s2 + 6 * dt + (4 if hl else 0)
        vr = add("VR", 0, i, deadline=first - 4)
        vr.deps.append((av[hl * 2 + dt], 1))
    # K(kt+2) reads: kh[st] (i = 2 st) is read by S MFMAs 6 st .. 6 st + 3, kl[st] by 6 st + 4, 6 st + 5
    for i in range(8):
        st, lo = i >> 1, i & 1
        ak = add("AK", 0, i)
        kr = add("KR", 0, i, earliest=6 * st + (6 if lo else 4))
        kr.deps.append((ak, 1))
    for i in range(4):
        add("DM", 0, i)
    return ops


def propagate_deadlines(ops):
    changed = True
    while changed:
        changed = False
        for o in ops:
            for d, dist in o.deps:
                nd = o.deadline - dist
                if nd < d.deadline:
                    d.deadline = nd
                    changed = True


def schedule(ops, cap=CAP):
    gaps = [[] for _ in range(NG)]
    cost = [0] * NG
    todo = set(ops)
    for g in range(NG):
        while True:
            ready = [o for o in todo if o.earliest <= g and all(d.gap is not None and d.gap <= g - dist for d, dist in o.deps)]
            if not ready:
                break
            # most urgent first: slack to the deadline, then long chains first (SB before AD ...)
            ready.sort(key=lambda o: (o.deadline - g, KINDS.index(o.kind), o.a, o.b))  # a total order: the table is reproducible
            placed = False
            for o in ready:
                urgent = o.deadline <= g
                if len(gaps[g]) < W and (cost[g] + COST[o.kind] <= cap or urgent):
                    o.gap = g
                    gaps[g].append(o)
                    cost[g] += COST[o.kind]
                    todo.discard(o)
                    placed = True
                    break
            if not placed:
                break
    if todo:
        return None
    late = [o for o in ops if o.gap > o.deadline]
    if late:
        raise SystemExit(f"deadline missed: {[(repr(o), o.gap, o.deadline) for o in late]}")
    return gaps, cost


def emit(gaps, cost, path):
    import io

    f = io.StringIO()
    _emit(gaps, cost, f)
    text = f.getvalue()
    try:
        if open(path).read() == text:
            return  # unchanged: keep the file's time stamp (build.py rebuilds when a source is newer than the library)
    except OSError:
        pass
    with open(path, "w") as out:
        out.write(text)


def _emit(gaps, cost, f):
    if True:
        f.write("// GENERATED by gen_attention_pw_sched.py -- do not edit.  Micro-op codes: kind << 8 | a << 4 | b (kinds in sch::, attention_pw_f16.hip).\n")
        f.write(f"constexpr int W = {W};\n")
        for name, lo in (("P1", 0), ("P2", 24)):
            f.write(f"constexpr int {name}[24][W] = {{\n")
            for g in range(lo, lo + 24):
                codes = [o.code for o in gaps[g]] + [0] * (W - len(gaps[g]))
                f.write("    {" + ", ".join(f"0x{c:03x}" for c in codes) + "},  // " + f"gap {g:2d}, {cost[g]:2d} cycles: " + " ".join(map(repr, gaps[g])) + "\n")
            f.write("};\n")


def main():
    import os

    for cap in range(CAP, 41):   # the smallest per-gap budget under which everything fits (the work is what it is)
        ops = build_ops()
        propagate_deadlines(ops)
        res = schedule(ops, cap)
        if res:
            break
    else:
        raise SystemExit("no schedule")
    gaps, cost = res
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "attention_pw_sched.inc")
    emit(gaps, cost, out)
    if "--report" in sys.argv:
        for g in range(NG):
            print(f"gap {g:2d} cost {cost[g]:2d}  " + " ".join(map(repr, gaps[g])))
    print(f"{len(ops)} micro-ops, {sum(cost)} issue cycles over {NG} gaps (max {max(cost)}, cap {cap}); {sum(max(0, c - 24) for c in cost)} cycles beyond the shadows -> {out}")


if __name__ == "__main__":
    main()
