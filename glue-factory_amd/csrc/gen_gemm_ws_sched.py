#!/usr/bin/env python3
"""Generates gemm_ws_sched.inc: the per-MFMA filler tables of gemm_ws_f16.hip (weight-stationary Linear for K = 256).

    python glue-factory_amd/csrc/gen_gemm_ws_sched.py [--report]

A STAGE of that kernel is one quarter of the contraction of a 128-token tile: 4 k-steps x 12 MFMAs = 48 gaps (model of a gap:
gen_attention_pw_sched.py / tests/micro/mfma_gap.hip -- ~24 issue cycles hide behind an MFMA, and only for instructions that do not
wait on one another).  Besides its MFMAs a stage carries
  * the 32 token-fragment reads of the NEXT k-steps (BR u tb hl: k-step u of this stage for u = 1..3, k-step 0 of the next stage
    for u = 0 -- those come after the ring barrier that sits in front of MFMA 36),
  * 8 LDS-DMA pieces of the stage three ahead (DM, after the barrier: they overwrite the slot this stage has finished reading),
  * the epilogue of ONE 32-token block of the PREVIOUS tile, an instruction at a time: BF bias (acc x winv + b), RA..RF rotary on a
    pair, SC scale, HC hi halves of a pair (v_cvt_pk_f16_f32), LA / LB lo halves (fma_mix), WR / RO the trip of the lane's four 16-byte
    pieces through a wave-private LDS image that turns them into full-line stores (the 4x4 quad transpose of gemm_f16.hip costs 64
    vector instructions per block in a kernel that is bound by its instruction issue), ST the four stores, LE the four rotary-table pieces of the NEXT block
    (LDS-DMA into slots private to the wave: a register destination would be at the mercy of hipcc's copies while the load is in flight --
    first version: zeros in one block per tile), ER the reads of this block's pieces out of those slots.
Four variants: (rotary, scale) = q / k / v columns of the self projection and the qk columns of the cross projection.
Vector-memory operations retire IN ORDER, so a wait for a young request is a wait for every older one: requests are therefore issued
in the order they are needed.  A stage issues its 4 stores before the barrier gap and, after it, ONE packet -- the 8 token pieces and
the 4 rotary pieces of the stage RD ahead (same lead for both; the first version asked for the rotary pieces one stage ahead and waited
for them at gap 0, which also waited for the token pieces behind which they queued: 29 % of the wave cycles parked).  The single
counted wait in front of MFMA 36 lets this stage's stores and the previous stage's operations fly.
"""
import io
import os
import sys

NG = 48
W = 7
BARRIER_GAP = 36

KINDS = ["NOP", "BR", "DM", "LE", "ER", "BF", "RA", "RB", "RC", "RD", "RE", "RF", "SC", "HC", "LA", "LB", "WR", "RO", "ST"]
COST = {"BR": 4, "DM": 20, "LE": 20, "ER": 4, "BF": 8, "RA": 4, "RB": 4, "RC": 4, "RD": 4, "RE": 4, "RF": 4, "SC": 4, "HC": 4, "LA": 8, "LB": 8,
        "WR": 13, "RO": 4, "ST": 12}


class Op:
    def __init__(self, kind, a, b, earliest=0, deadline=NG - 1):
        self.kind, self.a, self.b, self.earliest, self.deadline = kind, a, b, earliest, deadline
        self.deps, self.gap = [], None
        # epilogue ops belong to the chain of their register pair: the scheduler finishes a pair before it starts the next one, so that
        # few temporaries are alive at a time (the first table ran all 16 bias ops, then all 32 rotary products ...: 35 values per stage
        # were parked in AGPRs and came back through v_accvgpr_read, 8 issue cycles each)
        self.chain = {"BF": b >> 1, "SC": b >> 1}.get(kind, b if kind in ("RA", "RB", "RC", "RD", "RE", "RF", "HC", "LA", "LB") else 0)

    @property
    def code(self):
        return KINDS.index(self.kind) << 8 | self.a << 4 | self.b

    def __repr__(self):
        return f"{self.kind}({self.a},{self.b})"


def build(rot, sc):
    ops = []

    def add(*a, **k):
        o = Op(*a, **k)
        ops.append(o)
        return o

    # token fragment reads: a = u (target k-step), b = 2 tb + hl
    for u in range(4):
        for tb in range(4):
            for hl in range(2):
                if u == 0:
                    add("BR", u, 2 * tb + hl, earliest=BARRIER_GAP, deadline=min(47, 48 + (4 if hl else 0) + tb - 6))
                else:
                    add("BR", u, 2 * tb + hl, earliest=12 * (u - 1), deadline=min(12 * u + (4 if hl else 0) + tb - 6, 12 * (u - 1) + 8))
    for i in range(8):
        add("DM", 0, i, earliest=BARRIER_GAP)
    er = {}
    if rot:
        for i in range(4):
            add("LE", 0, i, earliest=BARRIER_GAP)  # requested WITH the token pieces of the same stage (same lead, same counted wait)
            er[i] = add("ER", 0, i)      # this stage's rotary table out of the wave's LDS slots (landed before the previous stage's barrier)
    # epilogue of one block: 16 values = 8 pairs
    v = {}
    for j in range(16):
        v[j] = add("BF", 0, j)
    if rot:
        for p in range(8):
            ra, rb, rc, rd = add("RA", 0, p, earliest=1), add("RB", 0, p, earliest=1), add("RC", 0, p, earliest=1), add("RD", 0, p, earliest=1)
            for o in (ra, rd):
                o.deps.append((v[2 * p], 1))
            for o in (rb, rc):
                o.deps.append((v[2 * p + 1], 1))
            for o in (ra, rc):
                o.deps.append((er[p >> 2], 3))       # cos pieces 0, 1; LDS latency: three gaps
            for o in (rb, rd):
                o.deps.append((er[2 + (p >> 2)], 3))  # sin pieces 2, 3
            re_, rf = add("RE", 0, p), add("RF", 0, p)
            re_.deps += [(ra, 1), (rb, 1), (rd, 0)]   # RE overwrites v[2p], which RD reads: same gap is fine if RD is listed first
            rf.deps += [(rc, 1), (rd, 1), (rb, 0)]
            v[2 * p], v[2 * p + 1] = re_, rf
    if sc:
        for j in range(16):
            s = add("SC", 0, j)
            s.deps.append((v[j], 1))
            v[j] = s
    hc, lb = {}, {}
    for p in range(8):
        h = add("HC", 0, p)
        h.deps += [(v[2 * p], 1), (v[2 * p + 1], 1)]
        la = add("LA", 0, p)
        la.deps.append((h, 1))
        l2 = add("LB", 0, p)
        l2.deps.append((la, 1))
        hc[p], lb[p] = h, l2
    # the lane's four 16-byte pieces (P0 = hi pairs 0-3, P1 = lo pairs 0-3, P2 = hi pairs 4-7, P3 = lo pairs 4-7) go through a 4-KB LDS
    # image private to the wave ([token][8 pieces], swizzled): WR q writes piece q, RO it reads back piece (lane & 7) of token
    # (lane >> 3) + 8 it, ST it stores it -- eight consecutive lanes then write the 128 contiguous bytes of one token: full lines
    wr = []
    for q in range(4):
        w = add("WR", 0, q)
        src = hc if q in (0, 2) else lb
        for p in range(4 * (q >> 1), 4 * (q >> 1) + 4):
            w.deps.append((src[p], 1))
        wr.append(w)
    for it in range(4):
        ro = add("RO", 0, it)
        for w in wr:
            ro.deps.append((w, 1))
        st = add("ST", 0, it, deadline=BARRIER_GAP - 1)   # all stores before the barrier gap (VMEM order LE, ST, DM)
        st.deps.append((ro, 4))                         # LDS latency
    return ops


def propagate(ops):
    ch = True
    while ch:
        ch = False
        for o in ops:
            for d, dist in o.deps:
                nd = o.deadline - dist
                if nd < d.deadline:
                    d.deadline, ch = nd, True


def schedule(ops, cap):
    gaps, cost, todo = [[] for _ in range(NG)], [0] * NG, set(ops)
    for g in range(NG):
        while True:
            ready = [o for o in todo if o.earliest <= g and all(d.gap is not None and d.gap <= g - dist for d, dist in o.deps)]
            if not ready:
                break
            ready.sort(key=lambda o: (o.deadline - g if o.kind in ('BR', 'DM', 'LE', 'ST') else 99, o.chain, KINDS.index(o.kind), o.a, o.b))
            for o in ready:
                if len(gaps[g]) < W and (cost[g] + COST[o.kind] <= cap or o.deadline <= g):
                    o.gap = g
                    gaps[g].append(o)
                    cost[g] += COST[o.kind]
                    todo.discard(o)
                    break
            else:
                break
    if todo or any(o.gap > o.deadline for o in ops):
        return None
    return gaps, cost


def main():
    out = io.StringIO()
    out.write("// GENERATED by gen_gemm_ws_sched.py -- do not edit.  Micro-op codes: kind << 8 | a << 4 | b (kinds in gsch::, gemm_ws_f16.hip).\n")
    out.write(f"constexpr int W = {W};\nconstexpr int BARRIER_GAP = {BARRIER_GAP};\n")
    for rot in (0, 1):
        for sc in (0, 1):
            for cap in range(20, 60):
                ops = build(rot, sc)
                propagate(ops)
                res = schedule(ops, cap)
                if res:
                    break
            else:
                raise SystemExit(f"no schedule for rot={rot} sc={sc}")
            gaps, cost = res
            # the VMEM order inside a stage must be LE, ST, DM (the kernel's counted waits are constants derived from it)
            order = ["ST" if o.kind == "ST" else "RQ" for g in gaps for o in g if o.kind in ("LE", "ST", "DM")]
            assert order == sorted(order, key=["ST", "RQ"].index), order
            out.write(f"constexpr int T{rot}{sc}[48][W] = {{  // rotary {rot}, scale {sc}: {len(ops)} micro-ops, {sum(cost)} issue cycles (cap {cap} per gap, max {max(cost)})\n")
            for g in range(NG):
                codes = [o.code for o in gaps[g]] + [0] * (W - len(gaps[g]))
                out.write("    {" + ", ".join(f"0x{c:04x}" for c in codes) + "},  // " + f"gap {g:2d}, {cost[g]:2d}: " + " ".join(map(repr, gaps[g])) + "\n")
            out.write("};\n")
            print(f"rot {rot} sc {sc}: {len(ops)} micro-ops, {sum(cost)} cycles, cap {cap}, max {max(cost)}, beyond shadows {sum(max(0, c - 24) for c in cost)}")
            if "--report" in sys.argv:
                for g in range(NG):
                    print(f"  gap {g:2d} {cost[g]:2d}  " + " ".join(map(repr, gaps[g])))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_ws_sched.inc")
    text = out.getvalue()
    try:
        if open(path).read() == text:
            return
    except OSError:
        pass
    open(path, "w").write(text)


if __name__ == "__main__":
    main()
