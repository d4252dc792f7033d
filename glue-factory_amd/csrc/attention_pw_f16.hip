// Flash-style attention, split-fp16 (SH8) operands, ONE WAVE PER SIMD with the whole 512-entry register file
// (same math as attention_f16.hip: lightglue.py:105-115 self, :200-210 cross).  The default attention kernel of the f16x3 mode since
// round 4; attention_f16.hip (8 waves, two per SIMD) stays behind the debug request "lg.attn8".
//
// Why a second kernel.  attention_f16.hip runs two waves per SIMD in lockstep behind one barrier per key tile; inside a wave the
// chain S -> max -> exp -> split -> PV is serial, and six re-arrangements over three rounds all died of the 256-register limit
// (spills, LDS re-reads).  The structure the guide describes (cdna_hip_programming.md, Appendix B "4-wave, one-wave-per-SIMD,
// persistent structure") spends the opposite budget: a workgroup of FOUR waves, one per SIMD, each with up to 512 registers (arch
// VGPRs + accumulator file), and the overlap done INSIDE the wave by software pipelining -- the matrix pipe executes tile t+1's QK^T
// while the vector pipe finishes tile t's soft-max, and tile t's PV while tile t+1's row maxima are taken.  One MFMA occupies the
// pipe for 32 cycles and the vector issue port for 8 of them: every MFMA gap hides ~24 cycles of other instructions -- five plain VALU
// or three v_exp / v_fma_mix, measured (tests/micro/mfma_gap.hip, profiles/r04_f_mfma_gap.txt) -- provided they do not wait for each
// other: a single wave has nobody to cover a dependency stall (a dependent v_add costs 8 cycles, a dependent v_exp 12).
//
// What it bought, measured (profiles/r04_h_attn_pmc.txt, r04_i_*): 451 k -> 388 k shader cycles per launch (MFMA-busy 0.44 -> 0.51), but
// the clock the chip holds falls with it, 2.17 -> 1.92 GHz: 208 -> 202 us in the stand-alone entry, 1.90 -> 1.83 ms per step for the nine
// layers in bench.py.  The launch time of every variant is the SUM of what its parts cost alone (MFMAs only 131 us, + LDS / DMA 23,
// + soft-max VALU 48 = 202): the board sits at ~1 260 W of its 1 400 W limit during the step (profiles/r04_i_power.txt) and the
// kernel is bound by ENERGY, not by the schedule -- see DESIGN.md, round 4.  The kernel is kept because it is the faster one and
// because its machinery (asm-owned accumulator classes, LDS-DMA rings with counted waits, generated per-instruction schedules) is
// what the other matrix kernels can reuse; further gains here need fewer operations per tile, not a better order of the same ones.
//
// Structure (per block: 256 queries of one (image, head); a wave owns 64 = two 32-query MFMA tiles; one PERSISTENT workgroup per CU
// walks the blocks of its XCD, the blocks of one (image, head) adjacent so that their K / V rows come out of one L2):
//   * K and V tiles (32 keys x 256 B each) arrive by LDS-DMA written as asm (buffer_load_dwordx4 ... lds: invisible to hipcc's waitcnt
//     insertion, which otherwise drains vmcnt to 0 in front of every later LDS read; rows past the sequence end are out of the
//     descriptor's range and land as zeros).  3-deep rings, tile tau in slot tau % 3; iteration kt requests K(kt+4) and V(kt+2) and
//     waits, at its top, for everything but its predecessor's four pieces (vmcnt retires in order): two iterations of flight.
//   * S^T = K Q^T as asm MFMAs with the accumulators in arch VGPRs (hipcc puts every MFMA builtin of a > 256-register kernel into the
//     accumulator file: 64 v_accvgpr moves per tile for a tile the soft-max reads element by element); Q fragments live in AGPRs
//     ("a" operands), O accumulates through builtins in AGPRs and is never touched by the vector pipe inside the loop.
//   * Masking of a partial last tile: the first MFMA of its chains takes a C operand with -inf in the rows of keys past the end.
//   * V is consumed through ds_read_b64_tr_b16 (the hardware transpose read of gfx950): the LDS image is the plain row copy the DMA
//     writes, no 2-byte transposing stores; the swizzle (row & 1) | ((row & 2) << 2) on the piece index makes the 32 lanes of a half
//     cover the 64 banks exactly once (SQ_LDS_BANK_CONFLICT = 0).
//   * P' = 2^7 exp2(s - m_run) split hi/lo in registers and used directly as the B operand of O^T += V^T P^T (accumulator-as-operand;
//     element j of lane half h is key 16 s + 8 (j >> 2) + 4 h + (j & 3), which is the row order the transposed reads deliver).
//   * The running offset is raised only when a query outgrows it by 2^8 (wave-uniform branch after the tile's PV MFMAs); four partial
//     row sums per query tile (register r adds to r & 3) keep the adds of a gap independent.
//   * Iteration kt = 48 MFMAs: 24 of S(kt+1), then 24 of PV(kt).  Every other instruction of the iteration is a micro-op of the
//     generated table attention_pw_sched.inc (gen_attention_pw_sched.py: costs, dependencies at >= one gap distance, deadlines), placed
//     behind a fixed MFMA and fenced by sched_barriers: the order in the table is the order in the binary.
#include <type_traits>
#include <utility>

#include "common.h"
#include "kernels.h"

namespace {

#define PIN() __builtin_amdgcn_sched_barrier(0)

constexpr int KT = 32, NWV = 4, QW = 64, QB = NWV * QW;  // keys per tile; waves; queries per wave / per workgroup
constexpr int RD = 3;                                    // K / V ring depth: a tile is requested two iterations before the barrier that publishes it
constexpr float RESCALE_THR = 8.f;                       // log2 units
constexpr float P_SHIFT = 7.f;                           // P' = 2^7 exp2(s - m_run) <= 2^15 (attention_f16.hip)

typedef __fp16 h16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

// The probabilities P' (>= 0, fp16 range) are split as in attention_f16.hip: hi = P' truncated to 11 bits (v_cvt_pkrtz_f16_f32 on a
// register pair), lo = fp16(P' - hi) rounded to nearest, evaluated by the mixed-precision FMA straight from the fp16 hi half.
// single instructions the schedule places one by one (asm: no canonicalising v_max in front of an IEEE fmaxf, no re-association)
__device__ __forceinline__ float max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float max2(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ unsigned mix_lo(unsigned hi, float x0) {  // low half: fp16(x0 - hi.lo), rounded to nearest
    unsigned lo;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(x0));
    return lo;
}
__device__ __forceinline__ unsigned mix_hi(unsigned lo, unsigned hi, float x1) {  // high half: fp16(x1 - hi.hi); the low half stays
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(x1));
    return lo;
}

template <int N>
using ic = std::integral_constant<int, N>;
// compile-time loop: f(ic<0>{}), f(ic<1>{}), ... -- the schedule tables are indexed with constants by construction, not by the unroller's grace
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(ic<I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// The S chains are written as asm with VGPR accumulators.  With 512 registers available hipcc selects the AGPR form for every MFMA
// builtin of the kernel (destination and C operand in the accumulator file), which is right for O -- never touched by the vector pipe
// inside the loop -- and wrong for S, which the soft-max reads element by element: 32 v_accvgpr_read + 32 v_accvgpr_write per key tile
// (first build of this file).  In asm the accumulator class is ours: D / C in arch VGPRs ("v"), the Q fragment (B operand, constant
// over the whole kernel) in the accumulator file ("a"; MFMA A / B operands may come from either file).  hipcc pads no hazards around an
// asm MFMA (cdna_hip_programming.md 5.7): every consumer of these results sits >= 4 MFMAs (128 cycles) later by construction of the
// schedule below, and the operands they read were written at least one phase earlier.
__device__ __forceinline__ void mfma_v_first(f32x16& d, const f16x8 a, const f16x8 b_acc, const f32x16 c) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "a"(b_acc), "v"(c));
}
__device__ __forceinline__ void mfma_v_first0(f32x16& d, const f16x8 a, const f16x8 b_acc) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "a"(b_acc));
}
__device__ __forceinline__ void mfma_v_acc(f32x16& d, const f16x8 a, const f16x8 b_acc) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(b_acc));
}
// LDS-DMA piece (1 KiB per wave-instruction) in asm: invisible to hipcc's s_waitcnt insertion, which otherwise assumes that every
// later ds_read may alias the piece and drains vmcnt to 0 in front of it (first build: a full memory latency exposed per tile).  M0
// (LDS byte address of the piece, wave-uniform) is written in the same statement that uses it; completion is awaited by the counted
// s_waitcnt + barrier at the top of the iteration that reads the data.
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma_piece(const i32x4 rsrc, unsigned lds_byte, int voffset) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_byte), "v"(voffset), "s"(rsrc) : "memory");
}
__device__ __forceinline__ i32x4 make_rsrc(const void* base, int bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));  // stride 0, no swizzle
    r[2] = __builtin_amdgcn_readfirstlane(bytes);
    r[3] = 0x00020000;
    return r;
}

// ---- the schedule: which instructions follow which MFMA (generated: gen_attention_pw_sched.py explains the model and the constraints) ----
namespace sch {
enum { NOP = 0, SB, EX, AD, CV, ML, MH, M1, M2, M3, SW, VR, AV, AK, KR, DM };  // = KINDS of the generator
#include "attention_pw_sched.inc"
}  // namespace sch

// EXP != 0: TIMING ablations (results are wrong on purpose; debug requests "lg.attn.exp<N>", bench A/B only): bit 0 drops the soft-max
// VALU work (exp / sums / split / row maxima), bit 1 the LDS-DMA requests and their waits, bit 2 the MFMAs, bit 3 the LDS fragment reads, bit 4 the epilogue (O / l, split, stores).
template <int EXP>
__global__ __launch_bounds__(256) void attention_f16x3_pw_kernel(const AttnF16Args p) {
    __shared__ u32x4 s_k[RD][KT * 16];
    __shared__ u32x4 s_v[RD][KT * 16];

    const int t = threadIdx.x, lane = t & 63;
    const int wu = __builtin_amdgcn_readfirstlane(t >> 6);
    // 1-D grid, XCD-aware: workgroups are dealt round-robin over the 8 XCDs, so id = (chunk * nqb + qblk) * 8 + xcd gives the nqb query
    // blocks of one (sequence, head) to ONE XCD in adjacent slots: they stream the same K / V rows, which then come out of that L2.
    const int Lmax = p.M > p.N ? p.M : p.N;
    const int nqb = (Lmax + QB - 1) / QB, groups = p.heads * 2 * p.B;
    // PERSISTENT: one workgroup per CU walks the slots of its XCD (a finished workgroup otherwise leaves its CU idle until the dispatcher
    // has placed the next one: ~8 k of the ~77 k cycles a 32-tile block takes, profiles/r04_h_attn_pmc.txt)
    const int xcd = blockIdx.x & 7, nslots = ((groups + 7) / 8) * nqb;
    for (int slot = blockIdx.x >> 3; slot < nslots; slot += gridDim.x >> 3) {
    const int grp = (slot / nqb) * 8 + xcd;
    if (grp >= groups) continue;
    const int head = grp % p.heads;
    const int seq = grp / p.heads;
    const int side = seq >= p.B ? 1 : 0;
    const int b = seq - side * p.B;
    const int Lq = side ? p.N : p.M;
    const int kside = p.cross ? (1 - side) : side;
    const int Lkv = kside ? p.N : p.M;
    const long long qbase = side ? ((long long)p.B * p.M + (long long)b * p.N) : (long long)b * p.M;
    const long long kbase = kside ? ((long long)p.B * p.M + (long long)b * p.N) : (long long)b * p.M;
    const int q0 = (slot % nqb) * QB;
    if (q0 >= Lq || Lkv <= 0) continue;  // workgroup-uniform

    const int i32 = lane & 31, h = lane >> 5;
    const int ld4 = p.ld >> 2;
    const int rowb = p.ld * 4;  // bytes per SH8 row
    const u32x4* Q = reinterpret_cast<const u32x4*>(p.q);

    // ---- Q fragments (B operand of S^T = K Q^T): 2 query tiles x 4 k-steps x (hi, lo) ----------------------------------------
    f16x8 qh[2][4], ql[2][4];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int qi = min(q0 + wu * QW + qt * 32 + i32, Lq - 1);
        const u32x4* qrow = Q + (qbase + qi) * ld4 + ((p.q_col + head * 64) >> 2);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            qh[qt][s] = __builtin_bit_cast(f16x8, qrow[(2 * s + h) * 2]);
            ql[qt][s] = __builtin_bit_cast(f16x8, qrow[(2 * s + h) * 2 + 1]);
        }
    }

    // ---- K / V tiles by LDS-DMA -------------------------------------------------------------------------------------------
    // A tile is 32 rows x 16 pieces of 16 B = 8 pieces of 1 KiB; wave w moves pieces 2w, 2w+1 (rows 8w .. 8w+7) of K and of V.
    // LDS slot (row, pos) receives source piece pos ^ swz(row): K rows are read by ds_read_b128 (swz = row & 15, conflict-free as in
    // attention_f16.hip), V rows by ds_read_b64_tr_b16 (swz = (row & 1) | ((row & 2) << 2): the 32 lanes of a half -- 4 rows x 2
    // column blocks x 4 pieces -- then cover the 64 banks exactly once).
    const char* kptr = reinterpret_cast<const char*>(p.k) + ((kbase * p.ld + p.k_col + head * 64) << 2);
    const char* vptr = reinterpret_cast<const char*>(p.v) + ((kbase * p.ld + p.v_col + head * 64) << 2);
    const int nrec = (Lkv - 1) * rowb + 256;  // bytes of this sequence's rows that belong to this head's 64 columns (last row: its 256 B only)
    int vo_k[2], vo_v[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (2 * wu + j) * 4 + (lane >> 4), pos = lane & 15;
        vo_k[j] = row * rowb + ((pos ^ (row & 15)) << 4);
        vo_v[j] = row * rowb + ((pos ^ ((row & 1) | ((row & 2) << 2))) << 4);
    }
    const int ntiles = (Lkv + KT - 1) / KT;
    // one descriptor per operand for the whole sequence; the tile offset goes into the per-lane offset, which IS range-checked: tiles
    // past the end are entirely out of range and land as zeros (every wave still issues the same number of pieces)
    const i32x4 rs_k = make_rsrc(kptr, nrec), rs_v = make_rsrc(vptr, nrec);
    const unsigned lds_k = (unsigned)(size_t)(__attribute__((address_space(3))) char*)reinterpret_cast<char*>(&s_k[0][0]);
    const unsigned lds_v = (unsigned)(size_t)(__attribute__((address_space(3))) char*)reinterpret_cast<char*>(&s_v[0][0]);
    auto dma_k = [&](int kt, int slot, int j) { dma_piece(rs_k, __builtin_amdgcn_readfirstlane(lds_k + slot * (KT * 256) + (2 * wu + j) * 1024), vo_k[j] + kt * (KT * rowb)); };
    auto dma_v = [&](int kt, int slot, int j) { dma_piece(rs_v, __builtin_amdgcn_readfirstlane(lds_v + slot * (KT * 256) + (2 * wu + j) * 1024), vo_v[j] + kt * (KT * rowb)); };

    // fragment addresses (16-byte piece index inside a tile for K, byte offset for V's transposed reads)
    int ka[4];  // K: row i32, piece (2 st + h) * 2 [+1 for lo]: the XOR with an even/odd pair keeps hi / lo adjacent -> lo = hi ^ 1
#pragma unroll
    for (int st = 0; st < 4; ++st) ka[st] = i32 * 16 + (((2 * st + h) * 2) ^ (i32 & 15));
    // V: lane = 16 G + 4 q + p: row 16 s2 + 8 jj + 4 h + q (h = G >> 1), source piece c = 8 dt + 4 (G & 1) + 2 (p >> 1) + hl, byte 8 (p & 1)
    int va[2];  // [hl]; + s2 * 16 rows, + jj * 8 rows, + dt * 8 pieces are compile-time offsets applied below (XOR-safe: see vaddr)
    {
        const int G = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
        const int rowq = 4 * (G >> 1) + q;                       // + 16 s2 + 8 jj (multiples of 8: swz unchanged, it uses row bits 0-1)
        const int swz = (rowq & 1) | ((rowq & 2) << 2);
#pragma unroll
        for (int hl = 0; hl < 2; ++hl) {
            const int c = 4 * (G & 1) + 2 * (pp >> 1) + hl;      // + 8 dt: bit 3, which the swizzle may flip -> applied as XOR below
            va[hl] = rowq * 256 + ((c ^ swz) << 4) + 8 * (pp & 1);
        }
    }
    // absolute LDS byte addresses for ring slot 0 (kept for the whole kernel); per iteration the AK / AV micro-ops add the slot offset
    unsigned ka0[8], va0[4];  // K: fragment read i = 2 st + (lo); V: base i = 2 hl + dt (dt toggles piece bit 3 = byte 128)
#pragma unroll
    for (int i = 0; i < 8; ++i) ka0[i] = lds_k + ((ka[i >> 1] ^ (i & 1)) << 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) va0[i] = lds_v + (va[i >> 1] ^ ((i & 1) << 7));
    unsigned kad[8], vad[4];  // this iteration's addresses
    typedef __attribute__((address_space(3))) f16x8 lds_f16x8;
    typedef __attribute__((address_space(3))) h16x4 lds_h16x4;

    f16x8 kh[4], kl[4];
    auto read_kf = [&](int i) {  // fragment read i of 8 at kad[i]
        const f16x8 v = *(const lds_f16x8*)(size_t)kad[i];
        if (i & 1) kl[i >> 1] = v;
        else kh[i >> 1] = v;
    };
    f16x8 vh[2][2], vl[2][2];  // [s2][dt]
    auto read_v = [&](int i) {  // transposed read i of 16: (s2, dt, hl, jj) at vad[2 hl + dt] + rows (16 s2 + 8 jj)
        const int s2 = i >> 3, dt = (i >> 2) & 1, hl = (i >> 1) & 1, jj = i & 1;
        const h16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h16x4*)(size_t)(vad[2 * hl + dt] + (16 * s2 + 8 * jj) * 256));
        u64x2 cur = __builtin_bit_cast(u64x2, hl ? vl[s2][dt] : vh[s2][dt]);
        cur[jj] = __builtin_bit_cast(unsigned long long, r);
        if (hl) vl[s2][dt] = __builtin_bit_cast(f16x8, cur);
        else vh[s2][dt] = __builtin_bit_cast(f16x8, cur);
    };
    auto set_kad = [&](int slot) {
#pragma unroll
        for (int i = 0; i < 8; ++i) kad[i] = ka0[i] + slot * (KT * 256);
    };

    // ---- state ---------------------------------------------------------------------------------------------------------------
    f32x16 o[2][2];    // [query tile][d tile]
    f32x16 cmask;      // C operand that starts the S chains of a PARTIAL last tile: -inf in the rows of keys past the end, else 0
    f32x16 s[2][2];    // [buffer][query tile]
    f16x8 ph[2][2], pl[2][2];  // [query tile][k-step]
    float m_off[2] = {0.f, 0.f};
    float l_run[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};  // four partial row sums per query tile (register r adds to r & 3)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        o[0][0][r] = 0.f; o[0][1][r] = 0.f; o[1][0][r] = 0.f; o[1][1][r] = 0.f;
        cmask[r] = 0.f;
    }

    // S(kt) MFMA number i of 24 into buffer B (product-major inside a k-step; consecutive MFMAs hit different accumulators)
    auto mfma_s = [&](auto bc, int i, bool masked) {
        constexpr int B = decltype(bc)::value;
        const int st = i / 6, pr = (i % 6) >> 1, qt = i & 1;
        const f16x8 a = pr == 2 ? kl[st] : kh[st];
        const f16x8 bq = pr == 1 ? ql[qt][st] : qh[qt][st];
        if (i < 2) {
            if (masked) mfma_v_first(s[B][qt], a, bq, cmask);  // wave-uniform: the one partial tile of the sequence
            else mfma_v_first0(s[B][qt], a, bq);
        } else {
            mfma_v_acc(s[B][qt], a, bq);
        }
    };
    auto mfma_pv = [&](int i) {
        const int s2 = i / 12, dt = (i % 12) / 6, pr = (i % 6) >> 1, qt = i & 1;
        const f16x8 a = pr == 2 ? vl[s2][dt] : vh[s2][dt];
        const f16x8 bp = pr == 1 ? pl[qt][s2] : ph[qt][s2];
        o[qt][dt] = mfma16(a, bp, o[qt][dt]);
    };
    // ---- the vector work of an iteration as single-instruction micro-ops (placed by the generated schedule) ---------------------------
    // finish of S buffer B (tile kt): SB subtract the offset, EX exponential (in place), AD row sum, CV / ML / MH split of a register pair
    auto op_sb = [&](auto bc, int qt, int r) { constexpr int B = decltype(bc)::value; s[B][qt][r] -= m_off[qt]; };
    auto op_ex = [&](auto bc, int qt, int r) { constexpr int B = decltype(bc)::value; s[B][qt][r] = __builtin_amdgcn_exp2f(s[B][qt][r]); };
    auto op_ad = [&](auto bc, int qt, int r) { constexpr int B = decltype(bc)::value; l_run[qt][r & 3] += s[B][qt][r]; };
    auto op_cv = [&](auto bc, int qt, int pr) {
        constexpr int B = decltype(bc)::value;
        u32x4 a = __builtin_bit_cast(u32x4, ph[qt][pr >> 2]);
        a[pr & 3] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(s[B][qt][2 * pr], s[B][qt][2 * pr + 1]));
        ph[qt][pr >> 2] = __builtin_bit_cast(f16x8, a);
    };
    auto op_ml = [&](auto bc, int qt, int pr) {
        constexpr int B = decltype(bc)::value;
        u32x4 c = __builtin_bit_cast(u32x4, pl[qt][pr >> 2]);
        c[pr & 3] = mix_lo(__builtin_bit_cast(u32x4, ph[qt][pr >> 2])[pr & 3], s[B][qt][2 * pr]);
        pl[qt][pr >> 2] = __builtin_bit_cast(f16x8, c);
    };
    auto op_mh = [&](auto bc, int qt, int pr) {
        constexpr int B = decltype(bc)::value;
        u32x4 c = __builtin_bit_cast(u32x4, pl[qt][pr >> 2]);
        c[pr & 3] = mix_hi(c[pr & 3], __builtin_bit_cast(u32x4, ph[qt][pr >> 2])[pr & 3], s[B][qt][2 * pr + 1]);
        pl[qt][pr >> 2] = __builtin_bit_cast(f16x8, c);
    };
    // row maxima of S buffer B (tile kt+1): M1 i = max3 of registers 3i .. 3i+2, M2 0 / 1 = max3(t0, t1, t2) / max3(t3, t4, reg 15), M3, SW = other lane half
    float mt1[2][5], mt2[2][2], mx[2], mt[2];
    auto op_m1 = [&](auto bc, int qt, int i) { constexpr int B = decltype(bc)::value; mt1[qt][i] = max3(s[B][qt][3 * i], s[B][qt][3 * i + 1], s[B][qt][3 * i + 2]); };
    auto op_m2 = [&](auto bc, int qt, int j) {
        constexpr int B = decltype(bc)::value;
        mt2[qt][j] = j == 0 ? max3(mt1[qt][0], mt1[qt][1], mt1[qt][2]) : max3(mt1[qt][3], mt1[qt][4], s[B][qt][15]);
    };
    auto op_m3 = [&](int qt) { mx[qt] = max2(mt2[qt][0], mt2[qt][1]); };
    auto op_sw = [&](int qt) {  // the other lane half holds the other 16 keys of the same query: one v_permlane32_swap
        float x = mx[qt], y = mx[qt];
        half_swap(x, y);        // lanes 32-63 of x <-> lanes 0-31 of y: one of (x, y) is now the partner's value, the other the lane's own
        mt[qt] = max2(x, y);
    };
    // the (rare) raise of the running offset once the row maxima of S(kt+1) are known: S - m_off <= RESCALE_THR + P_SHIFT must hold
    // before the exponentials of that tile.  Only scalars per query, l and O change: nothing the S chains read or write.
    auto rescale = [&](bool force) {
        if (force) {  // first tile: the offset puts every query's maximum at P_SHIFT (l = 0, O = 0: nothing to scale)
            m_off[0] = mt[0] - P_SHIFT;
            m_off[1] = mt[1] - P_SHIFT;
        } else if (__any(__builtin_fmaxf(mt[0] - m_off[0], mt[1] - m_off[1]) > RESCALE_THR + P_SHIFT)) {
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                const float delta = __builtin_fmaxf(mt[qt] - m_off[qt] - P_SHIFT, 0.f);  // never lowered
                const float alpha = __builtin_amdgcn_exp2f(-delta);
                m_off[qt] += delta;
#pragma unroll
                for (int c = 0; c < 4; ++c) l_run[qt][c] *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    o[qt][0][r] *= alpha;
                    o[qt][1][r] *= alpha;
                }
            }
        }
    };

    // ---- prologue ------------------------------------------------------------------------------------------------------------
    // tile tau lives in slot tau % 3.  Request order K0 K1 V0 K2 V1 | K3 (after K0 has been read by every wave: same slot); iteration kt
    // then requests K(kt+4) and V(kt+2) and waits, at its top, for everything but the four pieces of iteration kt-1 (vmcnt retires in order)
    dma_k(0, 0, 0); dma_k(0, 0, 1); dma_k(1, 1, 0); dma_k(1, 1, 1); dma_v(0, 0, 0); dma_v(0, 0, 1);
    dma_k(2, 2, 0); dma_k(2, 2, 1); dma_v(1, 1, 0); dma_v(1, 1, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // K0, K1 landed
    __builtin_amdgcn_s_barrier();
    set_kad(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) read_kf(i);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave has read K(0): its slot may take K(3)
    dma_k(3, 0, 0); dma_k(3, 0, 1);
    const int klast = Lkv - (ntiles - 1) * KT;  // keys in the last tile
    if (klast < KT) {
#pragma unroll
        for (int r = 0; r < 16; ++r) cmask[r] = acc_row(r, lane) >= klast ? -__builtin_inff() : 0.f;
        asm volatile("s_nop 7" ::: "memory");  // VALU-written registers -> C operand of an asm MFMA (no compiler padding around asm)
    }
#pragma unroll
    for (int i = 0; i < 24; ++i) mfma_s(ic<0>{}, i, ntiles == 1 && klast < KT);
    set_kad(1);
#pragma unroll
    for (int i = 0; i < 8; ++i) read_kf(i);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // asm MFMA results -> VALU readers (hipcc pads nothing around asm)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
        for (int i = 0; i < 5; ++i) op_m1(ic<0>{}, qt, i);
        op_m2(ic<0>{}, qt, 0);
        op_m2(ic<0>{}, qt, 1);
        op_m3(qt);
        op_sw(qt);
    }
    rescale(true);

    // ---- main loop -----------------------------------------------------------------------------------------------------------
    // iteration kt: S buffer C holds S'(kt) (maxima checked); kh / kl hold K(kt+1); LDS holds V(kt) and K(kt+2) after the barrier
    int sl = 0;  // kt % 3: ring slot of tile kt
    auto iter = [&](auto cc, auto nx, int kt) {
        constexpr int C = decltype(cc)::value;
        constexpr bool NEXT = decltype(nx)::value != 0;  // a tile kt+1 exists
        const bool masked = NEXT && kt + 2 == ntiles && klast < KT;  // tile kt+1 is the last one and partial (wave-uniform)
        const int sl0 = sl, sl1 = sl == 2 ? 0 : sl + 1, sl2 = sl == 0 ? 2 : sl - 1;  // kt % 3, (kt + 1) % 3 = (kt + 4) % 3, (kt + 2) % 3
        sl = sl1;
        if (!(EXP & 2)) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // K(kt+2), V(kt) landed (mine); the requests of iteration kt-1 may fly
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const unsigned off_v = sl0 * (KT * 256), off_k = sl2 * (KT * 256);  // ring slots of V(kt) and K(kt+2), bytes
        PIN();
        // the instructions of a gap come from the generated tables: after unrolling every code is a constant
        auto run = [&](auto codec) {
            constexpr int code = decltype(codec)::value;
            constexpr int kind = code >> 8, a = (code >> 4) & 15, bb = code & 15;
            if constexpr (kind == sch::SB) { if (!(EXP & 1)) op_sb(cc, a, bb); }
            else if constexpr (kind == sch::EX) { if (!(EXP & 1)) op_ex(cc, a, bb); }
            else if constexpr (kind == sch::AD) { if (!(EXP & 1)) op_ad(cc, a, bb); }
            else if constexpr (kind == sch::CV) { if (!(EXP & 1)) op_cv(cc, a, bb); }
            else if constexpr (kind == sch::ML) { if (!(EXP & 1)) op_ml(cc, a, bb); }
            else if constexpr (kind == sch::MH) { if (!(EXP & 1)) op_mh(cc, a, bb); }
            else if constexpr (kind == sch::M1) { if (NEXT && !(EXP & 1)) op_m1(ic<1 - C>{}, a, bb); }
            else if constexpr (kind == sch::M2) { if (NEXT && !(EXP & 1)) op_m2(ic<1 - C>{}, a, bb); }
            else if constexpr (kind == sch::M3) { if (NEXT && !(EXP & 1)) op_m3(a); }
            else if constexpr (kind == sch::SW) { if (NEXT && !(EXP & 1)) op_sw(a); }
            else if constexpr (kind == sch::AV) { if (!(EXP & 8)) vad[bb] = va0[bb] + off_v; }
            else if constexpr (kind == sch::VR) { if (!(EXP & 8)) read_v(bb); }
            else if constexpr (kind == sch::AK) { if (NEXT && !(EXP & 8)) kad[bb] = ka0[bb] + off_k; }
            else if constexpr (kind == sch::KR) { if (NEXT && !(EXP & 8)) read_kf(bb); }
            else if constexpr (kind == sch::DM) { if (!(EXP & 2)) { if (bb < 2) dma_k(kt + 4, sl1, bb); else dma_v(kt + 2, sl2, bb - 2); } }
        };
        // phase 1: the 24 MFMAs of S(kt+1); phase 2: the 24 MFMAs of PV(kt)
        static_for<24>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if (NEXT && !(EXP & 4)) mfma_s(ic<1 - C>{}, g, masked);
            PIN();  // MFMA first: nothing of this gap may be scheduled above it
            static_for<sch::W>([&](auto i) { run(ic<sch::P1[g][decltype(i)::value]>{}); });
            PIN();
        });
        static_for<24>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if (!(EXP & 4)) mfma_pv(g);
            PIN();
            static_for<sch::W>([&](auto i) { run(ic<sch::P2[g][decltype(i)::value]>{}); });
            PIN();
        });
        if (NEXT && !(EXP & 1)) rescale(false);
        PIN();
    };

    int kt = 0;
#pragma unroll 1
    for (; kt + 2 < ntiles; kt += 2) {
        iter(ic<0>{}, ic<1>{}, kt);
        iter(ic<1>{}, ic<1>{}, kt + 1);
    }
    if (kt + 2 == ntiles) {
        iter(ic<0>{}, ic<1>{}, kt);
        iter(ic<1>{}, ic<0>{}, kt + 1);
    } else {
        iter(ic<0>{}, ic<0>{}, kt);
    }

    // ---- epilogue: O / l, split, quad-transposed SH8 stores (attention_f16.hip) ---------------------------------------------------
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    if (!(EXP & 16))
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const float lsum = (l_run[qt][0] + l_run[qt][1]) + (l_run[qt][2] + l_run[qt][3]);
        const float inv = 1.f / (lsum + __shfl_xor(lsum, 32));
        const int q = q0 + wu * QW + qt * 32 + i32;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            f32x16 w;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a0 = o[qt][dt][e] * inv, b0 = o[qt][dt][8 + e] * inv, a1 = o[qt][dt][4 + e] * inv, b1 = o[qt][dt][12 + e] * inv;
                half_swap(a0, b0);
                half_swap(a1, b1);
                w[e] = a0; w[4 + e] = b0; w[8 + e] = a1; w[12 + e] = b1;
            }
            const f32x8 g0 = {w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7]};
            const f32x8 g1 = {w[8], w[9], w[10], w[11], w[12], w[13], w[14], w[15]};
            f16x8 hi0, lo0, hi1, lo1;
            split8(g0, hi0, lo0);
            split8(g1, hi1, lo1);
            u32x4 p0 = __builtin_bit_cast(u32x4, hi0), p1 = __builtin_bit_cast(u32x4, lo0);
            u32x4 p2 = __builtin_bit_cast(u32x4, hi1), p3 = __builtin_bit_cast(u32x4, lo1);
            const int ql4 = i32 & 3;
            quad_transpose(p0, p1, p2, p3, ql4);
            const int qq = q - ql4;
            u32x4* ob = reinterpret_cast<u32x4*>(p.out) + ((head * 64 + dt * 32 + 16 * h) >> 2) + ql4;
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx)
                if (qq + sidx < Lq) ob[(qbase + qq + sidx) * (p.ldo >> 2)] = sidx == 0 ? p0 : sidx == 1 ? p1 : sidx == 2 ? p2 : p3;
        }
    }
    // the ring is reused by the next block: every request of this one has landed (tiles past the end too) and every wave is done reading
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    }  // slot loop
}

}  // namespace

void launch_attention_f16x3_pw(const AttnF16Args& a, hipStream_t s) {
    const int L = a.M > a.N ? a.M : a.N;
    const int nqb = (L + QB - 1) / QB, groups = a.heads * 2 * a.B;
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu < 8) n_cu = 256;
    }
    const int nslots = ((groups + 7) / 8) * nqb;              // per XCD
    dim3 grid(8 * (nslots < n_cu / 8 ? nslots : n_cu / 8));   // one workgroup per CU at most (512 registers per wave: one workgroup fits)
    switch (a.variant >= 16 ? a.variant - 16 : 0) {
        case 0: hipLaunchKernelGGL(attention_f16x3_pw_kernel<0>, grid, dim3(NWV * 64), 0, s, a); break;
#define MG_ATTN_EXP_CASE(N) case N: hipLaunchKernelGGL(attention_f16x3_pw_kernel<N>, grid, dim3(NWV * 64), 0, s, a); break;
        MG_ATTN_EXP_CASE(1) MG_ATTN_EXP_CASE(2) MG_ATTN_EXP_CASE(3) MG_ATTN_EXP_CASE(4) MG_ATTN_EXP_CASE(5) MG_ATTN_EXP_CASE(8) MG_ATTN_EXP_CASE(9) MG_ATTN_EXP_CASE(10) MG_ATTN_EXP_CASE(11) MG_ATTN_EXP_CASE(16) MG_ATTN_EXP_CASE(27)
        default: hipLaunchKernelGGL(attention_f16x3_pw_kernel<0>, grid, dim3(NWV * 64), 0, s, a); break;
    }
}
