// Weight-stationary Linear for K = 256 on split-fp16 operands: the QKV projections of LightGlue (lightglue.py:144-152 self, :189-194
// cross) -- x (T, 256) . W^T (N, 256) + b, rotary on the q / k columns, the soft-max scale on the q (qk) columns, SH8 output.
//
// Why.  gemm_f16.hip runs these launches at 31 % MFMA-busy: with K = 256 a 256 x 128 tile is 16 stages of MFMAs followed by an epilogue
// of ~15 vector instructions per output value (bias, rotary, scale, split, the quad transpose for full-line stores) and 201 MB of
// stores, and the three add up -- two co-resident workgroups stay in lockstep.  Here
//   * a wave keeps ITS weights in registers for the whole kernel: 32 features x 256 k x (hi, lo) = 128 registers (the accumulator
//     file: MFMA A operands may live there), loaded once; a persistent workgroup of 4 waves (one per SIMD, 512 registers each) owns
//     128 features and streams 128-token tiles past them -- the only LDS traffic is the token stream, 32 KB per 48 MFMAs of a wave;
//   * the epilogue of tile t-1 is issued, one instruction at a time, in the shadows of the MFMAs of tile t (second accumulator set;
//     the same machinery as attention_pw_f16.hip: a generated table says which instruction follows which MFMA, gen_gemm_ws_sched.py);
//   * tokens arrive by LDS-DMA written as asm (3-slot ring of 32-KB quarter tiles, counted waits), rotary tables and output rows go
//     through buffer instructions whose range check drops the rows past M: every wave issues the same number of vector-memory
//     operations in every stage, so the two s_waitcnt of a stage are constants (3-slot ring 96 KB + rotary slots 32 KB + output staging 16 KB of LDS).
// The arithmetic is the generic kernel's, operation for operation (fma(acc, winv, bias); rotary as mul, mul, add; scale; hi = RTN
// fp16, lo = fp16(v - hi)): the outputs are bit-identical to gemm_f16x3_kernel (tests/test_gpu_gemm_ws.py), which stays the kernel of
// every other Linear and of shapes this one does not take (launch_gemm_f16x3 decides).
#include <type_traits>
#include <utility>

#include "common.h"
#include "kernels.h"

namespace {

#define PIN() __builtin_amdgcn_sched_barrier(0)

constexpr int TOK = 128, FEAT = 128, KK = 256, QK = 64;  // tokens per tile; features per workgroup; contraction; k per stage
constexpr int SLOT = TOK * QK * 4;                       // bytes of a quarter tile in LDS: 128 rows x 256 B
constexpr int RD = 3;                                    // ring depth: a quarter tile is requested RD - 1 stages before the barrier that publishes it (4 measured the same)

template <int N>
using ic = std::integral_constant<int, N>;
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(ic<I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) f16x8 lds_f16x8;

__device__ __forceinline__ i32x4 make_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}
// LDS-DMA piece / register load / store as asm: none of them is visible to hipcc's s_waitcnt insertion (a visible load would make it
// drain vmcnt in front of the first use, DMA pieces included); completion is awaited by the counted waits of the stage.
__device__ __forceinline__ void dma_piece(const i32x4 rsrc, unsigned lds_byte, int voffset) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_byte), "v"(voffset), "s"(rsrc) : "memory");
}
// (no immediate offset on an LDS-DMA piece: the instruction adds it to the LDS address as well as to the memory address)
__device__ __forceinline__ void store16(const u32x4 v, const i32x4 rsrc, int voffset) {
    asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(v), "v"(voffset), "s"(rsrc) : "memory");
}
__device__ __forceinline__ unsigned mix_lo(unsigned hi, float x0) {
    unsigned lo;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(x0));
    return lo;
}
__device__ __forceinline__ unsigned mix_hi(unsigned lo, unsigned hi, float x1) {
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(x1));
    return lo;
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace gsch {
enum { NOP = 0, BR, DM, LE, ER, BF, RA, RB, RC, RD_, RE, RF, SC, HC, LA, LB, WR, RO, ST };  // = KINDS of the generator
#include "gemm_ws_sched.inc"
}  // namespace gsch

struct WsArgs {
    const void* A;      // SH8 tokens (M, 256)
    const void* W;      // tiled [16][Npad][64 B]
    int Npad;
    float winv;
    const float* bias;
    void* Cs;           // SH8 (M, ldcs)
    int ldcs;
    int M, N;
    const float* enc;   // (M, 64): cos[32] | sin[32] per token, or null
    int rotary_cols;
    float scale;
    int scale_cols;
    int grp_wgs[8];     // persistent workgroups per XCD of feature group g (N / 128 <= 8 groups): more for the groups with the longer epilogue
};

template <int ROT, int SCL>
__device__ __forceinline__ void ws_body(const WsArgs& p, u32x4* s_ring, u32x4* s_enc, u32x4* s_out, int grp, int lane_in_grp, int per_grp, int xcd) {
    const int t = threadIdx.x, lane = t & 63;
    const int wu = __builtin_amdgcn_readfirstlane(t >> 6);
    const int i32 = lane & 31, h = lane >> 5;
    const int fb0 = grp * FEAT + 32 * wu;  // first feature of this wave (wave-uniform)
    const int fb = fb0 + 16 * h;           // first of the lane's 16 consecutive features

    // ---- weights into registers.  MFMA row m = (r & 3) + 8 (r >> 2) + 4 hh holds the lane-half-hh accumulator register r; giving that row
    // feature 16 hh + r makes a lane's 16 accumulator registers 16 CONSECUTIVE features of its token (no permlane swaps in the epilogue)
    const int frow = fb0 + 16 * ((i32 >> 2) & 1) + ((i32 & 3) + 4 * (i32 >> 3));
    const u32x4* Wg = reinterpret_cast<const u32x4*>(p.W);
    f16x8 wh[16], wl[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        const u32x4* wp = Wg + ((long long)ks * p.Npad + frow) * 4 + 2 * h;
        wh[ks] = __builtin_bit_cast(f16x8, wp[0]);
        wl[ks] = __builtin_bit_cast(f16x8, wp[1]);
    }
    float bias[16];
#pragma unroll
    for (int j = 0; j < 16; j += 4) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + fb + j);
        bias[j] = bv.x; bias[j + 1] = bv.y; bias[j + 2] = bv.z; bias[j + 3] = bv.w;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the compiler-visible loads above: done before the asm-counted region starts

    // ---- token stream -------------------------------------------------------------------------------------------------------
    const i32x4 rs_a = make_rsrc(p.A, (unsigned)p.M * 1024u);
    const i32x4 rs_c = make_rsrc(p.Cs, (unsigned)p.M * (unsigned)p.ldcs * 4u);
    const i32x4 rs_e = make_rsrc(p.enc, ROT ? (unsigned)p.M * 256u : 0u);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)reinterpret_cast<char*>(s_ring);
    // DMA piece j of 32 of a quarter tile: rows 4j .. 4j+3, LDS slot (row, pos) <- source piece pos ^ (row & 15); wave w moves j = 8w .. 8w+7
    int vo[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row = (8 * wu + j) * 4 + (lane >> 4), pos = lane & 15;
        vo[j] = row * 1024 + ((pos ^ (row & 15)) << 4);
    }
    // tiles of this workgroup: XCD x owns tiles x, x + 8, ...; inside the XCD the `per_grp` workgroups of a feature group take them round-robin
    const int ntiles = (p.M + TOK - 1) / TOK;
    auto tile_of = [&](int i) { return xcd + 8 * (lane_in_grp + i * per_grp); };  // >= ntiles: past the end (every access out of range)
    int n_my = 0;
    while (tile_of(n_my) < ntiles) ++n_my;
    // stage sigma = 4 i + qs of this workgroup's stream: quarter qs of its i-th tile, ring slot sigma % 3
    auto dma_stage = [&](int sigma, int slot, int j) {
        const int i = sigma >> 2, qs = sigma & 3;
        const long long tile = tile_of(i);
        const long long off = tile * (TOK * 1024) + qs * 256;
        const int o32 = off > 0x7fffff00ll ? 0x7fffff00 : (int)off;  // past 2 GiB: out of range anyway (M x 1 KiB < 2 GiB is checked by the launcher)
        dma_piece(rs_a, __builtin_amdgcn_readfirstlane(lds0 + slot * SLOT + (8 * wu + j) * 1024), vo[j] + o32);
    };
    // fragment addresses.  Row 32 tb + i32 of a slot, piece ((2 u + h) * 2 + hl) ^ (row & 15): the XOR only touches the piece bits, so
    // the address is (slot + row i32 base) | X[u][hl] + tb * 8 KiB.  X is a lane constant (8 registers for the whole kernel), the slot base is
    // added once per stage (8 vector adds: fcur for this stage's slot, fnext[hl] for k-step 0 of the next one) and the token block is the
    // instruction's immediate offset -- no address arithmetic per read (it was an xor and an add for each of the 32 reads of a stage)
    unsigned fx[4][2], fcur[4][2], fnext[2];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int hl = 0; hl < 2; ++hl) fx[u][hl] = lds0 + i32 * 256 + (((((2 * u + h) * 2 + hl) ^ (i32 & 15))) << 4);
    f16x8 bh[2][4], bl[2][4];  // [k-step parity][token block]
    auto read_b = [&](int u, int tb, int hl) {  // u = 0: k-step 0 of the NEXT stage (fnext), else k-step u of this stage (fcur)
        const unsigned a = (u == 0 ? fnext[hl] : fcur[u][hl]) + tb * 8192;
        const f16x8 v = *(const lds_f16x8*)(size_t)a;
        if (hl) bl[u & 1][tb] = v;
        else bh[u & 1][tb] = v;
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) { acc[0][tb][r] = 0.f; acc[1][tb][r] = 0.f; }

    // ---- epilogue state (one 32-token block at a time) ----------------------------------------------------------------------------
    float v[16], t0[8], t1[8], t2[8], t3[8];
    unsigned P[4][4];
    u32x4 ro[4];
    // rotary tables of a block: four 16-byte pieces per lane (cos 0-3, cos 4-7, sin 0-3, sin 4-7 of its token and its 8 pairs), moved by
    // LDS-DMA into slots PRIVATE to the wave ([ring slot][wave][piece][lane]) with the token pieces of their stage and read back by the lane that asked
    f32x4 encr[4];
    const unsigned enc0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)reinterpret_cast<char*>(s_enc) + wu * 4096;
    typedef __attribute__((address_space(3))) f32x4 lds_f32x4;
    // output staging (this wave's 4 KB): the lane writes its piece q = 0..3 to slot 4 h + q of token i32 (slot index XOR (token & 7): the
    // eight lanes of a ds_write_b128 group -- eight consecutive tokens, 128 B apart -- then hit eight different 16-byte columns), and
    // reads back slot (lane & 7) of token (lane >> 3) + 8 it: eight consecutive lanes hold the 128 contiguous bytes of one token
    const unsigned out0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)reinterpret_cast<char*>(s_out) + wu * 4096;
    unsigned wa[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) wa[q] = out0 + i32 * 128 + (((4 * h + q) ^ (i32 & 7)) << 4);
    const unsigned ra = out0 + (lane >> 3) * 128 + (((lane & 7) ^ ((lane >> 3) & 7)) << 4);  // + it * 1024
    typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
    const int enc_col = ROT ? (((fb & 63) >> 1) << 2) : 0;  // byte offset of the lane's first frequency inside a 256-byte enc row

    // one stage: MFMAs of (tile parity CUR, quarter QS) of tile `tile_cur` + epilogue of block QS of the previous tile (token row base
    // `tokp`; -1: none, its stores and loads are sent out of range) + rotary loads for the NEXT epilogue block (token row base `tokn`)
    auto stage = [&](auto curc, auto qsc, auto mmac, int sigma, int tokp) {
        constexpr int CUR = decltype(curc)::value, QS = decltype(qsc)::value;
        constexpr bool MMA = decltype(mmac)::value != 0;
        const unsigned off_cur = (unsigned)(sigma % RD) * SLOT, off_next = (unsigned)((sigma + 1) % RD) * SLOT;
        const int slot_free = sigma % RD;
        if (MMA) {
#pragma unroll
            for (int u = 1; u < 4; ++u) { fcur[u][0] = fx[u][0] + off_cur; fcur[u][1] = fx[u][1] + off_cur; }
            fnext[0] = fx[0][0] + off_next;
            fnext[1] = fx[0][1] + off_next;
        }
        // the packet requested in this stage belongs to stage sigma + RD: its token pieces, and the rotary pieces of the block whose epilogue
        // runs there -- block (sigma + RD) & 3 of the tile before tile (sigma + RD) >> 2 of this workgroup (none for its first tile)
        const int in = (sigma + RD) >> 2, qn = (sigma + RD) & 3;
        const int tokn = in > 0 ? tile_of(in - 1) * TOK + 32 * qn : -1;
        const int st_off = tokp >= 0 ? ((tokp + 32 * QS + (lane >> 3)) * p.ldcs + fb0) * 4 + 16 * (lane & 7) : (int)0x80000000;  // token (lane >> 3) [+ 8 it], piece lane & 7
        const int en_off = tokn >= 0 ? (tokn + i32) * 256 + enc_col : (int)0x80000000;
        auto run = [&](auto codec) {
            constexpr int code = decltype(codec)::value;
            constexpr int kind = code >> 8, a = (code >> 4) & 15, b = code & 15;
            if constexpr (kind == gsch::BR) { if (MMA) read_b(a, b >> 1, b & 1); }
            else if constexpr (kind == gsch::DM) dma_stage(sigma + RD, slot_free, b);
            else if constexpr (kind == gsch::LE) {
                const unsigned dst = __builtin_amdgcn_readfirstlane(enc0 + slot_free * 16384 + b * 1024);
                dma_piece(rs_e, dst, en_off + (b == 0 ? 0 : b == 1 ? 16 : b == 2 ? 128 : 144));  // cos 0-3, cos 4-7, sin 0-3, sin 4-7
            }
            else if constexpr (kind == gsch::ER) encr[b] = *(const lds_f32x4*)(size_t)(enc0 + slot_free * 16384 + b * 1024 + lane * 16);
            else if constexpr (kind == gsch::BF) v[b] = fma_winv(acc[1 - CUR][QS][b], p.winv, bias[b]);
            else if constexpr (kind == gsch::RA) t0[b] = __fmul_rn(v[2 * b], encr[b >> 2][b & 3]);
            else if constexpr (kind == gsch::RB) t1[b] = __fmul_rn(-v[2 * b + 1], encr[2 + (b >> 2)][b & 3]);
            else if constexpr (kind == gsch::RC) t2[b] = __fmul_rn(v[2 * b + 1], encr[b >> 2][b & 3]);
            else if constexpr (kind == gsch::RD_) t3[b] = __fmul_rn(v[2 * b], encr[2 + (b >> 2)][b & 3]);
            else if constexpr (kind == gsch::RE) v[2 * b] = __fadd_rn(t0[b], t1[b]);
            else if constexpr (kind == gsch::RF) v[2 * b + 1] = __fadd_rn(t2[b], t3[b]);
            else if constexpr (kind == gsch::SC) v[b] *= p.scale;
            else if constexpr (kind == gsch::HC) {
                // the values as ROUNDED fp32 registers: with the mix instructions enabled for this file (the asm below needs them) hipcc would
                // otherwise fuse the last multiply / add into the conversion (v_fma_mixlo_f16: one rounding of the exact result) and the hi
                // half would differ from split8's in the double-rounding cases (hazard 2; first run of tests/test_gpu_gemm_ws.py)
                asm volatile("" : "+v"(v[2 * b]), "+v"(v[2 * b + 1]));
                const f32x2 pr = {v[2 * b], v[2 * b + 1]};
                P[2 * (b >> 2)][b & 3] = __builtin_bit_cast(unsigned, __builtin_convertvector(pr, h16x2));  // v_cvt_pk_f16_f32: RTN, as split8
            }
            else if constexpr (kind == gsch::LA) P[2 * (b >> 2) + 1][b & 3] = mix_lo(P[2 * (b >> 2)][b & 3], v[2 * b]);
            else if constexpr (kind == gsch::LB) P[2 * (b >> 2) + 1][b & 3] = mix_hi(P[2 * (b >> 2) + 1][b & 3], P[2 * (b >> 2)][b & 3], v[2 * b + 1]);
            else if constexpr (kind == gsch::WR) *(lds_u32x4*)(size_t)wa[b] = u32x4{P[b][0], P[b][1], P[b][2], P[b][3]};
            else if constexpr (kind == gsch::RO) ro[b] = *(const lds_u32x4*)(size_t)(ra + b * 1024);
            else if constexpr (kind == gsch::ST) store16(ro[b], rs_c, st_off + b * 8 * p.ldcs * 4);
        };
        static_for<48>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            constexpr int u = g / 12, pr = (g % 12) >> 2, tb = g & 3, ks = 4 * QS + u;
            if constexpr (g == gsch::BARRIER_GAP) {
                // the quarter tile of the next stage has landed (mine: everything but the previous stage's operations and this stage's
                // LE + ST), every wave has finished reading this stage's slot: it takes the data of the stage three ahead
                // operations of this wave younger than the pieces of stage sigma + 1: (RD - 2) full stages (16 with rotary pieces, 12 without:
                // LE 4, ST 4, DM 8) + this stage's LE + ST; in the first RD - 2 stages of the workgroup the prologue's requests stand in for them
                // younger than the packet of stage sigma + 1 (requested RD - 1 = 2 stages ago): the previous stage's stores and packet, this
                // stage's stores; in the workgroup's first stage the prologue's last packet stands in for the previous stage
                constexpr int PK = ROT ? 12 : 8;
                static_assert(RD == 3, "the wait constants below are written out for a ring of three");
                if (sigma == 0) wait_vm<PK + 4>();
                else wait_vm<PK + 8>();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                PIN();
            }
            if constexpr (MMA) {
                const f16x8 aw = pr == 2 ? wl[ks] : wh[ks];
                const f16x8 bx = pr == 1 ? bl[u & 1][tb] : bh[u & 1][tb];
                if constexpr (ks == 0 && pr == 0) {  // first MFMA of a tile's chain: C = 0 (the set's previous tile was drained a tile ago)
                    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    acc[CUR][tb] = mfma16(aw, bx, zero);
                } else {
                    acc[CUR][tb] = mfma16(aw, bx, acc[CUR][tb]);
                }
            }
            PIN();
            static_for<gsch::W>([&](auto i) {
                if constexpr (ROT && SCL) run(ic<gsch::T11[g][decltype(i)::value]>{});
                else if constexpr (ROT) run(ic<gsch::T10[g][decltype(i)::value]>{});
                else if constexpr (SCL) run(ic<gsch::T01[g][decltype(i)::value]>{});
                else run(ic<gsch::T00[g][decltype(i)::value]>{});
            });
            PIN();
        });
    };

    // ---- prologue: stages 0, 1, 2 requested; stage 0 landed; its k-step-0 fragments read ----------------------------------------------
    // packets of stages 0 .. RD-1 (their rotary pieces belong to a tile before the first: out of range, zeros -- issued for the count only)
#pragma unroll
    for (int sgm = 0; sgm < RD; ++sgm) {
#pragma unroll
        for (int j = 0; j < 8; ++j) dma_stage(sgm, sgm, j);
        if (ROT) {
#pragma unroll
            for (int b = 0; b < 4; ++b) dma_piece(rs_e, __builtin_amdgcn_readfirstlane(enc0 + sgm * 16384 + b * 1024), (int)0x80000000);
        }
    }
    wait_vm<(ROT ? 12 : 8) * (RD - 1)>();
    __builtin_amdgcn_s_barrier();
    fnext[0] = fx[0][0];
    fnext[1] = fx[0][1];
#pragma unroll
    for (int tb = 0; tb < 4; ++tb) { read_b(0, tb, 0); read_b(0, tb, 1); }

    // ---- tile loop: tile i on accumulator set i & 1, epilogue of tile i - 1 behind its MFMAs ---------------------------------------------
    auto tile = [&](auto curc, auto mmac, int i) {
        const int tprev = i > 0 ? tile_of(i - 1) * TOK : -1;
        stage(curc, ic<0>{}, mmac, 4 * i + 0, tprev);
        stage(curc, ic<1>{}, mmac, 4 * i + 1, tprev);
        stage(curc, ic<2>{}, mmac, 4 * i + 2, tprev);
        stage(curc, ic<3>{}, mmac, 4 * i + 3, tprev);
    };
    int i = 0;
#pragma unroll 1
    for (; i + 1 < n_my; i += 2) {
        tile(ic<0>{}, ic<1>{}, i);
        tile(ic<1>{}, ic<1>{}, i + 1);
    }
    if (i < n_my) {
        tile(ic<0>{}, ic<1>{}, i);
        tile(ic<1>{}, ic<0>{}, i + 1);  // drain: the epilogue of the last tile, no MFMAs
    } else if (n_my > 0) {
        tile(ic<0>{}, ic<0>{}, i);
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
}

__global__ __launch_bounds__(256) void gemm_ws_f16x3_kernel(const WsArgs p) {
    __shared__ __attribute__((aligned(1024))) u32x4 s_ring[RD * SLOT / 16];
    __shared__ __attribute__((aligned(1024))) u32x4 s_enc[RD * 4 * 4 * 64];  // rotary pieces: [ring slot][wave][piece][lane]
    __shared__ __attribute__((aligned(1024))) u32x4 s_out[4 * 32 * 8];      // output staging: [wave][token of the block][8 pieces], swizzled
    const int ng = p.N / FEAT;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    int grp = 0, lane_in_grp = local;  // the XCD's workgroups are dealt to the feature groups in blocks of grp_wgs[g]
    while (grp + 1 < ng && lane_in_grp >= p.grp_wgs[grp]) lane_in_grp -= p.grp_wgs[grp++];
    const int per_grp = p.grp_wgs[grp];
    const int fb0 = grp * FEAT;
    const bool rot = p.enc != nullptr && fb0 < p.rotary_cols, scl = fb0 < p.scale_cols;  // workgroup-uniform (the launcher checks the multiples)
    if (rot && scl) ws_body<1, 1>(p, s_ring, s_enc, s_out, grp, lane_in_grp, per_grp, xcd);
    else if (rot) ws_body<1, 0>(p, s_ring, s_enc, s_out, grp, lane_in_grp, per_grp, xcd);
    else if (scl) ws_body<0, 1>(p, s_ring, s_enc, s_out, grp, lane_in_grp, per_grp, xcd);
    else ws_body<0, 0>(p, s_ring, s_enc, s_out, grp, lane_in_grp, per_grp, xcd);
}

}  // namespace

// true if the launch was taken
bool launch_gemm_ws_f16x3(const GemmF16Args& a, hipStream_t s) {
    if (a.K != KK || a.A1 || a.w_rowmajor || a.l2norm || a.C || !a.Cs || a.relu || a.post_scale || a.post_shift || a.residual || a.batch != 1 || !a.bias) return false;
    if (a.N % FEAT || a.lda0 != KK || a.M <= 0 || (long long)a.M * 1024 >= 0x7fffff00ll || (long long)a.M * a.ldcs * 4 >= 0x7fffff00ll) return false;
    if ((a.enc && a.rotary_cols % FEAT) || a.scale_cols % FEAT) return false;
    const int ng = a.N / FEAT;
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu < 8) n_cu = 256;
    }
    if (ng > 8) return false;
    // Workgroups per XCD and feature group.  A group's work per tile is its instruction count per stage (static counts of the four
    // variants: no rotary / no scale 276, scale 290, rotary 357, rotary + scale 372): the q / k groups of the self projection get more
    // workgroups than its v groups, so that every workgroup of a launch finishes at about the same time (largest-remainder split of the
    // XCD's CUs; with equal shares the q workgroups ran 13 tiles of the longest kind while the v workgroups idled for a quarter of the launch)
    const int ntiles = (a.M + TOK - 1) / TOK, tiles_per_xcd = (ntiles + 7) / 8, cus = n_cu / 8;
    if (cus < ng) return false;
    int cnt[8], cost[8], total = 0, used = 0;
    for (int g = 0; g < ng; ++g) {
        const bool rot = a.enc && g * FEAT < a.rotary_cols, scl = g * FEAT < a.scale_cols;
        cost[g] = rot ? (scl ? 372 : 357) : (scl ? 290 : 276);
        total += cost[g];
    }
    for (int g = 0; g < ng; ++g) { cnt[g] = cus * cost[g] / total; if (cnt[g] < 1) cnt[g] = 1; used += cnt[g]; }
    while (used < cus) {  // hand the remaining CUs to the groups with the most work per workgroup
        int best = 0;
        for (int g = 1; g < ng; ++g) if ((long long)cost[g] * cnt[best] > (long long)cost[best] * cnt[g]) best = g;
        ++cnt[best]; ++used;
    }
    while (used > cus) { int big = 0; for (int g = 1; g < ng; ++g) if (cnt[g] > cnt[big]) big = g; --cnt[big]; --used; }
    for (int g = 0; g < ng; ++g) if (cnt[g] > tiles_per_xcd) { used -= cnt[g] - tiles_per_xcd; cnt[g] = tiles_per_xcd > 0 ? tiles_per_xcd : 1; }  // no workgroups without tiles
    used = 0;
    for (int g = 0; g < ng; ++g) used += cnt[g];
    const int per_xcd = used;
    WsArgs w;
    w.A = a.A0; w.W = a.W; w.Npad = a.Npad; w.winv = a.winv; w.bias = a.bias; w.Cs = a.Cs; w.ldcs = a.ldcs; w.M = a.M; w.N = a.N;
    w.enc = a.enc; w.rotary_cols = a.enc ? a.rotary_cols : 0; w.scale = a.scale; w.scale_cols = a.scale_cols;
    for (int g = 0; g < 8; ++g) w.grp_wgs[g] = g < ng ? cnt[g] : 0;
    hipLaunchKernelGGL(gemm_ws_f16x3_kernel, dim3(8 * per_xcd), dim3(256), 0, s, w);
    return true;
}
