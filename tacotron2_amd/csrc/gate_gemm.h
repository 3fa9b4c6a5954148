// gate_gemm.h -- WaveGlow's two big contractions as tiled GEMMs on an LDS-DMA ring (gfx950 only).
//
// gate_gemm_kernel:   acts[n][c] = tanh(t) * sigmoid(s),   (t, s) = rows (c, C + c) of
//                     in_layers[i](H)[.][n] + cond_layer(spect)[2C i + .][n]              waveglow/glow.py:33-40,164-171
//                     (+ in its epilogue this layer's term of the folded skip path, waveglow.hip waveglow_build)
// res_gemm_kernel:    H[n][.] += res_skip_layers[i][:C] acts[n][.] + bias                  waveglow/glow.py:168-173
//
// GEMM view: D[m][n] = sum_k W[m][k] X[n][k];  N = time steps of one sequence (tiles of 256); for the gated conv M = 2C
// rows in (tanh, sigmoid) 16-row tile pairs (blocks of 256) and K = ktaps x C channels of the residual stream H (tap j
// reads row n + (j - (ktaps-1)/2) dil, zero outside [0, len)) followed by the n_mel x n_group channels of the upsampled
// mel; for the residual conv M = C (blocks of 128, or of 256 in res_gemm256_kernel) and K = C gated channels.  Both operands are bfloat16 in memory (the
// weights [M][K] row-major, activations channels-last), so a K tile of either one is rows of 128 contiguous bytes and
// travels global -> LDS by `global_load_lds_dwordx4` without touching a register.
//
// Block: 8 waves as 2 (M) x 4 (N); a wave owns (16 MTW) x 64 of the tile = MTW x 4 accumulator tiles of
// v_mfma_f32_16x16x32_bf16 (MTW = 8: 128 registers; MTW = 4: 64).  Results leave through an LDS tile in 16-byte units,
// by non-temporal stores (no launch reads its own output).
// LDS image: the 16-byte chunk c of row r sits at chunk position c ^ ((r >> 1) & 7) of the row.  The DMA writes
// lane-linear (base + 16 lane), so the permutation is applied to each lane's SOURCE address; the fragment reads apply
// the same XOR.  With it the four 16-lane groups of a ds_read_b128 (MI355X LDS: {0-3,12-15,20-27}, ...) each touch
// 16 distinct 16-byte slots of the 256-byte bank row: conflict-free.
//
// K loop.  A wave's tile splits into quadrants (m half, n half); fragments of one m half and of the two n halves live
// in registers.  The operands arrive in HALF-STAGES -- for one K tile of 64, the A rows of every wave's m half s and
// the B rows of every wave's n half s (MTW = 8: 32 KiB) -- through a ring of five slots (all 160 KiB of LDS): one is
// being read, one is next, the others are in flight.  The two wave rows are two GROUPS that run one barrier apart (waves
// w and w + 4 share a SIMD): every phase is a load part (DMA requests of the half-stage under the cursor, then this
// phase's fragment reads) and a multiply part (16 MFMAs), each closed by an s_barrier, so on every SIMD one wave
// multiplies while the other loads:
//     phase 0: request A rows of (t+2, 0);  read A0 (t, 0);  wait: own pieces of (t, 1) landed  | MFMA (A0, B0)
//     phase 1: request B rows of (t+2, 0);  read B1 (t, 1)                                       | MFMA (A0, B1)
//     phase 2: request A rows of (t+2, 1);  read A1 (t, 1);  wait: own pieces of (t+1, 0) landed | MFMA (A1, B0)
//     phase 3: request B rows of (t+2, 1);  read B0 (t+1, 0)                                     | MFMA (A1, B1)
// A wait is a counted vmcnt (the wave issues its pieces in a fixed order) and sits one load part before the first read
// of that half-stage, so that between the two lies a barrier BOTH groups have passed; a slot is refilled two barriers
// after its last read.  Without its DMA requests this loop runs at 94 % of the MFMA rate; with them it is paced by their
// issue (~100 cycles each): DESIGN.md section 4 has the measurements that shaped it.
#pragma once
#include "conv_mfma.h"

namespace mv {

struct GateGemmParams {
    const unsigned short* w;     // [M][K] bfloat16
    const float* bias;           // [M] (in_layer bias + cond_layer bias), same row order
    const unsigned short* h;     // [B][L][h_stride] bfloat16: operand copy of the residual stream (Cp channels used)
    const unsigned short* cond;  // [B][L][ncond] bfloat16: upsampled mel, regrouped
    unsigned short* out;         // [B][L][Cp] bfloat16: gated activations
    const unsigned short* zero;  // max(Cp, ncond) + 64 zero bfloat16 values, 16-byte aligned, ABOVE h and cond and less than
                                 // 4 GiB past their bases: a row outside its sequence is read from here through the same
                                 // scalar base (tensor + K chunk) as the real rows
    const int32_t* lengths;      // frames per sequence (NULL: Tmax)
    int Tmax, spf;               // time steps per frame
    long long L;                 // rows per sequence in h / cond / out (= Tmax * spf)
    int Cp, ncond, ktaps, dil;
    int h_stride;                // elements per row of h (>= Cp)
    int K;                       // ktaps * Cp + ncond
    int nx, ny;                  // N tiles per sequence, M blocks (= 2C / 256)
    // Folded skip path (waveglow.hip, waveglow_build): this layer's contribution (W_end W_skip_i) acts_i to the end
    // conv, formed here from the block's gated tile while it sits in LDS.  fold_w = NULL: not folded here.
    const unsigned short* fold_w;  // [2 (hi, lo)][16][Cp] bfloat16: the [fold_nout][C] fp32 matrix split in two bf16
    float* fold_out;               // [ny][B * L][8] float: one partial sum per M block (wg_tail_kernel adds them)
    long long fold_part_stride;    // elements between two partials (= B * L * 8)
    int fold_nout;                 // 2 * n_half <= 8 rows used
    int fold_accumulate;           // 0: store (first layer of a flow), 1: add
    // res_gemm_kernel (the residual 1x1 conv, glow.py:168-173 with the skip half folded away): H += W acts + bias.
    // There `h` is the gated activations [B][L][Cp] (ktaps = 1, ncond = 0, K = Cp), `out` is unused, and:
    void* res_h;                   // [B][L][res_stride] float16: the residual stream, updated in place
    unsigned short* res_hb;        // same indexing, bfloat16: its MFMA-operand copy (what the next gated conv reads)
    int res_stride;
};

constexpr int kGgTile = 256, kGgBK = 64;              // N tile (time steps), K tile
constexpr int kGgHalfB = 128 * kGgBK * 2;             // the B rows of a half-stage: 128 rows x one K tile = 16 KiB
constexpr int kGgSlots = 5;
// MTW = 16-row M tiles per wave: 8 for the gated conv (block = 256 gate rows), 4 for the residual conv (128 rows)
template <int MTW>
struct GgGeom {
    static constexpr int BM = 32 * MTW;              // rows of M per block (2 wave rows)
    static constexpr int HM = MTW / 2;               // M tiles of a wave's m half
    static constexpr int RW = HM * 16;               // A rows per wave row in a half-stage
    static constexpr int NA = MTW / 4;               // A pieces (8 rows x 128 B) per wave and half-stage
    static constexpr int HalfA = 2 * RW * kGgBK * 2; // the A rows of a half-stage
    static constexpr int Slot = HalfA + kGgHalfB;    // a half-stage: A part + B part
    static constexpr int Lds = kGgSlots * Slot;      // MTW = 8: 160 KiB, the whole LDS of a CU
    static constexpr int PerHalf = NA + 2;           // DMA requests per wave and half-stage
};
constexpr int kGgLds = GgGeom<8>::Lds;
constexpr int kGgOutRS = 128 * 2 + 16;                // epilogue tile: 256 rows x 128 channels bf16, padded

// s_waitcnt immediate (gfx9 layout): vmcnt in bits [3:0] and [15:14], lgkmcnt in [11:8], expcnt [6:4] left at "no wait"
template <int VM, int LGKM>
constexpr int kGgWait = (VM & 15) | ((VM >> 4) << 14) | 0x70 | (LGKM << 8);

#ifdef MV_GATE_GEMM_KERNEL  // the kernel itself is compiled in gate_gemm.hip only; other units see the parameters
__device__ __forceinline__ void glds16(const void* src, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// gate_act<true> (conv_mfma.h) on two values at a time: the multiplies and adds around the three transcendentals as
// packed fp32 operations (the epilogue is VALU-bound: 64 gates per lane, two waves per SIMD)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gate_act_x2(f32x2 t, f32x2 s) {
    constexpr float kLog2e = 1.44269504088896341f;
    t[0] = __builtin_amdgcn_fmed3f(t[0], -15.f, 15.f);
    t[1] = __builtin_amdgcn_fmed3f(t[1], -15.f, 15.f);
    const f32x2 a = t * (2.f * kLog2e), c = s * -kLog2e;
    const f32x2 e2t = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
    const f32x2 es = {__builtin_amdgcn_exp2f(c[0]), __builtin_amdgcn_exp2f(c[1])};
    const f32x2 den = (e2t + 1.f) * (es + 1.f);
    const f32x2 r = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
    return (e2t - 1.f) * r;
}

#ifdef MV_GG_STAMP  // diagnostic build (tools/gg_stamps.py): shader-clock stamps of block phases, in a buffer of their own
__device__ unsigned long long gg_stamps[4096][8];  // 0-4 shader clock at phase ends, 5 / 6 the 100 MHz clock at block end / start
#ifdef MV_GG_STAMP_RES
#define MV_GG_STAMPED 4
#else
#define MV_GG_STAMPED 8
#endif
#define MV_GG_STAMP_AT(i) \
    if (MTW == MV_GG_STAMPED && threadIdx.x == 0) \
    gg_stamps[(blockIdx.y * gridDim.x + blockIdx.x) & 4095][i] = __builtin_amdgcn_s_memtime()
#else
#define MV_GG_STAMP_AT(i)
#endif

// Results leave by non-temporal stores: nothing in this launch reads them again, and lines that are not left dirty in
// L2 shorten the end of the kernel (+1.0 % at B = 1, +0.5 % at B = 8, same box, alternating).
#define MV_GG_STORE(ptr, val) __builtin_nontemporal_store(val, ptr)
template <int MTW, bool RES>
__device__ __forceinline__ void gemm_ring_body(const GateGemmParams& p) {
    using G = GgGeom<MTW>;
    constexpr int HM = G::HM, RW = G::RW, NA = G::NA, kHalfA = G::HalfA, kSlot = G::Slot, kLds = G::Lds;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    MV_GG_STAMP_AT(0);
#ifdef MV_GG_STAMP
    if (MTW == MV_GG_STAMPED && threadIdx.x == 0)
        gg_stamps[(blockIdx.y * gridDim.x + blockIdx.x) & 4095][6] = __builtin_amdgcn_s_memrealtime();
#endif
    // XCD-aware order (as conv_cl_kernel): the ny M blocks of a row tile run back to back on one XCD
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int by = slot % p.ny;
    const int bx = (slot / p.ny) * 8 + xcd;
    if (bx >= p.nx) return;
    const int b = blockIdx.y;
    const int len = seq_frames(p.lengths, b, p.Tmax) * p.spf;
    const int n0 = bx * kGgTile;
    if (n0 >= len) return;

    // residual conv: the block's tile of the old H (a first touch of what the previous launch wrote, 8 x 16 bytes per
    // thread) is requested before anything else -- the oldest entries of the wave's memory queue, they are back by the
    // time the first half-stage is, and the epilogue finds them in registers
    // (the 256-row blocks, RES with MTW = 8, have no registers to park 16 such units through the K loop: they request
    // them at the head of the epilogue instead, under the conversion of the result tile)
    constexpr int kResUnits = RES ? 2 * MTW : 1;          // 16-byte units of the [256 n][BM m] float16 tile per thread
    constexpr int kResUPR = G::BM / 8;                    // ... per row
    u32x4 hold[kResUnits];
    auto request_old_h = [&]() {
        const char* hrow0 = (const char*)p.res_h + ((long long)b * p.L * p.res_stride + by * G::BM) * 2;
#pragma unroll
        for (int k = 0; k < kResUnits; ++k) {
            const int idx = threadIdx.x + k * 512;
            const int n = n0 + idx / kResUPR, u = idx % kResUPR;
            hold[k] = u32x4{0u, 0u, 0u, 0u};
            if (n < len) hold[k] = *(const u32x4*)(hrow0 + ((long long)n * p.res_stride + u * 8) * 2);
        }
    };
    if constexpr (RES && MTW == 4) request_old_h();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int lr = lane & 15, lg = lane >> 4;

    // ---- staging.  The unit of the pipeline is a HALF-STAGE = (K tile, half s): the 64 rows of A that each wave row
    // wr multiplies in its m half s (tile rows 128 wr + 64 s + 0..63) and the 32 rows of B of every wave column's n half
    // s (tile rows 64 wc + 32 s + 0..31): 128 + 128 rows x 128 B = 32 KiB, five of them in a ring.  Wave w moves pieces
    // 2w, 2w+1 (8 rows x 128 B each) of the A part and of the B part.  Lane l of a piece fills chunk position l & 7 of
    // local row 16 w + 8 j + (l >> 3), i.e. it fetches source chunk (l & 7) ^ ((row >> 1) & 7) = (l & 7) ^ (4 j + (l >> 4)).
    const int srow = lane >> 3;
    const unsigned c16[2] = {(unsigned)(((lane & 7) ^ (lane >> 4)) * 16), (unsigned)(((lane & 7) ^ (4 + (lane >> 4))) * 16)};
    const int b_r = (wave >> 1) * 64 + (wave & 1) * 16 + srow;   // tile row of the lane's B piece j, half s: b_r + 8 j + 32 s
    // A piece j of the wave: local row 8 NA w + 8 j + (l >> 3) of the half-stage's 2 RW rows = row r of wave row wr_s
    unsigned a_voff[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int lrow = wave * 8 * NA + 8 * j + srow;
        const int wr_s = lrow / RW, r = lrow % RW;
        const unsigned chunk = (unsigned)(((lane & 7) ^ ((lrow >> 1) & 7)) * 16);
        a_voff[j] = (unsigned)(by * G::BM + wr_s * 2 * RW + r) * (unsigned)p.K * 2u + chunk;  // (half 1: + a_half)
    }
    const int t0 = n0 + b_r;
    const char* wbase = (const char*)p.w;
    const char* hb = (const char*)(p.h + (long long)b * p.L * p.h_stride);
    const char* cb = (const char*)(p.cond + (long long)b * p.L * p.ncond);
    const char* zb = (const char*)p.zero;
    const int nkc_h = p.Cp / kGgBK, nkc_c = p.ncond / kGgBK;
    const int NK = p.ktaps * nkc_h + nkc_c;
    const int pad = (p.ktaps - 1) / 2 * p.dil;

    const unsigned rs_h = (unsigned)p.h_stride * 2u, rs_c = (unsigned)p.ncond * 2u, a_half = (unsigned)(2 * RW) * (unsigned)p.K;
    const int ktaps = p.ktaps, dil = p.dil;
    // wave-uniform cursor of the next half-stage to request: segment (tap of H, or ktaps = the cond tensor), K chunk in
    // it, K tile, half.  A half-stage is requested in two steps of two pieces: its A rows, then its B rows.
    int seg = 0, kc = 0, sk = 0, sh = 0;
    auto request_a = [&](int slot_off, int j) {  // piece j of the A rows
#if !(defined(MV_GG_ABL) && (MV_GG_ABL & 8))  // timing-only: bit 3 = no A requests, bit 4 = no B requests
        char* la = smem + slot_off + wave * (NA * 1024);
        const char* wk = wbase + (unsigned)sk * 128u + (sh ? a_half : 0u);
        glds16(wk + a_voff[j], la + j * 1024);
#endif
    };
    // The activation rows of a request: scalar base (this sequence's H or cond tensor + the K chunk) + one 32-bit offset per
    // lane, like the weights -- a row outside [0, len) becomes the offset of the zero row from that base.  The four
    // offsets a lane needs (pieces j = 0, 1 of halves s = 0, 1) change only when the cursor enters another segment, so
    // a request costs no vector arithmetic.  (Formed per request, as 64-bit addresses with their multiply and selects,
    // they made an activation request cost its wave 350 cycles against 70 for a weight request -- with the SIMD's
    // other wave multiplying at the time -- and the K loop 136 k cycles against 84 k with weight requests alone.)
    unsigned b_voff[2][2];
    auto open_segment = [&]() {
        const bool is_h = seg < ktaps;
        const int shift = is_h ? seg * dil - pad : 0;
        const unsigned rs = is_h ? rs_h : rs_c;
        const unsigned zoff = (unsigned)(zb - (is_h ? hb : cb));
#pragma unroll
        for (int sx = 0; sx < 2; ++sx)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int tt = t0 + 8 * j + 32 * sx + shift;
                // (a cond row is read for tt = its own time step; outside [0, len) it multiplies nothing that is kept)
                b_voff[sx][j] = ((unsigned)tt < (unsigned)len ? (unsigned)tt * rs : zoff) + c16[j];
            }
    };
    open_segment();
    auto request_b = [&](int slot_off, int j, int sx) {  // piece j of the B rows of half sx (= the cursor's); j = 1 steps the cursor
        char* lb = smem + slot_off + kHalfA + wave * 2048;
        const bool is_h = seg < ktaps;
        const char* base = (is_h ? hb : cb) + (unsigned)kc * 128u;
#if !(defined(MV_GG_ABL) && (MV_GG_ABL & 16))
        glds16(base + b_voff[sx][j], lb + j * 1024);
#endif
        if (j == 1) {
            const bool tile_done = sh == 1;
            const bool wrap = tile_done && kc + 1 == (is_h ? nkc_h : nkc_c);
            sh = __builtin_amdgcn_readfirstlane(sh ^ 1);
            sk = __builtin_amdgcn_readfirstlane(tile_done ? sk + 1 : sk);
            kc = __builtin_amdgcn_readfirstlane(wrap ? 0 : (tile_done ? kc + 1 : kc));
            seg = __builtin_amdgcn_readfirstlane(wrap ? seg + 1 : seg);
            if (wrap) open_segment();
        }
    };
    auto next_slot = [](int off) { return off + kSlot == kLds ? 0 : off + kSlot; };
    auto prev_slot = [](int off) { return off == 0 ? kLds - kSlot : off - kSlot; };

    // ---- fragment addresses within a half-stage slot
    const int swz0 = (lg ^ (lr >> 1)) * 16;  // chunk lg (k = 0..31) of a row; chunk 4 + lg (k = 32..63) is swz0 ^ 64
    const char* fa = smem + (wr * RW + lr) * 128;
    const char* fb = smem + kHalfA + (wc * 32 + lr) * 128;

    f32x4 acc[MTW][4];
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 A0[HM][2], A1[HM][2], B0[2][2], B1[2][2];

#if defined(MV_GG_ABL) && (MV_GG_ABL & 2)  // timing-only: bit 1 = no fragment reads in the K loop
#define MV_GG_RD(dst, src) asm volatile("" : "+v"(dst) : "v"(src))
#else
#define MV_GG_RD(dst, src) dst = *(const u32x4*)(src)
#endif
    auto read_a = [&](u32x4 (&A)[HM][2], int slot_off) {
        const char* base = fa + slot_off;
#pragma unroll
        for (int i = 0; i < HM; ++i) {
            MV_GG_RD(A[i][0], base + i * 2048 + swz0);
            MV_GG_RD(A[i][1], base + i * 2048 + (swz0 ^ 64));
        }
    };
    auto read_b = [&](u32x4 (&Bf)[2][2], int slot_off) {
        const char* base = fb + slot_off;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            MV_GG_RD(Bf[i][0], base + i * 2048 + swz0);
            MV_GG_RD(Bf[i][1], base + i * 2048 + (swz0 ^ 64));
        }
    };
    // the barrier that ends a load part: fragment reads issued, then everybody; the reads are waited for behind it
    auto bar_then_wait_reads = [&]() {
        __builtin_amdgcn_sched_barrier(0);
#if !(defined(MV_GG_ABL) && (MV_GG_ABL & 4))  // timing-only: bit 2 = no barriers in the K loop
        __builtin_amdgcn_s_barrier();
#endif
        __builtin_amdgcn_s_waitcnt(kGgWait<63, 0>);
        __builtin_amdgcn_sched_barrier(0);
    };
    // One multiply part: 16 MFMAs and the barrier that ends it.  (The DMA requests stay in the load parts: issued between
    // these MFMAs, with the SIMD's other wave in its load part, each request stalled the pipe ~185 cycles: 86 us against 68.)
    auto quad = [&](const u32x4 (&A)[HM][2], const u32x4 (&Bf)[2][2], int mh, int nh) {
        __builtin_amdgcn_s_setprio(kMfmaPrio);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < HM; ++i) mma16<true>(acc[mh * HM + i][nh * 2 + j], A[i][ks], Bf[j][ks]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
#if !(defined(MV_GG_ABL) && (MV_GG_ABL & 4))
        __builtin_amdgcn_s_barrier();
#endif
        __builtin_amdgcn_sched_barrier(0);
    };
    // req 1: the A pieces of the half-stage under the cursor, 2: its B pieces; sx = which half that is (the loop requests
    // first halves in phases 0 / 1 and second halves in phases 2 / 3)
    auto request = [&](int req, int slot_off, int sx) {
#if !(defined(MV_GG_ABL) && (MV_GG_ABL & 1))  // timing-only: bit 0 = no DMA in the K loop
        if (req == 1) {
#pragma unroll
            for (int j = 0; j < NA; ++j) request_a(slot_off, j);
        } else if (req == 2) {
            request_b(slot_off, 0, sx);
            request_b(slot_off, 1, sx);
        }
#endif
    };
    // This wave's pieces of the half-stage that a LATER load part reads have landed: in steady state that half-stage is
    // followed by two younger ones in the wave's request queue (2 (NA + 2) pieces); once the cursor has run off the end of K,
    // fewer are, and the wave waits for everything.  The wait sits one load part before the first read, so that between
    // the two lies a barrier both wave groups have passed (they run one barrier apart).
    auto wait_dma = [&](bool issued_here) {  // issued_here: this load part's NA weight requests are already in the queue
#if defined(MV_GG_ABL) && (MV_GG_ABL & 128)  // timing-only: bit 7 = requests are never waited for
        return;
#endif
        if (sk >= NK)
            __builtin_amdgcn_s_waitcnt(kGgWait<0, 15>);
        else if (issued_here)
            __builtin_amdgcn_s_waitcnt(kGgWait<2 * G::PerHalf + NA, 15>);
        else
            __builtin_amdgcn_s_waitcnt(kGgWait<2 * G::PerHalf, 15>);
    };

    // prologue: fill the ring (H >= 6: the host checks K >= 3 K tiles), half-stage 0 landed
#pragma unroll
    for (int i = 0; i < kGgSlots; ++i) {
#pragma unroll
        for (int j = 0; j < NA; ++j) request_a(i * kSlot, j);
        request_b(i * kSlot, 0, i & 1);
        request_b(i * kSlot, 1, i & 1);
    }
    __builtin_amdgcn_s_waitcnt(kGgWait<4 * G::PerHalf, 15>);  // all but the four youngest half-stages
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    read_b(B0, 0);
    MV_GG_STAMP_AT(1);
    // Two wave groups, one barrier apart: while the waves of group 0 (wr = 0, one per SIMD) multiply, those of group 1
    // issue their fragment reads and DMA requests, and vice versa -- every part of the loop is closed by a barrier.
    if (wr == 1) __builtin_amdgcn_s_barrier();
    int s0 = 0;
#if defined(MV_GG_ABL) && (MV_GG_ABL & 64)  // timing-only: bit 6 = no K loop (prologue + epilogue only)
    for (int t = 0; t < 0; ++t) {
#else
    for (int t = 0; t < NK; ++t) {
#endif
        const int s1 = next_slot(s0), s2 = next_slot(s1), sp = prev_slot(s0);
        // (in every load part the DMA requests go out BEFORE the fragment reads: their issue then overlaps the reads'
        // -- 67.8 -> 66.1 us per launch against reads first)
        // phase 0: A0 of (t, 0); make sure (t, 1) has landed; first half of the request for the slot before this one
        const int req0 = t > 0 && sk < NK ? 1 : 0;
        request(req0, sp, 0);
        read_a(A0, s0);
        wait_dma(req0 != 0);
        bar_then_wait_reads();
        quad(A0, B0, 0, 0);
        // phase 1: B1 of (t, 1)
        request(2 * req0, sp, 0);
        read_b(B1, s1);
        bar_then_wait_reads();
        quad(A0, B1, 0, 1);
        // phase 2: A1 of (t, 1); make sure (t+1, 0) has landed; (t, 0)'s slot is free: first half of its refill
        const int req2 = sk < NK ? 1 : 0;
        request(req2, s0, 1);
        read_a(A1, s1);
        wait_dma(req2 != 0);
        bar_then_wait_reads();
        quad(A1, B0, 1, 0);
        // phase 3: B0 of (t+1, 0) (past the last K tile: a harmless read of a slot nobody writes any more)
        request(2 * req2, s0, 1);
        read_b(B0, s2);
        bar_then_wait_reads();
        quad(A1, B1, 1, 1);
        s0 = s2;
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();  // (group 1 ran one barrier behind)
    __builtin_amdgcn_s_waitcnt(kGgWait<0, 0>);
    __builtin_amdgcn_s_barrier();
    MV_GG_STAMP_AT(2);

    if constexpr (!RES) {
    // ---- epilogue: bias, gate, bf16; the [256 n][128 c] tile leaves through LDS in 16-byte units along c
    // The biases first: they are needed at once, and with them ahead in the queue the loads behind them (the fold
    // matrix and the running partial sums, a first touch of what the previous layer's launch wrote) stay in flight
    // under the gate arithmetic.
    const float* bias = p.bias + by * kGgTile + wr * 128;
    f32x4 bts[4][2];
#pragma unroll
    for (int tp = 0; tp < 4; ++tp) {
        bts[tp][0] = *(const f32x4*)(bias + (2 * tp) * 16 + lg * 4);
        bts[tp][1] = *(const f32x4*)(bias + (2 * tp + 1) * 16 + lg * 4);
    }
    // (fold: the A fragments of the 16 x 128 slice of the fold matrix, hi and lo, requested now, used after the tile)
    u32x4 fw[2][4];
    if (p.fold_w) {
#pragma unroll
        for (int hl = 0; hl < 2; ++hl)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                fw[hl][ks] = *(const u32x4*)(p.fold_w + ((long long)(hl * 16 + lr) * p.Cp + by * 128 + ks * 32 + lg * 8));
    }
    // ... and the partial sums this block adds to (layers after a flow's first), so that their latency hides under the
    // gate arithmetic instead of ending the kernel
    f32x4 fold_old[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    float* fold_dst[2] = {nullptr, nullptr};
    if (p.fold_w) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int n = n0 + (wave * 2 + q) * 16 + lr;
            if (lg < 2 && lg * 4 < p.fold_nout && n < len) {
                fold_dst[q] = p.fold_out + by * p.fold_part_stride + ((long long)b * p.L + n) * 8 + lg * 4;
                if (p.fold_accumulate) fold_old[q] = *(const f32x4*)fold_dst[q];
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int tp = 0; tp < 4; ++tp) {
        const f32x4 bt = bts[tp][0], bs = bts[tp][1];
        const int cl = wr * 64 + tp * 16 + lg * 4;
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) {
            const int nl = wc * 64 + tn * 16 + lr;
            const f32x4 tv = acc[2 * tp][tn] + bt;
            const f32x4 sv = acc[2 * tp + 1][tn] + bs;
            const f32x2 g01 = gate_act_x2(f32x2{tv[0], tv[1]}, f32x2{sv[0], sv[1]});
            const f32x2 g23 = gate_act_x2(f32x2{tv[2], tv[3]}, f32x2{sv[2], sv[3]});
            const f32x4 g = {g01[0], g01[1], g23[0], g23[1]};
            *(unsigned long long*)(smem + nl * kGgOutRS + cl * 2) =
                ((unsigned long long)pack_bf16(g[2], g[3]) << 32) | pack_bf16(g[0], g[1]);
        }
    }
    __syncthreads();
    MV_GG_STAMP_AT(3);
    if (p.fold_w) {
        // out8[n][m] (+)= sum_c fold[m][c] acts[n][c] over the block's 128 channels: 16 n-tiles, two per wave, as
        // v_mfma_f32_16x16x32_bf16 with the fold matrix as A (rows >= fold_nout are zero) and the LDS tile as B
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int nl = (wave * 2 + q) * 16 + lr;
            f32x4 fa = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const u32x4 bq = *(const u32x4*)(smem + nl * kGgOutRS + (ks * 32 + lg * 8) * 2);
                mma16<true>(fa, fw[0][ks], bq);
                mma16<true>(fa, fw[1][ks], bq);
            }
            // lane: column n = nl, rows m = 4 lg .. 4 lg + 3
            if (fold_dst[q]) *(f32x4*)fold_dst[q] = fa + fold_old[q];
        }
    }
    unsigned short* ob = p.out + (long long)b * p.L * p.Cp + by * 128;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int idx = threadIdx.x + k * 512;
        const int nl = idx >> 4, u = idx & 15;
        const int n = n0 + nl;
        if (n < len) MV_GG_STORE((u32x4*)(ob + (long long)n * p.Cp + u * 8), *(const u32x4*)(smem + nl * kGgOutRS + u * 16));
    }
    } else {
    // ---- epilogue of the residual conv: H[n][m] = f16(H[n][m] + f16(acc + bias)), and its bfloat16 copy.  The old H
    // rows were requested when the block started (`hold`; by the 256-row blocks just now), the [256 n][BM m] result
    // tile goes through LDS as float16 and is combined in 16-byte units along m.
    constexpr int kRS = G::BM * 2 + 16;  // padded row of the tile (bank-conflict-free 8-byte writes, 16-byte reads)
    if constexpr (MTW == 8) request_old_h();
    char* hrow = (char*)p.res_h + ((long long)b * p.L * p.res_stride + by * G::BM) * 2;
    char* hbrow = (char*)p.res_hb + ((long long)b * p.L * p.res_stride + by * G::BM) * 2;
    const float* bias = p.bias + by * G::BM + wr * (16 * MTW);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
        const f32x4 bv = *(const f32x4*)(bias + mt * 16 + lg * 4);
        const int ml = wr * (16 * MTW) + mt * 16 + lg * 4;
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) {
            const int nl = wc * 64 + tn * 16 + lr;
            const f32x4 v = acc[mt][tn] + bv;
            f16x4 hv;
#pragma unroll
            for (int i = 0; i < 4; ++i) hv[i] = (_Float16)fminf(fmaxf(v[i], -65504.f), 65504.f);
            *(f16x4*)(smem + nl * kRS + ml * 2) = hv;
        }
    }
    __syncthreads();
    MV_GG_STAMP_AT(3);
#pragma unroll
    for (int k = 0; k < kResUnits; ++k) {
        const int idx = threadIdx.x + k * 512;
        const int nl = idx / kResUPR, u = idx % kResUPR;
        const int n = n0 + nl;
        if (n >= len) continue;
        const f16x8 r = __builtin_bit_cast(f16x8, *(const u32x4*)(smem + nl * kRS + u * 16));
        const f16x8 o = __builtin_bit_cast(f16x8, hold[k]);
        float y[8];
        f16x8 hh;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            y[i] = (float)r[i] + (float)o[i];
            hh[i] = (_Float16)fminf(fmaxf(y[i], -65504.f), 65504.f);
        }
        const long long e = ((long long)n * p.res_stride + u * 8) * 2;
        MV_GG_STORE((f16x8*)(hrow + e), hh);
        MV_GG_STORE((u32x4*)(hbrow + e), (u32x4{pack_bf16(y[0], y[1]), pack_bf16(y[2], y[3]), pack_bf16(y[4], y[5]), pack_bf16(y[6], y[7])}));
    }
    }
    MV_GG_STAMP_AT(4);
#ifdef MV_GG_STAMP
    if (MTW == MV_GG_STAMPED && threadIdx.x == 0)
        gg_stamps[(blockIdx.y * gridDim.x + blockIdx.x) & 4095][5] = __builtin_amdgcn_s_memrealtime();
#endif
}

__global__ __launch_bounds__(512) void gate_gemm_kernel(const GateGemmParams p) { gemm_ring_body<8, false>(p); }
// The residual 1x1 conv on the same ring: 128 x 256 tiles (4 M blocks x the row tiles: as many blocks as the gated conv)
__global__ __launch_bounds__(512) void res_gemm_kernel(const GateGemmParams p) { gemm_ring_body<4, true>(p); }
// ... and on 256 x 256 tiles, for calls with enough row tiles to fill the chip with those (a third fewer operand bytes
// per multiply-add: the residual conv is bound by what a CU can pull in)
__global__ __launch_bounds__(512) void res_gemm256_kernel(const GateGemmParams p) { gemm_ring_body<8, true>(p); }

#endif  // MV_GATE_GEMM_KERNEL

}  // namespace mv
