// fused_rb.h -- fused ResBlock1 kernel (bf16 operands), hifigan/models.py:96-103 of the reference.
//
// One block owns a WINDOW of W consecutive time rows of one sequence (all C channels) and runs
// NPAIR consecutive (dilated conv, conv) pairs on it without leaving the CU:
//
//     A16 = bf16(lrelu(x))                         LDS, [W + 2*guardA][C]
//     for each pair m:   Y16 = bf16(lrelu(conv1_m(A16) + b1))      LDS, [W + 2*guardY][C]
//                        x   = x + conv2_m(Y16) + b2               fp32
//                        A16 = bf16(lrelu(x))
//     out = MRF-combine(x)   for the rows the halo has not eaten
//
// Every conv is computed over the whole window; rows closer than H = sum of the pairs' halos to a
// window edge are wrong after the last pair and are simply not written, so a block emits
// NT = W - 2H rows (blocks overlap by 2H input rows).  Rows outside [0, len) of the sequence are
// forced to zero in A16 / Y16, which is exactly the zero padding each conv of the reference's
// batch-1 call sees.  HBM traffic per ResBlock drops from 15 activation passes (6 convs unfused)
// to ~2.3 with NPAIR = 3, ~7 with NPAIR = 1.
//
// The fp32 residual stream x lives in registers in MFMA accumulator layout (each wave keeps x for exactly the
// (channel, row) tiles it accumulates).
//
// Waves: WM x WN; wave (wm, wn) owns M tiles [wm*TM, +TM) (TM = C/16/WM) and N tiles [wn*TN, +TN).
// Weights stream L2 -> registers in A-fragment order exactly as in conv_mfma.h.
#pragma once
#include "conv_mfma.h"

namespace mv {

// MV_TRACE (diagnostic builds only, tools/trace.sh): thread 0 of every block accumulates shader-clock time
// per phase category and appends one record to a global table that tools/trace_report.py aggregates.
enum { TR_SETUP = 0, TR_LOADX, TR_BARRIER, TR_KLOOP, TR_EPI, TR_STORE, TR_POST, TR_NCAT };
#ifdef MV_TRACE
constexpr int kTraceSlots = 16;
constexpr int kTraceCap = 1 << 19;
__device__ unsigned long long g_trace[(size_t)kTraceCap * kTraceSlots];
__device__ unsigned int g_trace_n;
struct Tracer {
    unsigned long long* t;  // TR_NCAT accumulators in LDS, touched by thread 0 only
    unsigned long long last, first;
    __device__ __forceinline__ void begin() {
        if (threadIdx.x == 0) {
            for (int i = 0; i < TR_NCAT; ++i) t[i] = 0;
            first = last = clock64();
        }
    }
    __device__ __forceinline__ void lap(int cat) {
        if (threadIdx.x == 0) {
            const unsigned long long now = clock64();
            t[cat] += now - last;
            last = now;
        }
    }
    __device__ __forceinline__ void flush(int tag) {
        if (threadIdx.x != 0) return;
        const unsigned slot = atomicAdd(&g_trace_n, 1u);
        if (slot >= (unsigned)kTraceCap) return;
        unsigned long long* r = g_trace + (size_t)slot * kTraceSlots;
        r[0] = (unsigned long long)tag;
        r[1] = wall_clock64();
        r[2] = first;
        r[3] = last;
        for (int i = 0; i < TR_NCAT; ++i) r[4 + i] = t[i];
    }
};
#define MV_TRACER(tr)                                    \
    __shared__ unsigned long long tr##_lds[TR_NCAT];     \
    Tracer tr;                                           \
    tr.t = tr##_lds;                                     \
    tr.begin()
#define MV_LAP(tr, cat) (tr).lap(cat)
#define MV_FLUSH(tr, tag) (tr).flush(tag)
#else
#define MV_TRACER(tr) ((void)0)
#define MV_LAP(tr, cat) ((void)0)
#define MV_FLUSH(tr, tag) ((void)0)
#endif

struct FusedRbParams {
    const void* x;    // [B][rows][C] float32, or float16 when io16
    void* out;        // same layout and type
    int io16;
    long long batch_stride;
    const int32_t* lengths;
    int B;
    int Tmax;
    int rows_per_frame;
    int K;        // kernel size of every conv in this ResBlock
    int npair;    // pairs run by this launch (1 or 3)
    int dil[3];
    const void* w1[3];
    const void* w2[3];
    const float* b1[3];
    const float* b2[3];
    int mode;     // ACC_* for the final store (MRF accumulation)
    float div;
    int NT;       // rows written per block = W - 2*H
    int H;        // sum over pairs of (K-1)/2 * (dil+1)
    int guardA;   // max over pairs of dil*(K-1)/2
    int guardY;   // (K-1)/2
    int g1;       // apron rows (see issue_apron16): dil[0]*(K-1)/2 on the float16 path, else 0
};

// acc += conv over the window (the caller initialises acc)
template <int TM, int TN, int DEPTH = 4, bool X3 = false, int OPF = OP_BF16>
__device__ __forceinline__ void conv_window(f32x4 (&acc)[TM][TN], const char* lds_b, int RS, int ktaps, int dil,
                                            int nkc, const u32x4* wlane, long long mtile_stride,
                                            const APre<TM, DEPTH>* pre = nullptr) {
    mma_pass<true, TM, TN, DEPTH, X3, OPF>(acc, lds_b, RS, ktaps, dil, nkc, wlane, mtile_stride, pre);
}

// Biases of every conv of a launch live in LDS behind the two tiles ([conv][C] floats, filled once per persistent
// block): a conv's accumulators START from its bias (conv2's: from residual + bias, so the K loop accumulates
// straight into the residual registers) and the epilogues are left with lrelu + pack only.
template <int TM>
__device__ __forceinline__ void lds_bias(f32x4 (&bias)[TM], const float* bias_s, int conv, int C, int chl) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) bias[tm] = *(const f32x4*)(bias_s + conv * C + chl + 4 * tm);
}

// leaky_relu on 4 values as max(x, slope*x) (0 < slope <= 1): two packed multiplies + four v_max
__device__ __forceinline__ f32x4 lrelu4(const f32x4& v, float slope) {
    const f32x4 s = v * slope;
    return f32x4{fmaxf(v[0], s[0]), fmaxf(v[1], s[1]), fmaxf(v[2], s[2]), fmaxf(v[3], s[3])};
}

// (the pack*_bf16 helpers below produce MFMA operands of format OPF: bfloat16, or saturating float16)
template <int OPF = OP_BF16>
__device__ __forceinline__ unsigned long long pack4_bf16(const f32x4& v) {
    return ((unsigned long long)pack_op<OPF>(v[2], v[3]) << 32) | pack_op<OPF>(v[0], v[1]);
}

template <int OPF = OP_BF16>
__device__ __forceinline__ unsigned long long pack4_bf16_lrelu(const f32x4& v, float slope, bool keep) {
    return keep ? pack4_bf16<OPF>(lrelu4(v, slope)) : 0ull;
}

// 8 consecutive channels of one row = the lane's fragments of its two M tiles (pair-interleaved channel order,
// see fused_rb_kernel): one 16-byte LDS unit
template <int OPF = OP_BF16>
__device__ __forceinline__ u32x4 pack8_bf16(const f32x4& a, const f32x4& b) {
    return u32x4{pack_op<OPF>(a[0], a[1]), pack_op<OPF>(a[2], a[3]), pack_op<OPF>(b[0], b[1]), pack_op<OPF>(b[2], b[3])};
}
// leaky_relu + conversion to operands in one go.  float16 operands must not overflow to inf (inf * 0 = NaN in the
// next MFMA); the clamp is folded into the activation instead of costing two more instructions per value:
//   upper side: max(x, s x) clamped at 65504 is the MEDIAN of (x, s x, 65504) for 0 < s < 1 -- one v_med3 in place
//               of the v_max;
//   lower side: one packed v_pk_max_f16 with -65504 on the converted pair (-inf -> -65504);
// INRANGE: the values are float16 values already (a window as it arrived from memory): no clamp at all.
template <int OPF, bool INRANGE>
__device__ __forceinline__ unsigned lrelu_pack2(float x0, float x1, float s0, float s1) {
    if constexpr (OPF == OP_F16) {
        typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
        if constexpr (INRANGE) {
            const f16x2 v = {(_Float16)fmaxf(x0, s0), (_Float16)fmaxf(x1, s1)};
            return __builtin_bit_cast(unsigned, v);
        } else {
            f16x2 v = {(_Float16)__builtin_amdgcn_fmed3f(x0, s0, 65504.f), (_Float16)__builtin_amdgcn_fmed3f(x1, s1, 65504.f)};
            const f16x2 lo = {(_Float16)-65504.f, (_Float16)-65504.f};
            v = __builtin_elementwise_max(v, lo);
            return __builtin_bit_cast(unsigned, v);
        }
    } else {
        return pack_bf16(fmaxf(x0, s0), fmaxf(x1, s1));
    }
}
template <int OPF = OP_BF16, bool INRANGE = false>
__device__ __forceinline__ u32x4 pack8_lrelu(const f32x4& a, const f32x4& b, float slope) {
    const f32x4 sa = a * slope, sb = b * slope;
    return u32x4{lrelu_pack2<OPF, INRANGE>(a[0], a[1], sa[0], sa[1]), lrelu_pack2<OPF, INRANGE>(a[2], a[3], sa[2], sa[3]),
                 lrelu_pack2<OPF, INRANGE>(b[0], b[1], sb[0], sb[1]), lrelu_pack2<OPF, INRANGE>(b[2], b[3], sb[2], sb[3])};
}
template <int OPF = OP_BF16, bool INRANGE = false>
__device__ __forceinline__ u32x4 pack8_bf16_lrelu(const f32x4& a, const f32x4& b, float slope, bool keep) {
    return keep ? pack8_lrelu<OPF, INRANGE>(a, b, slope) : u32x4{0u, 0u, 0u, 0u};
}
__device__ __forceinline__ u32x4 pack8_f16_sat(const f32x4& a, const f32x4& b) {
    f16x8 h;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h[i] = (_Float16)fminf(fmaxf(a[i], -65504.f), 65504.f);
        h[4 + i] = (_Float16)fminf(fmaxf(b[i], -65504.f), 65504.f);
    }
    return __builtin_bit_cast(u32x4, h);
}
__device__ __forceinline__ void unpack8_f16(const u32x4& q, f32x4& a, f32x4& b) {
    const f16x8 h = __builtin_bit_cast(f16x8, q);
    a = f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    b = f32x4{(float)h[4], (float)h[5], (float)h[6], (float)h[7]};
}

__device__ __forceinline__ unsigned long long pack4_f16_sat(const f32x4& v) {
    f16x4 h;
#pragma unroll
    for (int i = 0; i < 4; ++i) h[i] = (_Float16)fminf(fmaxf(v[i], -65504.f), 65504.f);
    return __builtin_bit_cast(unsigned long long, h);
}

__device__ __forceinline__ f32x4 unpack4_f16(unsigned long long q) {
    const f16x4 h = __builtin_bit_cast(f16x4, q);
    return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
}

// ---- split operands (MV_PREC_BF16X3, conv_mfma.h split_bf16x2): an operand row holds its C channels twice,
// [hi: C bf16][lo: C bf16]; the fragment position of a lane is the same in both halves, C*2 bytes apart.
template <int C>
__device__ __forceinline__ void store8_x3(char* dst, const f32x4& a, const f32x4& b, bool keep) {
    u32x4 hi = {0u, 0u, 0u, 0u}, lo = {0u, 0u, 0u, 0u};
    if (keep) split_bf16x2(a, b, hi, lo);
    *(u32x4*)dst = hi;
    *(u32x4*)(dst + C * 2) = lo;
}
template <int C>
__device__ __forceinline__ void store4_x3(char* dst, const f32x4& a, bool keep) {
    u32x4 hi = {0u, 0u, 0u, 0u}, lo = {0u, 0u, 0u, 0u};
    if (keep) split_bf16x2(a, a, hi, lo);
    *(unsigned long long*)dst = ((unsigned long long)hi[1] << 32) | hi[0];
    *(unsigned long long*)(dst + C * 2) = ((unsigned long long)lo[1] << 32) | lo[0];
}

// threadIdx.x behind an opaque move: index arithmetic derived from it is recomputed where it is used
// instead of being hoisted out of the persistent window loop (where it would sit in ~40 VGPRs through
// every K loop)
__device__ __forceinline__ int tid_here() {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}

// Coalesced window load (float16 activations): rows [t0, t0+W) x C channels of one sequence, 16 bytes per
// lane along the channel axis.  issue_window16 puts every load of the thread in flight (into registers) and
// commit_raw16 drops them into an LDS tile as raw float16, zeros for rows outside [0, len).  The two halves are
// separate so that a persistent block can have the NEXT window's rows in flight while it still computes the
// current one.  The loads are unconditional (row index clamped into the sequence, zeroing deferred to the
// commit): predicated loads put a phi on every register and the compiler then waits for all of them (vmcnt(0))
// in the middle of the epilogue that is supposed to hide their latency.
template <int C, int W, int NTHREADS>
struct WinLoad {
    static constexpr int UPR = C / 8;
    static constexpr int TOTAL = W * UPR;
    static constexpr int PER = (TOTAL + NTHREADS - 1) / NTHREADS;
};

template <int C, int W, int NTHREADS>
__device__ __forceinline__ void issue_window16(const _Float16* xb, int t0, int len,
                                               u32x4 (&raw)[WinLoad<C, W, NTHREADS>::PER]) {
    using WL = WinLoad<C, W, NTHREADS>;
    const int tid = tid_here();
#pragma unroll
    for (int k = 0; k < WL::PER; ++k) {
        int idx = tid + k * NTHREADS;
        if (WL::TOTAL % NTHREADS != 0) idx = idx < WL::TOTAL ? idx : WL::TOTAL - 1;
        const int r = idx / WL::UPR, u = idx - r * WL::UPR;
        int g = t0 + r;
        g = g < 0 ? 0 : g;
        g = g < len ? g : len - 1;  // len >= 1: only windows of non-empty sequences exist
        raw[k] = *(const u32x4*)(xb + (long long)g * C + u * 8);
    }
}

// the window in flight -> tile T0 as raw float16 (converted in place afterwards, fragment by fragment)
template <int C, int W, int NTHREADS>
__device__ __forceinline__ void commit_raw16(const u32x4 (&raw)[WinLoad<C, W, NTHREADS>::PER], char* T0, int RS,
                                             int t0, int len) {
    using WL = WinLoad<C, W, NTHREADS>;
    const int tid = tid_here();
#pragma unroll
    for (int k = 0; k < WL::PER; ++k) {
        const int idx = tid + k * NTHREADS;
        if (WL::TOTAL % NTHREADS != 0 && idx >= WL::TOTAL) continue;
        const int r = idx / WL::UPR, u = idx - r * WL::UPR;
        const int g = t0 + r;
        const bool ok = g >= 0 && g < len;
        u32x4 v = raw[k];
        if (!ok) v = u32x4{0u, 0u, 0u, 0u};
        *(u32x4*)(T0 + r * RS + u * 16) = v;
    }
}

// Apron: the g1 = dil*(K-1)/2 rows either side of the window that only conv1 of a ResBlock's FIRST pair reads.
// With real rows there (instead of the zeros that stand for "outside the window") that conv's output is valid
// on all W rows, so the first pair's dilation costs no halo: a lone pair loses (K-1)/2 rows per side instead
// of (K-1)/2*(dil+1).  The rows land in the A tile's guard rows [-g1, 0) and [W, W+g1) as bf16(lrelu(x)), ready
// to be an MFMA operand (no lane keeps them as residual); later pairs read whatever is there as before (finite
// values that only reach outputs inside the halo).
constexpr int kMaxApron = 25;  // dilation 5, K = 11
template <int C, int NTHREADS>
struct ApronLoad {
    static constexpr int UPR = C / 8;
    static constexpr int PER = (2 * kMaxApron * UPR + NTHREADS - 1) / NTHREADS;
};

template <int C, int W, int NTHREADS>
__device__ __forceinline__ void issue_apron16(const _Float16* xb, int t0, int len, int g1,
                                              u32x4 (&raw)[ApronLoad<C, NTHREADS>::PER]) {
    using AL = ApronLoad<C, NTHREADS>;
    const int tid = tid_here();
    const int total = 2 * g1 * AL::UPR;
#pragma unroll
    for (int k = 0; k < AL::PER; ++k) {
        int idx = tid + k * NTHREADS;
        idx = idx < total ? idx : 0;
        const int r = idx / AL::UPR, u = idx - r * AL::UPR;
        int g = t0 + (r < g1 ? r - g1 : W + r - g1);
        g = g < 0 ? 0 : g;
        g = g < len ? g : len - 1;
        raw[k] = *(const u32x4*)(xb + (long long)g * C + u * 8);
    }
}

template <int C, int W, int NTHREADS, int OPF = OP_BF16>
__device__ __forceinline__ void commit_apron16(const u32x4 (&raw)[ApronLoad<C, NTHREADS>::PER], char* T0, int RS,
                                               int t0, int len, int g1) {
    using AL = ApronLoad<C, NTHREADS>;
    const int tid = tid_here();
    const int total = 2 * g1 * AL::UPR;
#pragma unroll
    for (int k = 0; k < AL::PER; ++k) {
        const int idx = tid + k * NTHREADS;
        if (idx >= total) continue;
        const int r = idx / AL::UPR, u = idx - r * AL::UPR;
        const int row = r < g1 ? r - g1 : W + r - g1;
        const int g = t0 + row;
        f32x4 a, b;
        unpack8_f16(raw[k], a, b);
        *(u32x4*)(T0 + row * RS + u * 16) = pack8_bf16_lrelu<OPF, true>(a, b, 0.1f, g >= 0 && g < len);
    }
}

// Persistent blocks walk the list of (sequence, window) pairs that actually exist: cum[b] = number of
// windows of the sequences before b (window = NT output rows), built once per block by wave 0.
constexpr int kMaxSeq = 1024;
#ifndef MV_XCD_WINDOWS
#define MV_XCD_WINDOWS 1
#endif

__device__ __forceinline__ int window_table(int* cum, const int32_t* lengths, int B, int Tmax, int rpf, int NT) {
    if (threadIdx.x < 64) {
        int carry = 0;
        for (int base = 0; base < B; base += 64) {
            const int i = base + (int)threadIdx.x;
            int v = 0;
            if (i < B) {
                const int len = seq_frames(lengths, i, Tmax) * rpf;
                v = (len + NT - 1) / NT;
            }
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(v, o);
                if ((int)threadIdx.x >= o) v += t;
            }
            if (i < B) cum[i + 1] = carry + v;
            carry += __shfl(v, 63);
        }
        if (threadIdx.x == 0) cum[0] = 0;
    }
    __syncthreads();
    return cum[B];
}

// Which windows a persistent block walks: v0, v0 + step, ... < vend.  Workgroups go to the 8 XCDs round-robin by
// index and every XCD has its own L2, so XCD x takes the x-th eighth of the window list and its blocks sweep it
// side by side: neighbouring windows (which share their apron / halo rows) then meet in one L2 instead of two.
struct WindowWalk {
    int v0, step, vend;
    __device__ __forceinline__ WindowWalk(int V) {
        if ((gridDim.x & 7) == 0 && MV_XCD_WINDOWS) {
            const int per = (V + 7) >> 3, x = blockIdx.x & 7;
            v0 = x * per + (int)(blockIdx.x >> 3);
            step = (int)(gridDim.x >> 3);
            vend = (x + 1) * per < V ? (x + 1) * per : V;
        } else {
            v0 = blockIdx.x;
            step = gridDim.x;
            vend = V;
        }
    }
};

// largest b with cum[b] <= v  (v < cum[B])
__device__ __forceinline__ int find_seq(const int* cum, int B, int v) {
    int lo = 0, hi = B;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (cum[mid] <= v)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

// Coalesced window store (float16 activations): rows [H, H+NT) of the float16 tile at T0 (row stride RS)
// go to rows t0+H.. of the output, 16 bytes per lane, with the MRF accumulation of models.py:154-160.
template <int C, int NTHREADS>
__device__ __forceinline__ void store_window16(_Float16* ob, const char* T0, int RS, int t0, int H, int NT, int len,
                                               int mode, float div) {
    const int tid = tid_here();
    constexpr int UPR = C / 8;
    constexpr int U = 4;
    int rows = len - (t0 + H);
    rows = rows < NT ? rows : NT;
    const int total = rows * UPR;
    for (int base = tid; base < total; base += NTHREADS * U) {
        u32x4 old[U];
        if (mode != ACC_STORE) {
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const int idx = base + k * NTHREADS;
                const int r = idx / UPR, u = idx - r * UPR;
                if (idx < total) old[k] = *(const u32x4*)(ob + (long long)(t0 + H + r) * C + u * 8);
            }
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int idx = base + k * NTHREADS;
            if (idx >= total) continue;
            const int r = idx / UPR, u = idx - r * UPR;
            u32x4 v = *(const u32x4*)(T0 + (H + r) * RS + u * 16);
            if (mode != ACC_STORE) {
                const f16x8 a = __builtin_bit_cast(f16x8, v), o = __builtin_bit_cast(f16x8, old[k]);
                f16x8 y;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float t = (float)a[i] + (float)o[i];
                    if (mode == ACC_ADD_DIV) t = t / div;
                    y[i] = (_Float16)fminf(fmaxf(t, -65504.f), 65504.f);
                }
                v = __builtin_bit_cast(u32x4, y);
            }
            *(u32x4*)(ob + (long long)(t0 + H + r) * C + u * 8) = v;
        }
    }
}

// IO16X: float16 window IO through LDS with prefetch (p.io16 set) -- compile-time for the same
// reason as in fused_stage_kernel (one variant per function keeps the register allocator from spilling).
// ONE: a single (conv, conv) pair per launch, known at compile time (most launches: the pair loop and the
// hand-over to a next pair drop out).
// X3: split bf16 operands (MV_PREC_BF16X3), float32 window IO; the tiles hold (hi, lo) pairs, every conv runs three
// K steps per k-chunk (mma_pass), and the first pair's apron rows are loaded as float32.
// OPF: operand format of the tiles and weights (OP_BF16 / OP_F16, conv_mfma.h)
template <int C, int WM, int WN, int TN, bool IO16X = false, bool ONE = false, bool X3 = false, int OPF = OP_BF16>
__global__ __launch_bounds__(WM* WN * 64) void fused_rb_kernel(const FusedRbParams p) {
    static_assert(!(X3 && IO16X), "split operands keep float32 activations between kernels");
    static_assert(!(X3 && OPF != OP_BF16), "split operands are bfloat16");
    constexpr int TM = C / 16 / WM;
    constexpr int W = 16 * TN * WN;
    constexpr int NTHREADS = WM * WN * 64;
    constexpr int RS = C * (X3 ? 4 : 2) + kLdsRowPad;
    constexpr int NKC = C / 32;
    constexpr int KV = X3 ? 3 : 1;  // K steps per k-chunk
    using WL = WinLoad<C, W, NTHREADS>;
    using AL = ApronLoad<C, NTHREADS>;
    static_assert(TM * 16 * WM == C, "C must split evenly over the M waves");
    const int npair = ONE ? 1 : p.npair;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int cum_s[kMaxSeq + 1];

    // ---- persistent block: walks its share of the V windows that exist (WindowWalk)
    const int V = window_table(cum_s, p.lengths, p.B, p.Tmax, p.rows_per_frame, p.NT);
    const WindowWalk walk(V);
    int v = walk.v0;
    if (v >= walk.vend) return;
    MV_TRACER(tr);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int lr = lane & 15, lg = lane >> 4;

    char* A16 = smem;                                   // rows [-guardA, W + guardA)
    char* Y16 = smem + (W + 2 * p.guardA) * RS;         // rows [-guardY, W + guardY)

    // zero the guard rows once (they stand for rows outside the window; their values only reach
    // outputs inside the halo, but they must be finite)
    {
        const int upr = RS / 16;
        for (int idx = threadIdx.x; idx < 2 * p.guardA * upr; idx += NTHREADS) {
            const int r = idx / upr, u = idx - r * upr;
            const int row = r < p.guardA ? r : W + r;  // [0,guardA) and [W+guardA, W+2guardA)
            *(u32x4*)(A16 + row * RS + u * 16) = u32x4{0u, 0u, 0u, 0u};
        }
        for (int idx = threadIdx.x; idx < 2 * p.guardY * upr; idx += NTHREADS) {
            const int r = idx / upr, u = idx - r * upr;
            const int row = r < p.guardY ? r : W + r;
            *(u32x4*)(Y16 + row * RS + u * 16) = u32x4{0u, 0u, 0u, 0u};
        }
    }
    char* A0 = A16 + p.guardA * RS;  // window row 0
    char* Y0 = Y16 + p.guardY * RS;
    float* bias_s = (float*)(Y16 + (W + 2 * p.guardY) * RS);  // [2*npair][C]: b1[0], b2[0], b1[1], ...
    for (int idx = threadIdx.x; idx < 2 * npair * C; idx += NTHREADS) {
        const int cv = idx / C, ch = idx - cv * C;
        bias_s[idx] = (cv & 1 ? p.b2[cv >> 1] : p.b1[cv >> 1])[ch];
    }
    __syncthreads();  // guard rows zeroed before the first apron lands in them; bias_s complete

    // this lane's fragment position (n, ch) of tile (tm, tn) as a byte offset into an A / Y tile
    // Output channels are pair-interleaved: MFMA row r of the wave's tile tm is channel 32*wm + 8*(r>>2) + 4*tm +
    // (r&3) (the host packs the weight rows in that order; biases stay in channel order, hifigan.hip), so a lane's two fragments of a row are 8
    // consecutive channels = ONE 16-byte LDS unit: half as many fragment stores / loads, 2-way instead of 4-way
    // bank conflicts on the stores, none on the loads (ablation: conflict-free fragment traffic is worth 6 %).
    static_assert(TM == 2, "pair-interleaved channel order needs two M tiles per wave");
    const int chl = (wm * TM) * 16 + lg * 8;  // first channel of the lane's tile-0 fragment; tile tm: + 4*tm
    const int frag0 = (wn * TN * 16 + lr) * RS + chl * 2;
    constexpr int FRAG_TN = 16 * RS;

    const long long mtile_stride = (long long)p.K * NKC * 64 * KV;
    const long long woff = (long long)(wm * TM) * mtile_stride + lane;
    const int pad2 = (p.K - 1) / 2;
    constexpr bool io16x = IO16X;             // float16 windows come and go through LDS, next one prefetched

    int b = find_seq(cum_s, p.B, v);
    int len = seq_frames(p.lengths, b, p.Tmax) * p.rows_per_frame;
    int t0 = (v - cum_s[b]) * p.NT - p.H;     // global row of window row 0
    if constexpr (io16x) {  // the block's first window arrives synchronously, every later one is prefetched
        u32x4 raw[WL::PER], rawa[AL::PER];
        issue_window16<C, W, NTHREADS>((const _Float16*)p.x + (long long)b * p.batch_stride, t0, len, raw);
        issue_apron16<C, W, NTHREADS>((const _Float16*)p.x + (long long)b * p.batch_stride, t0, len, p.g1, rawa);
        commit_raw16<C, W, NTHREADS>(raw, A0, RS, t0, len);
        commit_apron16<C, W, NTHREADS, OPF>(rawa, A0, RS, t0, len, p.g1);
        __syncthreads();
    }
    MV_LAP(tr, TR_SETUP);

    f32x4 X[TM][TN];  // the residual stream
    f32x4 acc[TM][TN];
    // the first weight fragments of every conv are requested while the conv before it winds down (epilogue,
    // barrier, window hand-over), so no K loop starts by waiting an L2 round trip
    APre<TM> wpre;
    wpre.request((const u32x4*)p.w1[0] + woff, mtile_stride, p.K * NKC * KV);
    f32x4 bias[TM];  // of the conv that runs next, read while the conv before it winds down
    lds_bias(bias, bias_s, 0, C, chl);
    for (;;) {
        const bool interior = t0 >= 0 && t0 + W <= len;  // no row of the window needs masking
        const void* xb = p.io16 ? (const void*)((const _Float16*)p.x + (long long)b * p.batch_stride)
                                : (const void*)((const float*)p.x + (long long)b * p.batch_stride);
        void* ob = p.io16 ? (void*)((_Float16*)p.out + (long long)b * p.batch_stride)
                          : (void*)((float*)p.out + (long long)b * p.batch_stride);
        if constexpr (io16x) {
            // The A tile holds the window as raw float16 (rows outside the sequence are zeros).  Each lane
            // takes its own fragments as the fp32 residual and leaves bf16(lrelu(x)) in their place.
            // (all reads first: one LDS round trip for the lot instead of one per fragment)
            u32x4 rawq[TN];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) rawq[tn] = *(const u32x4*)(A0 + frag0 + tn * FRAG_TN);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                unpack8_f16(rawq[tn], X[0][tn], X[1][tn]);
                *(u32x4*)(A0 + frag0 + tn * FRAG_TN) = pack8_lrelu<OPF, true>(X[0][tn], X[1][tn], 0.1f);
            }
        } else {
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int n = (wn * TN + tn) * 16 + lr;
                const int g = t0 + n;
                const bool in_seq = g >= 0 && g < len;
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) {
                    const int ch = chl + 4 * tm;
                    f32x4 q = {0.f, 0.f, 0.f, 0.f};
                    if (in_seq) q = ld4(xb, (long long)g * C + ch, p.io16);
                    X[tm][tn] = q;
                    if constexpr (X3)
                        store4_x3<C>(A0 + n * RS + ch * 2, lrelu4(q, 0.1f), in_seq);
                    else
                        *(unsigned long long*)(A0 + n * RS + ch * 2) = pack4_bf16_lrelu<OPF>(q, 0.1f, in_seq);
                }
            }
            if constexpr (X3) {
                // apron rows of the first pair (see issue_apron16), from the float32 tensor
                const int total = 2 * p.g1 * (C / 4);
                for (int idx = tid_here(); idx < total; idx += NTHREADS) {
                    const int r = idx / (C / 4), u = idx - r * (C / 4);
                    const int row = r < p.g1 ? r - p.g1 : W + r - p.g1;
                    const int g = t0 + row;
                    const bool ok = g >= 0 && g < len;
                    f32x4 q = {0.f, 0.f, 0.f, 0.f};
                    if (ok) q = ld4(xb, (long long)g * C + u * 4, 0);
                    store4_x3<C>(A0 + row * RS + u * 8, lrelu4(q, 0.1f), ok);
                }
            }
        }
        MV_LAP(tr, TR_LOADX);
        __syncthreads();  // A16 complete
        MV_LAP(tr, TR_BARRIER);

        // the window after this one (its rows are requested while this one is still being computed)
        const int vn = v + walk.step;
        const bool has_next = vn < walk.vend;
        int bn = 0, lenn = 0, t0n = 0;
        if (has_next) {
            bn = find_seq(cum_s, p.B, vn);
            lenn = seq_frames(p.lengths, bn, p.Tmax) * p.rows_per_frame;
            t0n = (vn - cum_s[bn]) * p.NT - p.H;
        }

        for (int m = 0; m < npair; ++m) {
            const int d = p.dil[m];
            const int pad1 = d * (p.K - 1) / 2;
            if (m > 0) {
                MV_LAP(tr, TR_EPI);
                __syncthreads();  // A16 complete
                MV_LAP(tr, TR_BARRIER);
            }
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = bias[tm];
            conv_window<TM, TN, 4, X3, OPF>(acc, A0 + ((wn * TN * 16 + lr) - pad1) * RS + lg * 16, RS, p.K, d, NKC,
                                       (const u32x4*)p.w1[m] + woff, mtile_stride, &wpre);
            MV_LAP(tr, TR_KLOOP);
            lds_bias(bias, bias_s, 2 * m + 1, C, chl);
            wpre.request((const u32x4*)p.w2[m] + woff, mtile_stride, p.K * NKC * KV);
            // Prefetch: conv1 of the last pair is done with the A tile after the next barrier and nothing reads
            // it again until the next window, so the next window's rows are requested here (their HBM latency
            // hides under this epilogue and the barrier) and dropped into the A tile before conv2 starts.
            const bool pf = io16x && has_next && m == npair - 1;
            u32x4 raw[WL::PER], rawa[AL::PER];
            if (pf) {
                issue_window16<C, W, NTHREADS>((const _Float16*)p.x + (long long)bn * p.batch_stride, t0n, lenn, raw);
                issue_apron16<C, W, NTHREADS>((const _Float16*)p.x + (long long)bn * p.batch_stride, t0n, lenn, p.g1,
                                              rawa);
            }
            if constexpr (X3) {
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    const int g = t0 + (wn * TN + tn) * 16 + lr;
                    store8_x3<C>(Y0 + frag0 + tn * FRAG_TN, lrelu4(acc[0][tn], 0.1f), lrelu4(acc[1][tn], 0.1f),
                                 interior || (g >= 0 && g < len));
                }
            } else if (interior) {
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    *(u32x4*)(Y0 + frag0 + tn * FRAG_TN) = pack8_lrelu<OPF>(acc[0][tn], acc[1][tn], 0.1f);
            } else {
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    const int g = t0 + (wn * TN + tn) * 16 + lr;
                    *(u32x4*)(Y0 + frag0 + tn * FRAG_TN) =
                        pack8_bf16_lrelu<OPF>(acc[0][tn], acc[1][tn], 0.1f, g >= 0 && g < len);
                }
            }
            MV_LAP(tr, TR_EPI);
            __syncthreads();  // Y16 complete, A16 free
            MV_LAP(tr, TR_BARRIER);
            if (pf) {
                commit_raw16<C, W, NTHREADS>(raw, A0, RS, t0n, lenn);
                commit_apron16<C, W, NTHREADS, OPF>(rawa, A0, RS, t0n, lenn, p.g1);
            }
            MV_LAP(tr, TR_LOADX);
            const int m_next = m + 1 < npair ? m + 1 : 0;
            // x + bias + conv2(...): the K loop accumulates into the residual registers themselves
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) X[tm][tn] += bias[tm];
            conv_window<TM, TN, 4, X3, OPF>(X, Y0 + ((wn * TN * 16 + lr) - pad2) * RS + lg * 16, RS, p.K, 1, NKC,
                                       (const u32x4*)p.w2[m] + woff, mtile_stride, &wpre);
            MV_LAP(tr, TR_KLOOP);
            lds_bias(bias, bias_s, 2 * m_next, C, chl);
            wpre.request((const u32x4*)p.w1[m_next] + woff, mtile_stride, p.K * NKC * KV);
            if (m + 1 < npair) {
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    const int g = t0 + (wn * TN + tn) * 16 + lr;
                    if constexpr (X3)
                        store8_x3<C>(A0 + frag0 + tn * FRAG_TN, lrelu4(X[0][tn], 0.1f), lrelu4(X[1][tn], 0.1f),
                                     interior || (g >= 0 && g < len));
                    else
                        *(u32x4*)(A0 + frag0 + tn * FRAG_TN) =
                            pack8_bf16_lrelu<OPF>(X[0][tn], X[1][tn], 0.1f, interior || (g >= 0 && g < len));
                }
            }
        }

        MV_LAP(tr, TR_EPI);
        // ---- store the rows the halo has not reached, with the MRF accumulation of models.py:154-160
        if constexpr (io16x) {
            // the result leaves as float16 through the Y tile (the A tile may already hold the next window)
            __syncthreads();  // every wave is done reading Y16 as conv2's operand
            MV_LAP(tr, TR_BARRIER);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) *(u32x4*)(Y0 + frag0 + tn * FRAG_TN) = pack8_f16_sat(X[0][tn], X[1][tn]);
            MV_LAP(tr, TR_EPI);
            __syncthreads();
            MV_LAP(tr, TR_BARRIER);
            store_window16<C, NTHREADS>((_Float16*)ob, Y0, RS, t0, p.H, p.NT, len, p.mode, p.div);
            // no barrier here: the next window's first barrier ("A16 complete") already separates these reads
            // of the Y tile from the next conv1 epilogue that overwrites it
        } else {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const int ch = chl + 4 * tm;
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    const int n = (wn * TN + tn) * 16 + lr;
                    const int g = t0 + n;
                    if (n < p.H || n >= p.H + p.NT || g >= len) continue;
                    const long long e = (long long)g * C + ch;
                    f32x4 q = X[tm][tn];
                    if (p.mode == ACC_ADD) {
                        q += ld4(ob, e, p.io16);
                    } else if (p.mode == ACC_ADD_DIV) {
                        q += ld4(ob, e, p.io16);
                        q[0] = q[0] / p.div;
                        q[1] = q[1] / p.div;
                        q[2] = q[2] / p.div;
                        q[3] = q[3] / p.div;
                    }
                    st4(ob, e, q, p.io16);
                }
            }
        }
        MV_LAP(tr, TR_STORE);
        if (!has_next) break;
        v = vn;
        b = bn;
        len = lenn;
        t0 = t0n;
        if constexpr (!io16x) {
            __syncthreads();  // every wave is done with this window's tiles before the next one lands in them
            MV_LAP(tr, TR_BARRIER);
        }
    }
    MV_FLUSH(tr, (C << 16) | (p.K << 8) | npair);
}

// ---------------------------------------------------------------------------------------------
// fused_stage_kernel: ALL ResBlocks of one MRF stage plus their mean (models.py:154-160) in one
// launch, and -- for the last stage -- the final leaky_relu(0.01) -> conv_post -> tanh
// (models.py:161-163) as well, so the largest activation of the network is read once and never
// written.  Same window scheme as fused_rb_kernel; every ResBlock restarts from the stage input x
// (re-read from L2 into registers) and the running sum of their outputs stays in registers.
// The window halo is that of the widest ResBlock (k = 11: 60 rows a side).
#ifndef MV_STAGE_ADEPTH
#define MV_STAGE_ADEPTH 2  // K steps of weights in flight in the stage kernels (register-bound)
#endif

struct FusedStageParams {
    const void* x;   // [B][rows][C] float32, or float16 when io16
    void* out;       // [B][rows][C] mean of the ResBlocks, same type (unused when post_w is set)
    int io16;
    long long batch_stride;
    const int32_t* lengths;
    int B;
    int Tmax;
    int rows_per_frame;
    int nrb;
    int K[3];
    int dil[3][3];
    const void* w1[3][3];
    const void* w2[3][3];
    const float* b1[3][3];
    const float* b2[3][3];
    float div;
    int NT;      // rows written per block
    int H;       // halo per side (incl. 3 rows for conv_post when fused)
    int guardA;
    int guardY;
    int g1[3];   // apron rows of each ResBlock's first pair (see issue_apron16), 0 on the float32 path
    // fused conv_post (last stage): weights [7][C] fp32, bias [1]; output [B][Tmax*rows_per_frame]
    const float* post_w;
    const float* post_b;
    void* wav;
    int wav_pcm16;
};

// IO16 (p.io16) and POST (p.post_w != null) are compile-time: with the four variants in one function the
// register allocator budgets for the union of their live ranges and spills ~50 VGPRs in the one that runs.
template <int C, int WM, int WN, int TN, bool IO16, bool POST, int OPF = OP_BF16>
__global__ __launch_bounds__(WM* WN * 64) void fused_stage_kernel(const FusedStageParams p) {
    constexpr int TM = C / 16 / WM;
    constexpr int W = 16 * TN * WN;
    constexpr int NTHREADS = WM * WN * 64;
    constexpr int RS = C * 2 + kLdsRowPad;
    constexpr int NKC = C / 32;
    constexpr int PS = 9;  // floats per row of the conv_post partial table (7 taps, odd stride)
    using WL = WinLoad<C, W, NTHREADS>;
    using AL = ApronLoad<C, NTHREADS>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int cum_s[kMaxSeq + 1];

    // persistent block over the windows that exist (the host zero-fills the waveform beforehand: rows past a
    // sequence's end are defined to be zero and no block visits them)
    const int V = window_table(cum_s, p.lengths, p.B, p.Tmax, p.rows_per_frame, p.NT);
    const WindowWalk walk(V);
    int v = walk.v0;
    if (v >= walk.vend) return;
    const long long Lmax = (long long)p.Tmax * p.rows_per_frame;
    MV_TRACER(tr);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int lr = lane & 15, lg = lane >> 4;

    char* A16 = smem;
    char* Y16 = smem + (W + 2 * p.guardA) * RS;
    {
        const int upr = RS / 16;
        for (int idx = threadIdx.x; idx < 2 * p.guardA * upr; idx += NTHREADS) {
            const int r = idx / upr, u = idx - r * upr;
            const int row = r < p.guardA ? r : W + r;
            *(u32x4*)(A16 + row * RS + u * 16) = u32x4{0u, 0u, 0u, 0u};
        }
        for (int idx = threadIdx.x; idx < 2 * p.guardY * upr; idx += NTHREADS) {
            const int r = idx / upr, u = idx - r * upr;
            const int row = r < p.guardY ? r : W + r;
            *(u32x4*)(Y16 + row * RS + u * 16) = u32x4{0u, 0u, 0u, 0u};
        }
    }
    char* A0 = A16 + p.guardA * RS;
    char* Y0 = Y16 + p.guardY * RS;
    float* bias_s = (float*)(Y16 + (W + 2 * p.guardY) * RS);  // [nrb][3][2][C], see lds_bias
    for (int idx = threadIdx.x; idx < 6 * p.nrb * C; idx += NTHREADS) {
        const int cv = idx / C, ch = idx - cv * C;
        const int jm = cv >> 1, jj = jm / 3, mm = jm - 3 * jj;
        bias_s[idx] = (cv & 1 ? p.b2[jj][mm] : p.b1[jj][mm])[ch];
    }
    __syncthreads();  // guard rows zeroed before the first apron lands in them; bias_s complete
    // Output channels are pair-interleaved: MFMA row r of the wave's tile tm is channel 32*wm + 8*(r>>2) + 4*tm +
    // (r&3) (the host packs the weight rows in that order; biases stay in channel order, hifigan.hip), so a lane's two fragments of a row are 8
    // consecutive channels = ONE 16-byte LDS unit: half as many fragment stores / loads, 2-way instead of 4-way
    // bank conflicts on the stores, none on the loads (ablation: conflict-free fragment traffic is worth 6 %).
    static_assert(TM == 2, "pair-interleaved channel order needs two M tiles per wave");
    const int chl = (wm * TM) * 16 + lg * 8;  // first channel of the lane's tile-0 fragment; tile tm: + 4*tm
    const int frag0 = (wn * TN * 16 + lr) * RS + chl * 2;
    constexpr int FRAG_TN = 16 * RS;

    int b = find_seq(cum_s, p.B, v);
    int len = seq_frames(p.lengths, b, p.Tmax) * p.rows_per_frame;
    int out0 = (v - cum_s[b]) * p.NT;
    if constexpr (IO16) {  // the block's first window arrives synchronously, every later one is prefetched
        u32x4 raw[WL::PER], rawa[AL::PER];
        const _Float16* xs = (const _Float16*)p.x + (long long)b * p.batch_stride;
        issue_window16<C, W, NTHREADS>(xs, out0 - p.H, len, raw);
        issue_apron16<C, W, NTHREADS>(xs, out0 - p.H, len, p.g1[0], rawa);
        commit_raw16<C, W, NTHREADS>(raw, A0, RS, out0 - p.H, len);
        commit_apron16<C, W, NTHREADS, OPF>(rawa, A0, RS, out0 - p.H, len, p.g1[0]);
        __syncthreads();
    }
    MV_LAP(tr, TR_SETUP);

    // The running sum of the ResBlocks' outputs is parked as packed float16 between ResBlocks (24 registers
    // instead of 48: with fp32 the compiler spilled half of it to scratch, ~1 GB of HBM traffic per launch);
    // it is added to in fp32 and rounded once per ResBlock (2^-11 relative, the storage format of the result).
    f32x4 X[TM][TN], acc[TM][TN];
    unsigned long long SUM16[TM][TN];
    constexpr int ADEPTH = MV_STAGE_ADEPTH;
    APre<TM, ADEPTH> wpre;  // see fused_rb_kernel
    wpre.request((const u32x4*)p.w1[0][0] + (long long)(wm * TM) * ((long long)p.K[0] * NKC * 64) + lane,
                 (long long)p.K[0] * NKC * 64, p.K[0] * NKC);
    f32x4 bias[TM];  // of the conv that runs next, see fused_rb_kernel
    lds_bias(bias, bias_s, 0, C, chl);
    const float inv_div = 1.f / p.div;
    for (;;) {
        const int t0 = out0 - p.H;
        const bool interior = t0 >= 0 && t0 + W <= len;
        const void* xb = IO16 ? (const void*)((const _Float16*)p.x + (long long)b * p.batch_stride)
                                : (const void*)((const float*)p.x + (long long)b * p.batch_stride);
        const int vn = v + walk.step;
        const bool has_next = vn < walk.vend;
        int bn = 0, lenn = 0, out0n = 0;
        if (has_next) {
            bn = find_seq(cum_s, p.B, vn);
            lenn = seq_frames(p.lengths, bn, p.Tmax) * p.rows_per_frame;
            out0n = (vn - cum_s[bn]) * p.NT;
        }
        for (int j = 0; j < p.nrb; ++j) {
            const int K = p.K[j];
            const long long mtile_stride = (long long)K * NKC * 64;
            const long long woff = (long long)(wm * TM) * mtile_stride + lane;
            const int pad2 = (K - 1) / 2;
            if constexpr (IO16) {
                // (every ResBlock restarts from the stage input: its rows were re-requested under the previous
                // ResBlock's last conv1 epilogue and sit in the A tile as raw float16, like a new window's)
                // raw float16 window in the A tile -> fp32 residual registers + bf16(lrelu) in place
                // (all reads first: one LDS round trip for the lot instead of one per fragment)
                u32x4 rawq[TN];
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) rawq[tn] = *(const u32x4*)(A0 + frag0 + tn * FRAG_TN);
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    unpack8_f16(rawq[tn], X[0][tn], X[1][tn]);
                    *(u32x4*)(A0 + frag0 + tn * FRAG_TN) = pack8_lrelu<OPF, true>(X[0][tn], X[1][tn], 0.1f);
                }
            } else {
                if (j > 0 || v != walk.v0) __syncthreads();  // every wave is done with the tiles
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    const int n = (wn * TN + tn) * 16 + lr;
                    const int g = t0 + n;
                    const bool in_seq = g >= 0 && g < len;
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) {
                        const int ch = chl + 4 * tm;
                        f32x4 q = {0.f, 0.f, 0.f, 0.f};
                        if (in_seq) q = ld4(xb, (long long)g * C + ch, 0);
                        X[tm][tn] = q;
                        *(unsigned long long*)(A0 + n * RS + ch * 2) = pack4_bf16_lrelu<OPF>(q, 0.1f, in_seq);
                    }
                }
            }
            MV_LAP(tr, TR_LOADX);
            __syncthreads();  // A16 complete
            MV_LAP(tr, TR_BARRIER);
            for (int m = 0; m < 3; ++m) {
                const int d = p.dil[j][m];
                const int pad1 = d * (K - 1) / 2;
                if (m > 0) {
                    MV_LAP(tr, TR_EPI);
                    __syncthreads();
                    MV_LAP(tr, TR_BARRIER);
                }
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = bias[tm];
                conv_window<TM, TN, ADEPTH, false, OPF>(acc, A0 + ((wn * TN * 16 + lr) - pad1) * RS + lg * 16, RS, K, d, NKC,
                                    (const u32x4*)p.w1[j][m] + woff, mtile_stride, &wpre);
                MV_LAP(tr, TR_KLOOP);
                lds_bias(bias, bias_s, (j * 3 + m) * 2 + 1, C, chl);
                wpre.request((const u32x4*)p.w2[j][m] + woff, mtile_stride, K * NKC);
                // Under the last conv1 epilogue of a ResBlock: request the rows the next ResBlock starts from -- the
                // same window again (L2-hot), or the next window after the last ResBlock (see fused_rb_kernel).
                const bool last_rb = j == p.nrb - 1;
                const bool pf = IO16 && m == 2 && (!last_rb || has_next);
                u32x4 raw[WL::PER], rawa[AL::PER];
                // (one call site each: the source is selected, not the code duplicated)
                const _Float16* xpf = last_rb ? (const _Float16*)p.x + (long long)bn * p.batch_stride : (const _Float16*)xb;
                const int t0pf = last_rb ? out0n - p.H : t0, lenpf = last_rb ? lenn : len;
                const int g1pf = p.g1[last_rb ? 0 : j + 1];
                if (pf) {
                    issue_window16<C, W, NTHREADS>(xpf, t0pf, lenpf, raw);
                    issue_apron16<C, W, NTHREADS>(xpf, t0pf, lenpf, g1pf, rawa);
                }
                if (interior) {
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        *(u32x4*)(Y0 + frag0 + tn * FRAG_TN) =
                            pack8_lrelu<OPF>(acc[0][tn], acc[1][tn], 0.1f);
                } else {
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) {
                        const int g = t0 + (wn * TN + tn) * 16 + lr;
                        *(u32x4*)(Y0 + frag0 + tn * FRAG_TN) =
                            pack8_bf16_lrelu<OPF>(acc[0][tn], acc[1][tn], 0.1f, g >= 0 && g < len);
                    }
                }
                MV_LAP(tr, TR_EPI);
                __syncthreads();
                MV_LAP(tr, TR_BARRIER);
                if (pf) {
                    commit_raw16<C, W, NTHREADS>(raw, A0, RS, t0pf, lenpf);
                    commit_apron16<C, W, NTHREADS, OPF>(rawa, A0, RS, t0pf, lenpf, g1pf);
                }
                MV_LAP(tr, TR_LOADX);
                // x + bias + conv2(...): the K loop accumulates into the residual registers themselves
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) X[tm][tn] += bias[tm];
                conv_window<TM, TN, ADEPTH, false, OPF>(X, Y0 + ((wn * TN * 16 + lr) - pad2) * RS + lg * 16, RS, K, 1, NKC,
                                    (const u32x4*)p.w2[j][m] + woff, mtile_stride, &wpre);
                MV_LAP(tr, TR_KLOOP);
                {
                    // next conv: the next pair of this ResBlock, or the next ResBlock's (the next window's) first
                    const int jn = m < 2 ? j : (j + 1 < p.nrb ? j + 1 : 0);
                    const int mn = m < 2 ? m + 1 : 0;
                    const long long msn = (long long)p.K[jn] * NKC * 64;
                    lds_bias(bias, bias_s, (jn * 3 + mn) * 2, C, chl);
                    wpre.request((const u32x4*)p.w1[jn][mn] + (long long)(wm * TM) * msn + lane, msn, p.K[jn] * NKC);
                }
                if (m < 2) {
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) {
                        const int g = t0 + (wn * TN + tn) * 16 + lr;
                        *(u32x4*)(A0 + frag0 + tn * FRAG_TN) =
                            pack8_bf16_lrelu<OPF>(X[0][tn], X[1][tn], 0.1f, interior || (g >= 0 && g < len));
                    }
                }
            }
            if (j + 1 < p.nrb) {
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        SUM16[tm][tn] = pack4_f16_sat(j == 0 ? X[tm][tn] : unpack4_f16(SUM16[tm][tn]) + X[tm][tn]);
            }
        }
        // mean over the ResBlocks (models.py:160); the last ResBlock's output is still in fp32 registers
        f32x4 SUM[TM][TN];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                SUM[tm][tn] = p.nrb > 1 ? unpack4_f16(SUM16[tm][tn]) + X[tm][tn] : X[tm][tn];
#pragma unroll
                for (int i = 0; i < 4; ++i) SUM[tm][tn][i] = SUM[tm][tn][i] * inv_div;
            }

        MV_LAP(tr, TR_EPI);
        // The result leaves through the Y tile (the A tile may already hold the next window).
        __syncthreads();  // every wave is done reading Y16 as the last conv2's operand
        MV_LAP(tr, TR_BARRIER);
        if constexpr (!POST) {
            if constexpr (IO16) {
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    *(u32x4*)(Y0 + frag0 + tn * FRAG_TN) = pack8_f16_sat(SUM[0][tn], SUM[1][tn]);
                MV_LAP(tr, TR_EPI);
                __syncthreads();
                MV_LAP(tr, TR_BARRIER);
                store_window16<C, NTHREADS>((_Float16*)p.out + (long long)b * p.batch_stride, Y0, RS, t0, p.H, p.NT,
                                            len, ACC_STORE, 1.f);
            } else {
                float* ob = (float*)p.out + (long long)b * p.batch_stride;
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) {
                    const int ch = chl + 4 * tm;
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) {
                        const int n = (wn * TN + tn) * 16 + lr;
                        const int g = t0 + n;
                        if (n < p.H || n >= p.H + p.NT || g >= len) continue;
                        st4(ob, (long long)g * C + ch, SUM[tm][tn], 0);
                    }
                }
            }
            MV_LAP(tr, TR_STORE);
        } else {
            // ---- fused conv_post (models.py:161-163): y[t] = tanh(b + sum_k sum_c w[k][c] * lrelu(mean, 0.01)[t+k-3][c]).
            // Pass 1, from registers: every lane dots its 4*TM channels of each of its rows with the 7 taps, the
            // four lanes (lg) that share a row add up through ds_bpermute, and P[wm][n][k] lands in LDS (in the Y
            // tile).  Pass 2: one output sample per thread sums 7*WM entries of P.
            float* P = (float*)Y0;  // [WM][W][PS], clear of the (zero) guard rows
            {
                f32x4 wk[7][TM];
#pragma unroll
                for (int k = 0; k < 7; ++k)
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) wk[k][tm] = *(const f32x4*)(p.post_w + k * C + chl + 4 * tm);
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    const int n = (wn * TN + tn) * 16 + lr;
                    const int g = t0 + n;
                    const bool keep = g >= 0 && g < len;
                    float part[7];
#pragma unroll
                    for (int k = 0; k < 7; ++k) part[k] = 0.f;
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) {
                        f32x4 q = lrelu4(SUM[tm][tn], 0.01f);
                        if (!keep) q = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int k = 0; k < 7; ++k) {
                            part[k] += q[0] * wk[k][tm][0];
                            part[k] += q[1] * wk[k][tm][1];
                            part[k] += q[2] * wk[k][tm][2];
                            part[k] += q[3] * wk[k][tm][3];
                        }
                    }
#pragma unroll
                    for (int k = 0; k < 7; ++k) {
                        part[k] += __shfl_xor(part[k], 16);
                        part[k] += __shfl_xor(part[k], 32);
                    }
                    // lane lg writes taps lg and lg + 4
                    float* row = P + ((size_t)wm * W + n) * PS;
                    const float lo = lg == 0 ? part[0] : lg == 1 ? part[1] : lg == 2 ? part[2] : part[3];
                    const float hi = lg == 0 ? part[4] : lg == 1 ? part[5] : part[6];
                    row[lg] = lo;
                    if (lg < 3) row[lg + 4] = hi;
                }
            }
            MV_LAP(tr, TR_POST);
            __syncthreads();
            MV_LAP(tr, TR_BARRIER);
            const float bias = p.post_b[0];
            for (int i = tid_here(); i < p.NT; i += NTHREADS) {
                const int n = p.H + i;
                const long long t = (long long)out0 + i;
                if (t >= Lmax) break;
                float y = 0.f;
                if (t < len) {
                    float a = bias;
#pragma unroll
                    for (int k = 0; k < 7; ++k)
#pragma unroll
                        for (int w = 0; w < WM; ++w) a += P[((size_t)w * W + (n + k - 3)) * PS + k];
                    y = tanhf(a);
                }
                if (p.wav_pcm16)
                    ((int16_t*)p.wav)[b * Lmax + t] = pcm16_of(y);
                else
                    ((float*)p.wav)[b * Lmax + t] = y;
            }
            MV_LAP(tr, TR_POST);
        }
        if (!has_next) break;
        v = vn;
        b = bn;
        len = lenn;
        out0 = out0n;
        // io16: the barrier after the next window's in-place conversion separates these reads of the Y tile
        // from the next conv1 epilogue that overwrites it; the fp32 path syncs at the top of the loop
    }
    MV_FLUSH(tr, (C << 16) | (POST ? 0xfe00 : 0xff00) | p.nrb);
}

}  // namespace mv
