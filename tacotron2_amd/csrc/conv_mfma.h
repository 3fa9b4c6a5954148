// conv_mfma.h -- the implicit-GEMM 1-D convolution kernel of libmelvoc (gfx950 only).
//
// One kernel covers every contraction of the two vocoders:
//   Conv1d (any taps / dilation)           hifigan/models.py:99,101,150   waveglow/glow.py:164,168
//   ConvTranspose1d as a polyphase conv    hifigan/models.py:153          waveglow/glow.py:252
//   1x1 convs and the cond_layer slice fused as a second input            waveglow/glow.py:159,168
//
// GEMM view:  D[m][n] = sum_pass sum_tap sum_c  W[m][pass,tap,c] * X_pass[n + tap*dil - pad][c]
//   M = output channels (for a transposed conv: phase-major  m = phase*Cout + co)
//   N = time, K = taps x input channels.
// Activations are float32, channels-last ([B][rows][C]) in HBM; a block stages the rows its
// N-tile needs (halo included) into LDS, applying the consumer's leaky_relu and -- in bf16
// mode -- the f32->bf16 rounding on the way, with rows outside [0, len) written as zeros:
// that is the per-sequence zero padding of the reference's batch-1 call.  Weights are
// pre-packed per 16-row M tile in MFMA A-fragment order and stream L2 -> registers.
//
// MFMA: bf16 mode  v_mfma_f32_16x16x32_bf16  (A,B: 8 bf16 per lane, k = 8*(lane>>4)+j)
//       fp32 mode  v_mfma_f32_16x16x4_f32 x4 (A,B: 4 f32 per lane; MFMA j takes element j,
//                  so both operands still load as one 16-byte vector per lane)
//       C/D: col = lane&15 (time), row = 4*(lane>>4)+reg (output channel).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mv {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int kMaxPass = 16;
constexpr int kMfmaPrio = 3;    // s_setprio while a wave issues an MFMA group: the SIMD's other wave does its loads meanwhile
constexpr int kLdsRowPad = 32;  // bytes; row stride/16 == 2 (mod 4) makes ds_read_b128 conflict-free

enum { EPI_LINEAR = 0, EPI_GATE = 1 };
// 16-bit MFMA operand formats (template parameter OPF of the "BF16" kernels): bfloat16 (MV_PREC_BF16, MV_PREC_BF16X3) or
// float16 (MV_PREC_F16: v_mfma_f32_16x16x32_f16 takes the same cycles; 11 significant bits instead of 8, range +-65504,
// so every conversion to it saturates)
enum { OP_BF16 = 0, OP_F16 = 1 };
enum { ACC_STORE = 0, ACC_ADD = 1, ACC_ADD_DIV = 2 };

struct ConvPass {
    const void* src;         // channels-last [B][rows][row_stride], float32 or (src_f16) float16
    long long batch_stride;  // elements
    int row_stride;          // elements
    int cin_off;             // first channel staged by this pass
    int cb;                  // channels staged (multiple of 32 in bf16 mode, 16 in fp32 mode)
    int ktaps;
    int dil;
    int pad_left;            // input row of (n, tap) = n + tap*dil - pad_left
    int rows_per_frame;      // valid input rows of sequence b = lengths[b] * rows_per_frame
    float slope;             // leaky_relu slope applied while staging (1 = identity)
    int src_f16;             // source storage: 0 float32, 1 float16 (bf16 mode's activations), 2 bfloat16 (a tensor
                             // that is only ever an MFMA operand: staged by plain 16-byte copies, no activation)
};

struct ConvOut {
    void* ptr;                   // float32, or float16 when f16 is set
    long long batch_stride;      // elements
    long long off;               // element offset added to n*row_stride + m (may be negative)
    long long valid_per_frame;   // elements e with 0 <= e < lengths[b]*valid_per_frame are written
    int row_stride;              // elements per n
    int mode;                    // ACC_*
    int f16;                     // storage: 0 float32, 1 float16, 2 bfloat16 (plain stores of operand-only tensors)
    void* copy_bf16;             // optional second copy of what is stored, as bfloat16, same indexing (the MFMA-operand
                                 // view of a float32 residual stream)
};

struct ConvParams {
    ConvPass pass[kMaxPass];
    int npass;
    const void* w;       // [M/16][steps_total][64 lanes][16 bytes]
    int steps_total;
    const float* bias;   // [M]
    const int32_t* lengths;
    int Tmax;            // frames
    int n_rows_per_frame;
    int n_extra;         // N extent of sequence b = lengths[b]*n_rows_per_frame + n_extra
    int split;           // m >= split goes to out[1] with m -= split
    ConvOut out[2];
    const void* resid;   // optional, indexed (and typed) like out[0]
    long long resid_batch_stride;
    float div;           // ACC_ADD_DIV divisor
    int act;             // linear epilogue: 0 = none, 1 = tanh(acc + bias) before residual / accumulate (model.py:172)
    int epi_lds;         // linear epilogue, one float16 output: leave through LDS with 16-byte coalesced stores
    int xcd_ny;          // > 0: 1-D grid; the ny M-blocks of a row tile run back to back on ONE XCD (see conv_cl_kernel)
    int nx;              // row tiles (xcd_ny > 0)
};

template <bool BF16, int OPF = OP_BF16>
__device__ __forceinline__ void mma16(f32x4& acc, const u32x4& a, const u32x4& b) {
#if defined(MV_ABLATE) && (MV_ABLATE & 1)
    // timing-only build (tools/ablate.sh; MV_ABLATE is a bit mask): keep the operands live, issue no MFMA
    asm volatile("" ::"v"(a), "v"(b));
    return;
#endif
    if constexpr (BF16 && OPF == OP_F16) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0,
                                                     0, 0);
    } else if constexpr (BF16) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                      acc, 0, 0, 0);
    } else {
        const f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0], bf[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1], bf[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[2], bf[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[3], bf[3], acc, 0, 0, 0);
    }
}

__device__ __forceinline__ float lrelu(float x, float slope) { return x >= 0.f ? x : x * slope; }

// `32768.0 * y` -> host -> numpy astype('int16') as the reference's driver executes it (do_syn.py:11,389-394, x86
// host: cvttss2si to int32, then the low 16 bits): truncation toward zero, NO clip, wrap-around on overflow, and 0 for
// NaN or |32768 y| >= 2^31 (the conversion's "integer indefinite" 0x80000000).  Spelled out because the plain C++ cast
// is undefined there and the GPU's own conversion saturates instead.
__device__ __forceinline__ int16_t pcm16_of(float y) {
    const float v = 32768.0f * y;
    if (!(v >= -2147483648.f && v < 2147483648.f)) return 0;
    return (int16_t)(int32_t)v;
}

// Frames of sequence b.  lengths[] comes from the caller of the C ABI: it is clamped into [0, Tmax] where it is read,
// so an out-of-range entry cannot walk a kernel out of the [B][Tmax * ..] buffers (NULL = every sequence is Tmax long).
__device__ __forceinline__ int seq_frames(const int32_t* __restrict__ lengths, int b, int Tmax) {
    if (!lengths) return Tmax;
    const int v = lengths[b];
    return v < 0 ? 0 : (v > Tmax ? Tmax : v);
}

// tanh(t) * sigmoid(s) (waveglow/glow.py:33-40).  fp32 mode: libm.  bf16 mode: (e^2t - 1) / ((e^2t + 1)(1 + e^-s))
// on the hardware exp2 / rcp (two v_exp_f32 + one v_rcp_f32, ~1 ulp each, against ~3x the instructions for
// tanhf + expf + an IEEE divide; the gate epilogue was a third of the gated conv's MFMA time).  t is clamped to
// +-15 (tanh is +-1 to fp32 precision beyond 9.1) so e^2t stays finite; e^-s may overflow to inf, which gives the
// correct limit 0.
template <bool FAST>
__device__ __forceinline__ float gate_act(float t, float s) {
    if constexpr (FAST) {
        t = fminf(fmaxf(t, -15.f), 15.f);
        const float e2t = __expf(2.f * t);
        const float es = __expf(-s);
        return (e2t - 1.f) * __builtin_amdgcn_rcpf((e2t + 1.f) * (1.f + es));
    } else {
        return tanhf(t) * (1.0f / (1.0f + expf(-s)));
    }
}

// 4 consecutive activations at element index e of a float32 / float16 tensor.  In bf16 mode the
// activations that travel between kernels are stored as float16 (11-bit mantissa: 2^-11 relative
// rounding per store, far below the bf16 operand rounding; halves every HBM pass); stores saturate
// at +-65504 instead of producing inf.
__device__ __forceinline__ f32x4 ld4(const void* base, long long e, int f16) {
    if (f16) {
        const f16x4 h = *(const f16x4*)((const _Float16*)base + e);
        return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    }
    return *(const f32x4*)((const float*)base + e);
}
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi);
template <int OPF>
__device__ __forceinline__ unsigned pack_op(float lo, float hi);
template <int OPF = OP_BF16>
__device__ __forceinline__ void st4(void* base, long long e, const f32x4& v, int f16) {
    if (f16 == 2) {  // MFMA-operand format (operand-only tensors, see ConvPass::src_f16)
        *(unsigned long long*)((unsigned short*)base + e) =
            ((unsigned long long)pack_op<OPF>(v[2], v[3]) << 32) | pack_op<OPF>(v[0], v[1]);
    } else if (f16) {
        f16x4 h;
#pragma unroll
        for (int i = 0; i < 4; ++i) h[i] = (_Float16)fminf(fmaxf(v[i], -65504.f), 65504.f);
        *(f16x4*)((_Float16*)base + e) = h;
    } else {
        *(f32x4*)((float*)base + e) = v;
    }
}

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, v);
}
// two values as MFMA operands of format OPF (round to nearest even; float16 saturates at +-65504: an operand that
// became inf would turn every product with a zero into NaN)
template <int OPF>
__device__ __forceinline__ unsigned pack_op(float lo, float hi) {
    if constexpr (OPF == OP_F16) {
        typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
        f16x2 v = {(_Float16)__builtin_amdgcn_fmed3f(lo, -65504.f, 65504.f),
                   (_Float16)__builtin_amdgcn_fmed3f(hi, -65504.f, 65504.f)};
        return __builtin_bit_cast(unsigned, v);
    } else {
        return pack_bf16(lo, hi);
    }
}

// The same for a bfloat16 source (ConvPass::src_f16 == 2): the tile is a plain copy, 16 bytes = 8 channels per unit.
template <int NTHREADS>
__device__ __forceinline__ void stage_tile_raw(char* smem, const ConvPass& cp, const void* src_b, int row0, int rows,
                                               int len_rows, int RS) {
    constexpr int U = 8;
    const int upr = cp.cb / 8;
    const int total = rows * upr;
    for (int base = threadIdx.x; base < total; base += NTHREADS * U) {
        u32x4 x[U];
        int dst[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int idx = base + k * NTHREADS;
            const int r = idx / upr;
            const int u = idx - r * upr;
            const int in_row = row0 + r;
            dst[k] = idx < total ? r * RS + u * 16 : -1;
            x[k] = u32x4{0u, 0u, 0u, 0u};
            if (idx < total && in_row >= 0 && in_row < len_rows)
                x[k] = *(const u32x4*)((const unsigned short*)src_b + (long long)in_row * cp.row_stride + cp.cin_off + u * 8);
        }
#pragma unroll
        for (int k = 0; k < U; ++k)
            if (dst[k] >= 0) *(u32x4*)(smem + dst[k]) = x[k];
    }
}

// Stage rows [row0, row0+rows) x channels [cin_off, cin_off+cb) of one sequence into LDS.
// Loads are issued U units at a time before any is consumed, so a block that is alone on its CU
// pays the HBM latency once per batch instead of once per 16-byte unit.
// Split-operand mode (MV_PREC_BF16X3): a float32 value travels as TWO bfloat16 operands, hi = bf16(x) and
// lo = bf16(x - hi); a product w*x is formed as w_hi*x_hi + w_hi*x_lo + w_lo*x_hi on the bf16 MFMA with fp32
// accumulation (the dropped w_lo*x_lo and the residue of the split are <= 2^-16 relative: measured 5e-5 max-abs on
// the waveform against 1e-3 allowed for fp32 results, at a third of the bf16 MFMA rate instead of a sixteenth).
// A staged row holds its cb channels twice: [hi: cb bf16][lo: cb bf16].
__device__ __forceinline__ void split_bf16x2(const f32x4& a, const f32x4& b, u32x4& hi, u32x4& lo) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const __bf16 h0 = (__bf16)v[2 * i], h1 = (__bf16)v[2 * i + 1];
        const bf16x2 hh = {h0, h1};
        const bf16x2 ll = {(__bf16)(v[2 * i] - (float)h0), (__bf16)(v[2 * i + 1] - (float)h1)};
        hi[i] = __builtin_bit_cast(unsigned, hh);
        lo[i] = __builtin_bit_cast(unsigned, ll);
    }
}

template <bool BF16, int NTHREADS, bool X3 = false, int OPF = OP_BF16>
__device__ __forceinline__ void stage_tile(char* smem, const ConvPass& cp, const void* src_b, int row0, int rows,
                                           int len_rows, int RS) {
    constexpr int UC = BF16 ? 8 : 4;  // channels per 16-byte LDS unit
    constexpr int U = 8;              // units in flight per thread
    const int upr = cp.cb / UC;
    const int total = rows * upr;
    const float slope = cp.slope;
    for (int base = threadIdx.x; base < total; base += NTHREADS * U) {
        f32x4 x0[U], x1[BF16 ? U : 1];
        int dst[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int idx = base + k * NTHREADS;
            const int r = idx / upr;
            const int u = idx - r * upr;
            const int in_row = row0 + r;
            dst[k] = idx < total ? r * RS + u * 16 : -1;
            x0[k] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (BF16) x1[k] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (idx < total && in_row >= 0 && in_row < len_rows) {
                const long long e = (long long)in_row * cp.row_stride + cp.cin_off + u * UC;
                if (cp.src_f16) {
                    if constexpr (BF16) {
                        const f16x8 h = *(const f16x8*)((const _Float16*)src_b + e);
                        x0[k] = f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
                        x1[k] = f32x4{(float)h[4], (float)h[5], (float)h[6], (float)h[7]};
                    } else {
                        x0[k] = ld4(src_b, e, 1);
                    }
                } else {
                    const float* g = (const float*)src_b + e;
                    x0[k] = *(const f32x4*)g;
                    if constexpr (BF16) x1[k] = *(const f32x4*)(g + 4);
                }
            }
        }
        const bool act = slope != 1.f;  // slope 1 = no activation (WaveGlow, conv_pre): skip the max/mul
#pragma unroll
        for (int k = 0; k < U; ++k) {
            if (dst[k] < 0) continue;
            u32x4 v;
            f32x4 y0 = x0[k];
            if (act) {
                const f32x4 s0 = y0 * slope;  // leaky_relu as max(x, slope*x), 0 < slope < 1
                y0 = f32x4{fmaxf(y0[0], s0[0]), fmaxf(y0[1], s0[1]), fmaxf(y0[2], s0[2]), fmaxf(y0[3], s0[3])};
            }
            if constexpr (BF16) {
                f32x4 y1 = x1[k];
                if (act) {
                    const f32x4 s1 = y1 * slope;
                    y1 = f32x4{fmaxf(y1[0], s1[0]), fmaxf(y1[1], s1[1]), fmaxf(y1[2], s1[2]), fmaxf(y1[3], s1[3])};
                }
                if constexpr (X3) {
                    u32x4 lo;
                    split_bf16x2(y0, y1, v, lo);
                    *(u32x4*)(smem + dst[k] + cp.cb * 2) = lo;
                } else {
                    v[0] = pack_op<OPF>(y0[0], y0[1]);
                    v[1] = pack_op<OPF>(y0[2], y0[3]);
                    v[2] = pack_op<OPF>(y1[0], y1[1]);
                    v[3] = pack_op<OPF>(y1[2], y1[3]);
                }
            } else {
                v = __builtin_bit_cast(u32x4, y0);
            }
            *(u32x4*)(smem + dst[k]) = v;
        }
    }
}

// B-fragment pipeline: N tiles are consumed in groups of G; while the MFMAs of group g run, the
// ds_read_b128 of group g+1 (or of the next K step's group 0) are already in flight, so one wave
// per SIMD keeps the matrix pipe busy without relying on other waves to hide the LDS latency.
template <int TM, int TN>
struct BPipe {
    static constexpr int G = (8 + TM - 1) / TM < TN ? (8 + TM - 1) / TM : TN;  // >= 8 MFMAs per group
    static constexpr int NG = (TN + G - 1) / G;
};

template <int TN, int G>
__device__ __forceinline__ void load_bgroup(u32x4 (&q)[G], const char* bp, int RS, int g) {
#pragma unroll
    for (int j = 0; j < G; ++j)
        if (g * G + j < TN) {
#if defined(MV_ABLATE) && (MV_ABLATE & 2)
            q[j] = u32x4{(unsigned)(size_t)bp, 0u, 0u, 0u};  // timing-only build: no LDS read
#elif defined(MV_ABLATE) && (MV_ABLATE & 16)
            if (j & 1)  // timing-only build: half the LDS reads
                q[j] = u32x4{(unsigned)(size_t)bp, 0u, 0u, 0u};
            else
                q[j] = *(const u32x4*)(bp + (g * G + j) * 16 * RS);
#else
            q[j] = *(const u32x4*)(bp + (g * G + j) * 16 * RS);
#endif
        }
}

// One K step.  PAR = which of the two buffers holds this step's group 0 on entry; on exit the next
// step's group 0 (read through bp_next) sits in buffer (PAR + NG) & 1.
#if defined(MV_ABLATE) && (MV_ABLATE & 64)
#define MV_XA 1  // timing-only build: one extra L2-loaded fragment per K step rides along (stands for an LDS fill)
#else
#define MV_XA 0
#endif

template <bool BF16, int TM, int TN, int PAR, int OPF = OP_BF16>
__device__ __forceinline__ void mma_step(f32x4 (&acc)[TM][TN], const u32x4 (&a)[TM + MV_XA], const char* bp,
                                         const char* bp_next, int RS, u32x4 (&q0)[BPipe<TM, TN>::G],
                                         u32x4 (&q1)[BPipe<TM, TN>::G]) {
    constexpr int G = BPipe<TM, TN>::G, NG = BPipe<TM, TN>::NG;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const bool cur0 = ((g + PAR) & 1) == 0;
        if (g + 1 < NG) {
            if (cur0)
                load_bgroup<TN, G>(q1, bp, RS, g + 1);
            else
                load_bgroup<TN, G>(q0, bp, RS, g + 1);
        } else {
            if (cur0)
                load_bgroup<TN, G>(q1, bp_next, RS, 0);
            else
                load_bgroup<TN, G>(q0, bp_next, RS, 0);
        }
        // pin the order: hipcc otherwise sinks every ds_read to just before its MFMA (lgkmcnt(0)
        // after each read), which exposes the full LDS latency at one wave per SIMD
        __builtin_amdgcn_sched_barrier(0);
#if MV_XA
        if (g == 0) asm volatile("" ::"v"(a[TM]));
#endif
        __builtin_amdgcn_s_setprio(kMfmaPrio);
#pragma unroll
        for (int j = 0; j < G; ++j) {
            if (g * G + j < TN) {
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) mma16<BF16, OPF>(acc[tm][g * G + j], a[tm], cur0 ? q0[j] : q1[j]);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// The weight stream of one wave: a scalar buffer descriptor over the wave's first M tile + a 32-bit lane offset.
// The wave-uniform part of the address (every lane of a wave reads the same M tiles) sits in SGPRs, so stepping
// through K costs scalar adds only (global_load took two 64-bit VALU adds per fragment, between the MFMAs).
template <int TM>
struct WStream {
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned lane_off;
    int mtile_bytes;
    int S;  // K steps of the pass (fragments per M tile)

    __device__ __forceinline__ WStream(const u32x4* wlane, long long mtile_stride, int steps) {
        lane_off = (threadIdx.x & 63u) * 16u;
        const unsigned long long wu = (unsigned long long)(wlane - (threadIdx.x & 63u));
        // (readfirstlane returns int: go through unsigned, or the low half sign-extends into the high half)
        const unsigned w_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(wu >> 32));
        const unsigned w_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)wu);
        void* wbase = (void*)(((unsigned long long)w_hi << 32) | (unsigned long long)w_lo);
        rsrc = __builtin_amdgcn_make_buffer_rsrc(wbase, 0, 0x7fffffff, 0x00020000);
        mtile_bytes = (int)(mtile_stride * 16);
        S = steps;
    }
    // fragments of K step `step` (clamped to the last one: prefetches past the end re-read it, harmlessly)
    __device__ __forceinline__ void fetch(u32x4 (&a)[TM + MV_XA], int step) const {
        int sn = step < S ? step : S - 1;
#if defined(MV_ABLATE) && (MV_ABLATE & 256)
        sn &= 1;  // timing-only build: the weight stream re-reads two fragments (L1-resident): no L2 traffic
#endif
#if MV_XA
        a[TM] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off, sn * 1024, 0);
#endif
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
#if defined(MV_ABLATE) && (MV_ABLATE & 4)
            a[tm] = u32x4{(unsigned)sn, 0u, 0u, 0u};  // timing-only build: no weight loads
#elif defined(MV_ABLATE) && (MV_ABLATE & 1024)
            // timing-only build: the weight fragments come out of LDS (whatever sits at its start) instead of L2
            a[tm] = *(const u32x4*)((const __attribute__((address_space(3))) char*)0 + lane_off + ((tm + 2 * sn) & 7) * 1024);
#elif defined(MV_ABLATE) && (MV_ABLATE & 512)
            // timing-only build: one weight load per K step instead of TM
            a[tm] = tm == 0 ? __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off, sn * 1024, 0) : a[0];
#else
            a[tm] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off, tm * mtile_bytes + sn * 1024, 0);
#endif
        }
    }
};

// The first four K steps' fragments of a pass, requested ahead of it (fused kernels: under the previous conv's
// epilogue and barrier, so a conv does not start by waiting an L2 round trip for its first weights).
template <int TM, int DEPTH = 4>
struct APre {
    u32x4 a[DEPTH][TM + MV_XA];
    __device__ __forceinline__ void request(const u32x4* wlane, long long mtile_stride, int steps) {
        const WStream<TM> ws(wlane, mtile_stride, steps);
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) ws.fetch(a[i], i);
    }
};

// One pass of the K loop: S = ktaps * nkc steps.  A fragments stream L2 -> registers FOUR steps
// ahead (a wave keeps 4*TM KiB in flight: with two steps the weight stream, not the matrix pipe,
// paced the loop), B fragments come from LDS one group ahead.  The loop body is branch-free
// (prefetch indices are clamped, not guarded) so the accumulators stay in place.  DEPTH = 2 keeps two steps
// in flight instead of four (16 VGPRs less; for the register-bound stage kernels with their short K loops).
// X3 (split operands, see split_bf16x2): a tap walks 3*nkc K steps over a row's [hi | lo] halves -- the hi chunks
// against w_hi, the lo chunks against w_hi, then the hi chunks again against w_lo (the host packs the weight
// fragments in that order) -- so the B pointer steps back over both halves once per tap.
template <bool BF16, int TM, int TN, int DEPTH = 4, bool X3 = false, int OPF = OP_BF16>
__device__ __forceinline__ void mma_pass(f32x4 (&acc)[TM][TN], const char* lds_b, int RS, int ktaps, int dil, int nkc,
                                         const u32x4* wlane, long long mtile_stride,
                                         const APre<TM, DEPTH>* pre = nullptr) {
    static_assert(DEPTH == 2 || DEPTH == 4, "two or four K steps of weights in flight");
    constexpr int G = BPipe<TM, TN>::G, NG = BPipe<TM, TN>::NG;
    constexpr int P1 = NG & 1;  // buffer parity of the 2nd / 4th step of an unrolled group
    const int nkv = X3 ? 3 * nkc : nkc;  // K steps per tap
    const int S = ktaps * nkv;
    const int tap_wrap = dil * RS - nkc * 64;
    u32x4 a0[TM + MV_XA], a1[TM + MV_XA], a2[DEPTH == 4 ? TM + MV_XA : 1], a3[DEPTH == 4 ? TM + MV_XA : 1];
    u32x4 q0[G], q1[G];
    const WStream<TM> ws(wlane, mtile_stride, S);
    auto fetch = [&](u32x4 (&a)[TM + MV_XA], int step) { ws.fetch(a, step); };
#if defined(MV_ABLATE) && (MV_ABLATE & 8)
    return;  // timing-only build: no K loop at all (staging, epilogues and barriers remain)
#endif
    if (pre) {
#pragma unroll
        for (int t = 0; t < TM + MV_XA; ++t) {
            a0[t] = pre->a[0][t];
            a1[t] = pre->a[1][t];
            if constexpr (DEPTH == 4) {
                a2[t] = pre->a[2][t];
                a3[t] = pre->a[3][t];
            }
        }
    } else {
        fetch(a0, 0);
        fetch(a1, 1);
        if constexpr (DEPTH == 4) {
            fetch(a2, 2);
            fetch(a3, 3);
        }
    }
    int kc = 0;
    const char* bp = lds_b;  // advances 64 B per k-chunk, wraps to the next tap
    load_bgroup<TN, G>(q0, bp, RS, 0);
    auto next_bp = [&](const char* cur) {
        const char* n = cur + 64;
        ++kc;
        if (X3 && kc == 2 * nkc) n -= 2 * nkc * 64;
        if (kc == nkv) {
            kc = 0;
            n += tap_wrap;
        }
        return n;
    };
    int s = 0;
    if constexpr (DEPTH == 2) {
        for (; s + 2 <= S; s += 2) {
            const char* bp1 = next_bp(bp);
            mma_step<BF16, TM, TN, 0, OPF>(acc, a0, bp, bp1, RS, q0, q1);
            fetch(a0, s + 2);
            const char* bp2 = s + 2 < S ? next_bp(bp1) : bp1;
            mma_step<BF16, TM, TN, P1, OPF>(acc, a1, bp1, bp2, RS, q0, q1);
            fetch(a1, s + 3);
            bp = bp2;
        }
        if (s < S) mma_step<BF16, TM, TN, 0, OPF>(acc, a0, bp, bp, RS, q0, q1);
    } else {
    for (; s + 4 <= S; s += 4) {
        const char* bp1 = next_bp(bp);
        mma_step<BF16, TM, TN, 0, OPF>(acc, a0, bp, bp1, RS, q0, q1);
        fetch(a0, s + 4);
        const char* bp2 = next_bp(bp1);
        mma_step<BF16, TM, TN, P1, OPF>(acc, a1, bp1, bp2, RS, q0, q1);
        fetch(a1, s + 5);
        const char* bp3 = next_bp(bp2);
        mma_step<BF16, TM, TN, 0, OPF>(acc, a2, bp2, bp3, RS, q0, q1);
        fetch(a2, s + 6);
        // past the last step there is nothing to read ahead: re-read this one (harmless)
        const char* bp4 = s + 4 < S ? next_bp(bp3) : bp3;
        mma_step<BF16, TM, TN, P1, OPF>(acc, a3, bp3, bp4, RS, q0, q1);
        fetch(a3, s + 7);
        bp = bp4;
    }
    if (s < S) {
        const char* bp1 = s + 1 < S ? next_bp(bp) : bp;
        mma_step<BF16, TM, TN, 0, OPF>(acc, a0, bp, bp1, RS, q0, q1);
        if (s + 1 < S) {
            const char* bp2 = s + 2 < S ? next_bp(bp1) : bp1;
            mma_step<BF16, TM, TN, P1, OPF>(acc, a1, bp1, bp2, RS, q0, q1);
            if (s + 2 < S) mma_step<BF16, TM, TN, 0, OPF>(acc, a2, bp2, bp2, RS, q0, q1);
        }
    }
    }
}

template <int OPF = OP_BF16>
__device__ __forceinline__ void store_out(const ConvOut& o, int b, long long e, long long valid, f32x4 v,
                                          const void* resid_b, float div) {
    if (e < 0 || e >= valid) return;
    const long long eb = (long long)b * o.batch_stride + e;
    if (resid_b) v += ld4(resid_b, e, o.f16);
    if (o.mode == ACC_ADD) {
        v += ld4(o.ptr, eb, o.f16);
    } else if (o.mode == ACC_ADD_DIV) {
        v += ld4(o.ptr, eb, o.f16);
        v[0] = v[0] / div;
        v[1] = v[1] / div;
        v[2] = v[2] / div;
        v[3] = v[3] / div;
    }
    st4<OPF>(o.ptr, eb, v, o.f16);
    if (o.copy_bf16) st4<OPF>(o.copy_bf16, eb, v, 2);
}

// 4-wave tiles with 64 accumulator registers fit 3 waves per SIMD (<= 168 VGPRs; the allocator lands just above
// that on its own): a third resident block per CU is what the HBM-bound low-channel layers need (bytes in flight).
template <bool BF16, int TM, int TN, int WM, int WN, int EPI, bool X3 = false, int OPF = OP_BF16>
__global__ __launch_bounds__(WM* WN * 64) __attribute__((amdgpu_waves_per_eu(TM * TN <= 16 && WM * WN == 4 ? 3 : 2)))
void conv_cl_kernel(const ConvParams p) {
    static_assert(BF16 || !X3, "split operands are bf16 operands");
    static_assert(OPF == OP_BF16 || (BF16 && !X3), "float16 operands: the 16-bit path, unsplit");
    constexpr int MT = 16 * TM * WM, NT = 16 * TN * WN, NTHREADS = WM * WN * 64;
    constexpr int ES = BF16 && !X3 ? 2 : 4;  // bytes of LDS per staged element (X3: a bf16 pair)
    constexpr int CH = BF16 ? 32 : 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int b = blockIdx.z;
    // Workgroups go to the 8 XCDs round-robin by linear index, and each XCD has its own L2.  With the natural
    // (row tile, M block) grid the M blocks of one row tile land on different XCDs and every one of them pulls the
    // tile's activation rows from HBM / Infinity Cache again.  Remapped, XCD x runs row tiles x, x+8, ... and for each
    // of them its ny M-blocks in consecutive slots, so the rows are fetched once per tile and re-read from that L2.
    int bx = blockIdx.x, by = blockIdx.y;
    if (p.xcd_ny > 0) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        by = slot % p.xcd_ny;
        bx = (slot / p.xcd_ny) * 8 + xcd;
        if (bx >= p.nx) return;
    }
    const int n0 = bx * NT;
    const int len = seq_frames(p.lengths, b, p.Tmax);
    const int n_ext = len * p.n_rows_per_frame + p.n_extra;
    if (n0 >= n_ext) return;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int lr = lane & 15, lg = lane >> 4;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int mtile0 = by * (MT / 16) + wm * TM;
    const long long mtile_stride = (long long)p.steps_total * 64;
    const u32x4* wlane = (const u32x4*)p.w + (long long)mtile0 * mtile_stride + lane;

    int step_base = 0;
    for (int ps = 0; ps < p.npass; ++ps) {
        const ConvPass& cp = p.pass[ps];
        const int rows = NT + (cp.ktaps - 1) * cp.dil;
        const int RS = cp.cb * ES + kLdsRowPad;
        if (ps > 0) __syncthreads();
        const long long src_off = (long long)b * cp.batch_stride;
        const void* src_b = cp.src_f16 ? (const void*)((const _Float16*)cp.src + src_off)
                                       : (const void*)((const float*)cp.src + src_off);
        if (BF16 && !X3 && cp.src_f16 == 2)
            stage_tile_raw<NTHREADS>(smem, cp, src_b, n0 - cp.pad_left, rows, len * cp.rows_per_frame, RS);
        else
            stage_tile<BF16, NTHREADS, X3, OPF>(smem, cp, src_b, n0 - cp.pad_left, rows, len * cp.rows_per_frame, RS);
        __syncthreads();
        const int nkc = cp.cb / CH;
        const char* lds_b = smem + (wn * TN * 16 + lr) * RS + lg * 16;
        mma_pass<BF16, TM, TN, 4, X3, OPF>(acc, lds_b, RS, cp.ktaps, cp.dil, nkc, wlane + (long long)step_base * 64,
                                           mtile_stride);
        step_base += cp.ktaps * nkc * (X3 ? 3 : 1);
    }

    // ---- epilogue: lane holds 4 consecutive output channels of one time step per (tm, tn)
    const void* resid_b = nullptr;
    if (p.resid) {
        const long long ro = (long long)b * p.resid_batch_stride;
        resid_b = p.out[0].f16 ? (const void*)((const _Float16*)p.resid + ro) : (const void*)((const float*)p.resid + ro);
    }
    if constexpr (BF16) {
        if (p.epi_lds) {
            // The accumulator layout gives every lane 4 channels of one row: stored directly, a wave
            // instruction touches 16 rows with 8 bytes each.  Instead the block's [NT][MO] result tile goes
            // to LDS as float16 and leaves with 16 bytes per lane along m, which is contiguous in memory for
            // a Conv1d (m = channel) and for the phase-major transposed conv alike (m = phase*Cout + co).
            // MO = MT, or MT/2 for the gate epilogue (one output channel per (tanh, sigmoid) row pair).
            constexpr int MO = EPI == EPI_GATE ? MT / 2 : MT;
            constexpr int TRS = MO * 2 + 16;
            __syncthreads();  // every wave is done reading the operand tiles
            if constexpr (EPI == EPI_LINEAR) {
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) {
                    const int ml = (wm * TM + tm) * 16 + lg * 4;
                    const f32x4 bias = *(const f32x4*)(p.bias + by * MT + ml);
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) {
                        const int nl = (wn * TN + tn) * 16 + lr;
                        f32x4 v = acc[tm][tn] + bias;
                        if (p.act == 1) v = f32x4{tanhf(v[0]), tanhf(v[1]), tanhf(v[2]), tanhf(v[3])};
                        if (p.out[0].f16 == 2) {  // operand-format output (operand-only tensor): the tile holds it as such
                            *(unsigned long long*)(smem + nl * TRS + ml * 2) =
                                ((unsigned long long)pack_op<OPF>(v[2], v[3]) << 32) | pack_op<OPF>(v[0], v[1]);
                        } else {
                            f16x4 h;
#pragma unroll
                            for (int i = 0; i < 4; ++i) h[i] = (_Float16)fminf(fmaxf(v[i], -65504.f), 65504.f);
                            *(f16x4*)(smem + nl * TRS + ml * 2) = h;
                        }
                    }
                }
            } else {
#pragma unroll
                for (int tp = 0; tp < TM / 2; ++tp) {
                    const int mt = (mtile0 + 2 * tp) * 16 + lg * 4;
                    const f32x4 bt = *(const f32x4*)(p.bias + mt);
                    const f32x4 bs = *(const f32x4*)(p.bias + mt + 16);
                    const int cl = ((wm * TM) / 2 + tp) * 16 + lg * 4;
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) {
                        const int nl = (wn * TN + tn) * 16 + lr;
                        const f32x4 t = acc[2 * tp][tn] + bt;
                        const f32x4 sg = acc[2 * tp + 1][tn] + bs;
                        f32x4 gv;
#pragma unroll
                        for (int j = 0; j < 4; ++j) gv[j] = gate_act<BF16>(t[j], sg[j]);
                        if (p.out[0].f16 == 2) {
                            *(unsigned long long*)(smem + nl * TRS + cl * 2) =
                                ((unsigned long long)pack_op<OPF>(gv[2], gv[3]) << 32) | pack_op<OPF>(gv[0], gv[1]);
                        } else {
                            f16x4 h;
#pragma unroll
                            for (int j = 0; j < 4; ++j) h[j] = (_Float16)gv[j];
                            *(f16x4*)(smem + nl * TRS + cl * 2) = h;
                        }
                    }
                }
            }
            __syncthreads();
            // a block's M range lies on one side of the split (launch_layer checks split % MT == 0)
            int mbase = by * MO;
            const int oi = (EPI == EPI_LINEAR && mbase >= p.split) ? 1 : 0;
            if (oi) mbase -= p.split;
            const ConvOut& o = p.out[oi];
            const long long valid = (long long)len * o.valid_per_frame;
            const void* rb = oi ? nullptr : resid_b;
            constexpr int UPR = MO / 8, TOTAL = NT * UPR, U = 4;
            for (int base = threadIdx.x; base < TOTAL; base += NTHREADS * U) {
                f32x4 r0[U][2], r1[U][2];
                long long ek[U];
#pragma unroll
                for (int k = 0; k < U; ++k) {
                    const int idx = base + k * NTHREADS;
                    const int nl = idx / UPR, u = idx - nl * UPR;
                    const int n = n0 + nl;
                    const long long e = (long long)n * o.row_stride + mbase + u * 8 + o.off;
                    ek[k] = (idx < TOTAL && n < n_ext && e >= 0 && e < valid) ? e : -1;
                    if (ek[k] >= 0) {
                        if (rb) {
                            r0[k][0] = ld4(rb, e, o.f16);
                            r0[k][1] = ld4(rb, e + 4, o.f16);
                        }
                        if (o.mode != ACC_STORE) {
                            const long long eb = (long long)b * o.batch_stride + e;
                            r1[k][0] = ld4(o.ptr, eb, o.f16);
                            r1[k][1] = ld4(o.ptr, eb + 4, o.f16);
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < U; ++k) {
                    if (ek[k] < 0) continue;
                    const int idx = base + k * NTHREADS;
                    const int nl = idx / UPR, u = idx - nl * UPR;
                    const u32x4 raw = *(const u32x4*)(smem + nl * TRS + u * 16);
                    const long long eb = (long long)b * o.batch_stride + ek[k];
                    if (o.f16 && !rb && o.mode == ACC_STORE) {
                        *(u32x4*)((_Float16*)o.ptr + eb) = raw;
                        continue;
                    }
                    const f16x8 a = __builtin_bit_cast(f16x8, raw);
                    f32x4 y[2] = {f32x4{(float)a[0], (float)a[1], (float)a[2], (float)a[3]},
                                  f32x4{(float)a[4], (float)a[5], (float)a[6], (float)a[7]}};
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        if (rb) y[q] += r0[k][q];
                        if (o.mode != ACC_STORE) y[q] += r1[k][q];
                        if (o.mode == ACC_ADD_DIV) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) y[q][i] = y[q][i] / p.div;
                        }
                    }
                    if (o.f16) {
                        f16x8 hh;
#pragma unroll
                        for (int i = 0; i < 8; ++i)
                            hh[i] = (_Float16)fminf(fmaxf(y[i >> 2][i & 3], -65504.f), 65504.f);
                        *(f16x8*)((_Float16*)o.ptr + eb) = hh;
                    } else {
                        *(f32x4*)((float*)o.ptr + eb) = y[0];
                        *(f32x4*)((float*)o.ptr + eb + 4) = y[1];
                    }
                    if (o.copy_bf16)
                        *(u32x4*)((unsigned short*)o.copy_bf16 + eb) =
                            u32x4{pack_op<OPF>(y[0][0], y[0][1]), pack_op<OPF>(y[0][2], y[0][3]),
                                  pack_op<OPF>(y[1][0], y[1][1]), pack_op<OPF>(y[1][2], y[1][3])};
                }
            }
            return;
        }
    }
    if constexpr (EPI == EPI_LINEAR) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            int m = (mtile0 + tm) * 16 + lg * 4;
            const f32x4 bias = *(const f32x4*)(p.bias + m);
            const int oi = m >= p.split ? 1 : 0;
            const ConvOut& o = p.out[oi];
            if (oi) m -= p.split;
            const long long valid = (long long)len * o.valid_per_frame;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int n = n0 + (wn * TN + tn) * 16 + lr;
                if (n >= n_ext) continue;
                const long long e = (long long)n * o.row_stride + m + o.off;
                f32x4 v = acc[tm][tn] + bias;
                if (p.act == 1) v = f32x4{tanhf(v[0]), tanhf(v[1]), tanhf(v[2]), tanhf(v[3])};
                store_out<OPF>(o, b, e, valid, v, oi ? nullptr : resid_b, p.div);
            }
        }
    } else {
        // gate (waveglow/glow.py:33-40): M tiles come in (tanh, sigmoid) pairs of the same 16 channels
        static_assert(EPI != EPI_GATE || TM % 2 == 0, "gate epilogue needs tile pairs");
        const ConvOut& o = p.out[0];
        const long long valid = (long long)len * o.valid_per_frame;
#pragma unroll
        for (int tp = 0; tp < TM / 2; ++tp) {
            const int mt = (mtile0 + 2 * tp) * 16 + lg * 4;
            const f32x4 bt = *(const f32x4*)(p.bias + mt);
            const f32x4 bs = *(const f32x4*)(p.bias + mt + 16);
            const int ch = ((mtile0 >> 1) + tp) * 16 + lg * 4;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int n = n0 + (wn * TN + tn) * 16 + lr;
                if (n >= n_ext) continue;
                const f32x4 t = acc[2 * tp][tn] + bt;
                const f32x4 s = acc[2 * tp + 1][tn] + bs;
                f32x4 g;
#pragma unroll
                for (int j = 0; j < 4; ++j) g[j] = gate_act<BF16>(t[j], s[j]);
                const long long e = (long long)n * o.row_stride + ch + o.off;
                store_out<OPF>(o, b, e, valid, g, nullptr, 1.f);
            }
        }
    }
}

}  // namespace mv
