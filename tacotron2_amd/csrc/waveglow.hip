// waveglow.hip -- WaveGlow.infer (waveglow/glow.py:251-295 of the reference) on conv_cl_kernel.
//
// Data in HBM (channels-last, L = 256*T/n_group time steps per sequence):
//   SPECT [B][L][n_mel*n_group]  upsampled + regrouped mel (glow.py:252-260); the transposed conv
//                                is a 4-tap polyphase conv whose M order is permuted so that its
//                                natural output [frame][phase*80+mel] IS [l][mel*n_group+g]
//   AUD   [B][L][n_group]        the flow state, right-aligned: the C_k live channels occupy
//                                [n_group-C_k, n_group); prepending early-output noise (glow.py:287-292)
//                                just lowers the start.  After flow 0 this buffer is the waveform
//                                (sample = l*n_group + c, glow.py:294).
//   H, SKIP, ACTS [B][L][C]      WN state (glow.py:153-175)
// fp32 mode: everything float32.  Per flow: start (VALU) -> n_layers x { in_layer + cond slice + gate (MFMA,
// EPI_GATE), res_skip (MFMA, split epilogue: H += res, SKIP += skip) } -> tail (VALU: end 1x1, affine coupling
// inverse, W^-1 1x1, early-output noise).  The cond_layer (glow.py:159) is never materialised: each layer's
// 2C-slice of it is a second 1-tap input of that layer's GEMM.
// bf16 mode: tensors that are only ever MFMA operands (SPECT, ACTS, and a copy Hb of the float32 residual stream H
// kept by the kernels that update it) are stored as bfloat16 and staged by plain copies; the skip halves of the
// res_skip layers are folded with the end conv into one [2*n_half][C] matrix per layer (wg_skipfold_kernel, fp32
// VALU), so the MFMA layer keeps its residual half and there is no skip tensor: see waveglow_build.
#include <cmath>

#include "gate_gemm.h"
#include "handle.h"

namespace mv {

struct WgFlow {
    int Ck = 0, n_half = 0;
    float* start_w = nullptr;  // device [C][n_half]
    float* start_b = nullptr;  // device [C]
    float* end_w = nullptr;    // device [2*n_half][Cp]
    float* end_b = nullptr;    // device [2*n_half]
    float* winv = nullptr;     // device [Ck][Ck]
    std::vector<Layer> in_layers, res_skip;
    // bf16 mode ("folded skip path", see waveglow_build): the residual half of res_skip alone, and end o skip_i as one
    // [2*n_half][Cp] matrix per layer
    std::vector<Layer> res_only;
    float* fold_w = nullptr;     // device [n_layers][2*n_half][Cp]
    float* fold_b = nullptr;     // device [2*n_half]: end.bias + end.weight * (sum of the skip biases)
    // bf16 mode: the in_layers' weights once more as plain [2C][K] bfloat16 matrices for gate_gemm_kernel
    std::vector<unsigned short*> gg_w;
    std::vector<unsigned short*> rg_w;  // the residual halves of res_skip as [C][Cp] bfloat16 for res_gemm_kernel
    unsigned short* gg_fold = nullptr;  // device [n_layers][2 (hi, lo)][16][Cp] bfloat16: fold_w split for the MFMA
};

struct WaveglowHandle : mv_handle_s {
    mv_waveglow_cfg cfg;
    int C = 0, Cp = 0, ncond = 0, melp = 0, n_rem = 0;
    bool fold = false;  // folded skip path (WgFlow)
    bool gg = false;    // gate_gemm_kernel weights were built (bf16 operand tensors, shapes it supports)
    Layer up;
    std::vector<WgFlow> flows;
    double macs = 0;

    ~WaveglowHandle() override {
        free_layer(&up);
        for (auto& f : flows) {
            for (auto& l : f.in_layers) free_layer(&l);
            for (auto& l : f.res_skip) free_layer(&l);
            for (auto& l : f.res_only) free_layer(&l);
            for (unsigned short* p : f.gg_w)
                if (p) (void)hipFree(p);
            for (unsigned short* p : f.rg_w)
                if (p) (void)hipFree(p);
            if (f.fold_w) (void)hipFree(f.fold_w);
            if (f.gg_fold) (void)hipFree(f.gg_fold);
            if (f.fold_b) (void)hipFree(f.fold_b);
            for (float* p : {f.start_w, f.start_b, f.end_w, f.end_b, f.winv})
                if (p) (void)hipFree(p);
        }
    }
    int hop() const override { return 256; }
    double macs_per_frame() const override { return macs; }
    static size_t al(size_t bytes) { return (bytes + 255) / 256 * 256; }
    // one zero row as long as the longer of an H row and a cond row (+ one 128-byte K chunk): gate_gemm_kernel adds
    // the K chunk's byte offset to the base it shares with the real rows
    size_t zero_bytes() const { return al((size_t)2 * std::max(Cp, ncond) + 128); }
    size_t workspace_bytes(int B, int Tmax) const override {
        const size_t L = (size_t)Tmax * 256 / cfg.n_group;
        const size_t bl = (size_t)B * L;
        // (folded skip path: every layer's gated activations of a flow are kept, as bfloat16, for the one fold launch
        // at its end -- n_layers - 1 slots beyond the ACTS buffer)
        // (+ the zero row gate_gemm_kernel reads outside a sequence, see waveglow_infer)
        return al((size_t)B * Tmax * melp * 4) + al(bl * ncond * 4) + 3 * al(bl * Cp * 4) + al(bl * cfg.n_group * 4) +
               (fold ? (size_t)(cfg.n_layers - 1) * al(bl * Cp * 2) : 0) + zero_bytes();
    }
};

// AUD[b][l][off + c] = sigma * noise[b][c][l]
__global__ __launch_bounds__(256) void wg_noise_kernel(const float* __restrict__ noise, int nc, int L, float sigma,
                                                       float* __restrict__ aud, int ng, int off) {
    const int b = blockIdx.y;
    const int l = blockIdx.x * 256 + threadIdx.x;
    if (l >= L) return;
    for (int c = 0; c < nc; ++c) aud[((long long)b * L + l) * ng + off + c] = sigma * noise[((long long)b * nc + c) * L + l];
}

// glow.py:155: H = start(audio_0) (1x1, n_half -> C)
constexpr int kStartRowsPerThread = 8;
constexpr int kGgMinBlocks = 192;  // 256 x 256 tiles of a gated conv below which conv_cl_kernel's smaller tiles fill the chip better
__global__ __launch_bounds__(256) void wg_start_kernel(const float* __restrict__ aud, int ng, int off, int n_half,
                                                       const float* __restrict__ w, const float* __restrict__ bias,
                                                       int C, int Cp, long long rows, void* __restrict__ H, int h16,
                                                       unsigned short* __restrict__ Hb, int Hs) {
    // a thread owns 4 channels (its weights and bias stay in registers) and walks the block's rows
    const int q = Cp / 4;  // float4 units per row; the host launches with 256 % q == 0
    const int c0 = (int)(threadIdx.x % q) * 4;
    float wr[4][4], br[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + i;
        br[i] = c < C ? bias[c] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) wr[i][j] = (c < C && j < n_half) ? w[c * n_half + j] : 0.f;
    }
    const int rpb = 256 / q * kStartRowsPerThread;  // rows per block
    for (int it = 0; it < kStartRowsPerThread; ++it) {
        const long long r = (long long)blockIdx.x * rpb + it * (256 / q) + threadIdx.x / q;
        if (r >= rows) return;
        float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < n_half) a[j] = aud[r * ng + off + j];
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            v[i] = (wr[i][0] * a[0] + wr[i][1] * a[1] + wr[i][2] * a[2] + wr[i][3] * a[3]) + br[i];
        st4(H, r * Hs + c0, v, h16);
        if (Hb) st4(Hb, r * Hs + c0, v, 2);  // the MFMA-operand view of H (see waveglow_infer)
    }
}

// generic shapes (Cp / 4 does not divide 256, or n_half > 4)
__global__ __launch_bounds__(256) void wg_start_generic_kernel(const float* __restrict__ aud, int ng, int off, int n_half,
                                                               const float* __restrict__ w,
                                                               const float* __restrict__ bias, int C, int Cp,
                                                               long long rows, void* __restrict__ H, int h16,
                                                               unsigned short* __restrict__ Hb, int Hs) {
    const int q = Cp / 4;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * q) return;
    const long long r = idx / q;
    const int c0 = (int)(idx - r * q) * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    float a[4];
    for (int j = 0; j < n_half; ++j) a[j] = aud[r * ng + off + j];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + i;
        if (c < C) {
            float s = 0.f;
            for (int j = 0; j < n_half; ++j) s += w[c * n_half + j] * a[j];
            v[i] = s + bias[c];
        }
    }
    st4(H, r * Hs + c0, v, h16);
    if (Hb) st4(Hb, r * Hs + c0, v, 2);  // the MFMA-operand view of H (see waveglow_infer)
}

// Folded skip path: end(sum_i skip_i(acts_i)) = sum_i (W_end W_skip_i) acts_i + const, so layer i adds
// out8[row][j] (+)= sum_c fold_w[j][c] * acts[row][c] (2*n_half <= 8 outputs per row) instead of a 512-channel skip
// tensor that is read-modify-written by every layer and only ever seen by the end conv.  16 lanes per row, each
// lane 16-byte units lane + 16q of the row; the 16 partial sums meet in four shuffles.
__global__ __launch_bounds__(256) void wg_skipfold_kernel(const void* __restrict__ acts0,
                                                          const void* __restrict__ acts_rest, long long slot_elems,
                                                          int nlayers, int acts_dt, int Cp,
                                                          const float* __restrict__ fold_w, int nout, long long rows,
                                                          float* __restrict__ out8, int accumulate) {
    // layer i reads acts0 (i = 0) or acts_rest + (i-1)*slot_elems and the i-th [nout][Cp] matrix of fold_w: with
    // every layer's activations kept (bf16 mode), one launch per flow does the work of n_layers
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ws = (float*)smem;  // [nout][Cp]
    const int l16 = threadIdx.x & 15;
    const long long row = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool live = row < rows;
    const int esz = acts_dt == 0 ? 4 : 2;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int layer = 0; layer < nlayers; ++layer) {
        if (layer > 0) __syncthreads();  // every wave is done with the previous layer's matrix
        const float* wl = fold_w + (long long)layer * nout * Cp;
        for (int i = threadIdx.x; i < nout * Cp; i += 256) ws[i] = wl[i];
        __syncthreads();
        const void* acts = layer == 0 ? acts0 : (const void*)((const char*)acts_rest + (layer - 1) * slot_elems * esz);
        if (live) {
            for (int u = l16; u < Cp / 8; u += 16) {
                f32x4 a, b;
                if (acts_dt == 2) {
                    const u32x4 raw = *(const u32x4*)((const unsigned short*)acts + row * Cp + u * 8);
                    a = f32x4{__uint_as_float(raw[0] << 16), __uint_as_float(raw[0] & 0xffff0000u),
                              __uint_as_float(raw[1] << 16), __uint_as_float(raw[1] & 0xffff0000u)};
                    b = f32x4{__uint_as_float(raw[2] << 16), __uint_as_float(raw[2] & 0xffff0000u),
                              __uint_as_float(raw[3] << 16), __uint_as_float(raw[3] & 0xffff0000u)};
                } else {
                    a = ld4(acts, row * Cp + u * 8, acts_dt);
                    b = ld4(acts, row * Cp + u * 8 + 4, acts_dt);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (j < nout) {
                        const f32x4 q0 = *(const f32x4*)(ws + j * Cp + u * 8), q1 = *(const f32x4*)(ws + j * Cp + u * 8 + 4);
                        acc[j] += a[0] * q0[0] + a[1] * q0[1] + a[2] * q0[2] + a[3] * q0[3] + b[0] * q1[0] +
                                  b[1] * q1[1] + b[2] * q1[2] + b[3] * q1[3];
                    }
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) acc[j] += __shfl_xor(acc[j], o);
    }
    if (!live || l16 >= nout) return;
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (j == l16) v = acc[j];
    float* o = out8 + row * 8 + l16;
    *o = accumulate ? *o + v : v;
}

// st4 (conv_mfma.h) with non-temporal stores, for results that the storing launch does not read again
__device__ __forceinline__ void st4_nt(void* base, long long e, const f32x4& v, int f16) {
    if (f16 == 2) {
        __builtin_nontemporal_store(((unsigned long long)pack_bf16(v[2], v[3]) << 32) | pack_bf16(v[0], v[1]),
                                    (unsigned long long*)((unsigned short*)base + e));
    } else if (f16) {
        f16x4 h;
#pragma unroll
        for (int i = 0; i < 4; ++i) h[i] = (_Float16)fminf(fmaxf(v[i], -65504.f), 65504.f);
        __builtin_nontemporal_store(h, (f16x4*)((_Float16*)base + e));
    } else {
        __builtin_nontemporal_store(v, (f32x4*)((float*)base + e));
    }
}

// The start conv of the flow that runs next, fused behind wg_tail_kernel (a row's new state is all it reads).
struct WgNextStart {
    const float* w;     // [C][n_half], nullptr = not fused
    const float* bias;  // [C]
    int n_half, C;
    void* H;            // residual stream (float16 / float32 by h16), row stride Hs
    int h16;
    unsigned short* Hb;  // its bfloat16 operand copy, or nullptr
    int Hs;
};

// glow.py:175 (end), 280-285 (coupling inverse, W^-1), 287-292 (early noise).  One wave per 16 steps:
// lane = 16*part + step; the 4 parts split the channels of SKIP and are summed with two shuffles.
__global__ __launch_bounds__(256) void wg_tail_kernel(const void* __restrict__ skip, int skip16, int Cp, const float* __restrict__ end_w,
                                                      const float* __restrict__ end_b, int n_half, int Ck,
                                                      const float* __restrict__ winv, float* __restrict__ aud, int ng,
                                                      const int32_t* __restrict__ lengths, int Tmax, int L,
                                                      int steps_per_frame, const float* __restrict__ noise, int n_early,
                                                      float sigma, const float* __restrict__ folded, int fold_parts,
                                                      long long fold_part_stride, WgNextStart nx) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ws = (float*)smem;  // [2*n_half][Cp]
    float* st = ws + 2 * n_half * Cp;  // [64][8]: the block's rows of the new flow state, for the next flow's start conv
    const int nout = 2 * n_half;
    // the next flow's start conv (nx.w != nullptr): a thread owns 4 channels, as in wg_start_kernel; its weights are
    // requested first and arrive under the coupling arithmetic
    const int sq = Cp / 4;
    const int sc0 = (int)(threadIdx.x % sq) * 4;
    float swr[4][4], sbr[4];
    if (nx.w) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = sc0 + i;
            sbr[i] = c < nx.C ? nx.bias[c] : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) swr[i][j] = (c < nx.C && j < nx.n_half) ? nx.w[c * nx.n_half + j] : 0.f;
        }
    }
    if (!folded)
        for (int i = threadIdx.x; i < nout * Cp; i += 256) ws[i] = end_w[i];
    __syncthreads();
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int part = lane >> 4;
    const int l = (blockIdx.x * 4 + wave) * 16 + (lane & 15);
    const int len = seq_frames(lengths, b, Tmax) * steps_per_frame;
    const bool live = l < L;
    const bool valid = live && l < len;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    if (valid && folded) {  // the end conv's outputs arrive ready-made (wg_skipfold_kernel); end_b = the folded bias
        // (gate_gemm_kernel leaves one partial sum per M block: the four lane groups take them in turn and the
        // shuffles below add them up)
        for (int q = part; q < fold_parts; q += 4) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < nout) acc[j] += folded[q * fold_part_stride + ((long long)b * L + l) * 8 + j];
        }
    } else if (valid) {
        const long long se = ((long long)b * L + l) * Cp;
        for (int c4 = part; c4 < Cp / 4; c4 += 4) {
            const f32x4 v = ld4(skip, se + c4 * 4, skip16);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (j < nout) {
                    const f32x4 q = *(const f32x4*)(ws + j * Cp + c4 * 4);
                    acc[j] += v[0] * q[0] + v[1] * q[1] + v[2] * q[2] + v[3] * q[3];
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        acc[j] += __shfl_xor(acc[j], 16);
        acc[j] += __shfl_xor(acc[j], 32);
    }
    if (!nx.w && (part != 0 || !live)) return;
    float* srow = st + (wave * 16 + (lane & 15)) * 8;  // (the host fuses the start conv only when ng <= 8)
    if (part == 0 && nx.w) {
#pragma unroll
        for (int c = 0; c < 8; ++c) srow[c] = 0.f;  // rows past the sequence's end: state 0, H = bias
    }
    if (part == 0 && live) {
        float* ar = aud + ((long long)b * L + l) * ng;
        const int off = ng - Ck;
        if (!valid) {
            for (int c = 0; c < ng; ++c) ar[c] = 0.f;
        } else {
            float a[8], o[8];
            for (int c = 0; c < Ck; ++c) a[c] = ar[off + c];
            for (int c = 0; c < n_half; ++c) {
                const float bb = acc[c] + end_b[c];
                const float s = acc[n_half + c] + end_b[n_half + c];
                a[n_half + c] = (a[n_half + c] - bb) / expf(s);
            }
            for (int r = 0; r < Ck; ++r) {
                float s = 0.f;
                for (int c = 0; c < Ck; ++c) s += winv[r * Ck + c] * a[c];
                o[r] = s;
            }
            for (int c = 0; c < Ck; ++c) {
                ar[off + c] = o[c];
                if (nx.w) srow[n_early + c] = o[c];
            }
            for (int j = 0; j < n_early; ++j) {
                const float z = sigma * noise[((long long)b * n_early + j) * L + l];
                ar[off - n_early + j] = z;
                if (nx.w) srow[j] = z;
            }
        }
    }
    if (!nx.w) return;
    // glow.py:155 of the next flow: H = start(audio_0), audio_0 = the first n_half channels of the state just written
    __syncthreads();
    for (int it = (int)threadIdx.x / sq; it < 64; it += 256 / sq) {
        const int l2 = blockIdx.x * 64 + it;
        if (l2 >= L) break;
        const float* a = st + it * 8;
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (swr[i][0] * a[0] + swr[i][1] * a[1] + swr[i][2] * a[2] + swr[i][3] * a[3]) + sbr[i];
        const long long r = (long long)b * L + l2;
        st4_nt(nx.H, r * nx.Hs + sc0, v, nx.h16);
        if (nx.Hb) st4_nt(nx.Hb, r * nx.Hs + sc0, v, 2);
    }
}

static inline unsigned short bf16_bits(float x) {  // round to nearest even
    uint32_t u;
    memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static inline float bf16_value(unsigned short h) {
    const uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// Gauss-Jordan with partial pivoting in double.  The reference's W.float().inverse() (glow.py:91) raises on a singular
// matrix; here a pivot below 1e-12 of the largest entry, or a non-finite result, is MV_EINVAL instead of inf / NaN
// weights that would silently turn every waveform into NaN.
static int invert_small(const float* A, int n, float* out) {
    double a[8 * 16];
    double amax = 0;
    for (int i = 0; i < n * n; ++i) amax = std::max(amax, (double)std::fabs(A[i]));
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            a[i * 2 * n + j] = A[i * n + j];
            a[i * 2 * n + n + j] = i == j ? 1.0 : 0.0;
        }
    for (int c = 0; c < n; ++c) {
        int p = c;
        for (int r = c + 1; r < n; ++r)
            if (std::fabs(a[r * 2 * n + c]) > std::fabs(a[p * 2 * n + c])) p = r;
        if (p != c)
            for (int j = 0; j < 2 * n; ++j) std::swap(a[c * 2 * n + j], a[p * 2 * n + j]);
        const double d = a[c * 2 * n + c];
        if (!(std::fabs(d) > 1e-12 * amax)) return MV_EINVAL;
        for (int j = 0; j < 2 * n; ++j) a[c * 2 * n + j] /= d;
        for (int r = 0; r < n; ++r) {
            if (r == c) continue;
            const double f = a[r * 2 * n + c];
            for (int j = 0; j < 2 * n; ++j) a[r * 2 * n + j] -= f * a[c * 2 * n + j];
        }
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            out[i * n + j] = (float)a[i * 2 * n + n + j];
            if (!std::isfinite(out[i * n + j])) return MV_EINVAL;
        }
    return MV_OK;
}

static int expect(const HostTensor* t, std::initializer_list<int64_t> shape, const char* name) {
    if (!t) return MV_ENOTFOUND;
    size_t i = 0;
    bool ok = t->shape.size() >= shape.size();
    for (int64_t d : shape) ok = ok && i < t->shape.size() && t->shape[i++] == d;
    for (; i < t->shape.size(); ++i) ok = ok && t->shape[i] == 1;
    if (!ok) {
        set_error("%s: unexpected shape", name);
        return MV_EINVAL;
    }
    return MV_OK;
}

static int waveglow_build(WaveglowHandle* h, WeightMap& W) {
    const mv_waveglow_cfg& c = h->cfg;
    const bool bf16 = h->precision == MV_PREC_BF16;
    const int prec = h->precision;  // MV_PREC_* as the layer builders take it
    const int C = c.n_channels, nl = c.n_layers, ks = c.kernel_size, ng = c.n_group, nm = c.n_mel_channels;
    h->C = C;
    h->Cp = round_up(C, 32);
    h->ncond = nm * ng;
    h->melp = round_up(nm, 32);
    if (C % 32 != 0 || h->ncond % 32 != 0 || ks % 2 != 1 || ng > 8 || ng % 2 != 0 || 256 % ng != 0) {
        set_error("waveglow: unsupported shape (n_channels=%d, n_group=%d, kernel_size=%d)", C, ng, ks);
        return MV_EUNSUPPORTED;
    }
    char name[160];
    int rc;
    double macs_step = 0;
    // the gated conv as a 256 x 256 GEMM needs bfloat16 operand tensors (io16) and whole 256-row M blocks
    h->gg = bf16 && h->opt.io16 && h->opt.wg_gemm != 0 && gate_gemm_supported(C, h->Cp, h->ncond, ks);
    {
        // glow.py:183-185,252-260: ConvTranspose1d(nm, nm, 1024, stride 256) + trim + regroup
        const int K = 1024, u = 256, J = 4;
        const HostTensor* w = W.get("upsample.weight");
        if ((rc = expect(w, {nm, nm, K}, "upsample.weight"))) return rc;
        const HostTensor* b = W.get("upsample.bias");
        if ((rc = expect(b, {nm}, "upsample.bias"))) return rc;
        const float* wd = w->data.data();
        const int per_row = nm * ng;  // channels of one SPECT row
        SrcSpec s;
        s.C = h->melp;
        s.ktaps = J;
        s.dil = 1;
        s.pad_left = J - 1;
        // m = (p / ng) * per_row + mel * ng + (p % ng), p = phase of the stride-256 transposed conv
        s.w = [=](int m, int tap, int ci) -> float {
            const int row = m / per_row, rem = m - row * per_row;
            const int mel = rem / ng, g = rem - mel * ng;
            const int p = row * ng + g;
            const int k = p + (J - 1 - tap) * u;
            return ci < nm ? wd[((size_t)ci * nm + mel) * K + k] : 0.f;
        };
        const int M = u * nm;
        std::vector<float> bias(M);
        for (int m = 0; m < M; ++m) bias[m] = b->data[(m % per_row) / ng];
        // always exact fp32: 0.08 % of the FLOPs, and it feeds every flow
        if ((rc = build_layer(&h->up, {s}, M, bias, MV_PREC_FP32, EPI_LINEAR))) return rc;
        h->up.alg_macs_per_n = (double)nm * nm * K;
    }
    h->flows.resize(c.n_flows);
    int n_half = ng / 2, n_rem = ng;
    for (int k = 0; k < c.n_flows; ++k) {
        if (k % c.n_early_every == 0 && k > 0) {
            n_half -= c.n_early_size / 2;
            n_rem -= c.n_early_size;
        }
        if (n_half < 1 || n_rem != 2 * n_half) {
            set_error("waveglow: flow %d has %d channels / n_half %d", k, n_rem, n_half);
            return MV_EUNSUPPORTED;
        }
        WgFlow& f = h->flows[k];
        f.Ck = n_rem;
        f.n_half = n_half;
        snprintf(name, sizeof name, "convinv.%d.conv.weight", k);
        const HostTensor* wi = W.get(name);
        if ((rc = expect(wi, {n_rem, n_rem}, name))) return rc;
        std::vector<float> inv((size_t)n_rem * n_rem);
        if (invert_small(wi->data.data(), n_rem, inv.data())) {
            set_error("%s is singular (or not finite): no inverse for the reverse 1x1 convolution", name);
            return MV_EINVAL;
        }
        if ((rc = upload(inv, &f.winv))) return rc;

        snprintf(name, sizeof name, "WN.%d.start.weight", k);
        const HostTensor* sw = W.get(name);
        if ((rc = expect(sw, {C, n_half}, name))) return rc;
        snprintf(name, sizeof name, "WN.%d.start.bias", k);
        const HostTensor* sbias = W.get(name);
        if ((rc = expect(sbias, {C}, name))) return rc;
        if ((rc = upload(sw->data, &f.start_w)) || (rc = upload(sbias->data, &f.start_b))) return rc;

        snprintf(name, sizeof name, "WN.%d.end.weight", k);
        const HostTensor* ew = W.get(name);
        if ((rc = expect(ew, {2 * n_half, C}, name))) return rc;
        snprintf(name, sizeof name, "WN.%d.end.bias", k);
        const HostTensor* eb = W.get(name);
        if ((rc = expect(eb, {2 * n_half}, name))) return rc;
        std::vector<float> ewp((size_t)2 * n_half * h->Cp, 0.f);
        for (int j = 0; j < 2 * n_half; ++j)
            for (int cc = 0; cc < C; ++cc) ewp[(size_t)j * h->Cp + cc] = ew->data[(size_t)j * C + cc];
        if ((rc = upload(ewp, &f.end_w)) || (rc = upload(eb->data, &f.end_b))) return rc;

        snprintf(name, sizeof name, "WN.%d.cond_layer.weight", k);
        const HostTensor* cw = W.get(name);
        if ((rc = expect(cw, {2 * C * nl, h->ncond}, name))) return rc;
        snprintf(name, sizeof name, "WN.%d.cond_layer.bias", k);
        const HostTensor* cb = W.get(name);
        if ((rc = expect(cb, {2 * C * nl}, name))) return rc;

        f.in_layers.resize(nl);
        f.res_skip.resize(nl);
        // Folded skip path (bf16 mode, MV_WG_FOLD): the skip halves of the res_skip layers feed nothing but the end
        // conv, and end(sum_i skip_i) is linear, so layer i contributes (W_end W_skip_i) acts_i -- a [2*n_half][C]
        // matrix applied on the VALU in fp32 -- and the MFMA layer keeps its residual half only.  Halves the res_skip
        // FLOPs and drops the skip tensor's read-modify-write; the values differ from the reference order of
        // operations by fp32 rounding only (the products are formed in double here).
        const bool fold = bf16 && h->opt.wg_fold != 0;
        h->fold = fold;
        std::vector<double> fold_bias(2 * n_half, 0.0);
        std::vector<float> fold_all(fold ? (size_t)nl * 2 * n_half * h->Cp : 0, 0.f);
        if (fold) {
            f.res_only.resize(nl);
            for (int j = 0; j < 2 * n_half; ++j) fold_bias[j] = eb->data[j];
        }
        for (int i = 0; i < nl; ++i) {
            snprintf(name, sizeof name, "WN.%d.in_layers.%d.weight", k, i);
            const HostTensor* iw = W.get(name);
            if ((rc = expect(iw, {2 * C, C, ks}, name))) return rc;
            snprintf(name, sizeof name, "WN.%d.in_layers.%d.bias", k, i);
            const HostTensor* ib = W.get(name);
            if ((rc = expect(ib, {2 * C}, name))) return rc;
            const float* iwd = iw->data.data();
            const float* cwd = cw->data.data() + (size_t)i * 2 * C * h->ncond;
            const float* cbd = cb->data.data() + (size_t)i * 2 * C;
            // M tiles in (tanh, sigmoid) pairs: tile 2j -> channels 16j.., tile 2j+1 -> C + 16j..
            auto orig = [=](int m) {
                const int tile = m >> 4, r = m & 15;
                return ((tile & 1) ? C : 0) + (tile >> 1) * 16 + r;
            };
            const int d = 1 << i;
            SrcSpec sh;
            sh.C = h->Cp;
            sh.ktaps = ks;
            sh.dil = d;
            sh.pad_left = d * (ks - 1) / 2;
            sh.w = [=](int m, int tap, int ci) -> float {
                return ci < C ? iwd[((size_t)orig(m) * C + ci) * ks + tap] : 0.f;
            };
            SrcSpec sc;
            sc.C = h->ncond;
            sc.ktaps = 1;
            sc.dil = 1;
            sc.pad_left = 0;
            const int ncond = h->ncond;
            sc.w = [=](int m, int, int ci) -> float { return cwd[(size_t)orig(m) * ncond + ci]; };
            std::vector<float> bias(2 * C);
            for (int m = 0; m < 2 * C; ++m) bias[m] = ib->data[orig(m)] + cbd[orig(m)];
            // bf16: 256 x 128 tiles (a weight fragment feeds 8 MFMAs instead of 4: the L1 weight stream paces
            // the 256 x 64 tile)
            const bool op16 = bf16 || h->precision == MV_PREC_F16;  // (float16 operands: the reference's split, layer kernels)
            const int wide = op16 && (2 * C) % 256 == 0 ? 1 : -1;
            if ((rc = build_layer(&f.in_layers[i], {sh, sc}, 2 * C, bias, prec, EPI_GATE, wide))) return rc;
            f.in_layers[i].alg_macs_per_n = (double)2 * C * C * ks + (double)2 * C * h->ncond;
            if (h->gg) {
                unsigned short* gw = nullptr;
                if ((rc = build_gate_gemm_weights({sh, sc}, 2 * C, &gw))) return rc;
                f.gg_w.push_back(gw);
            }

            const int rsC = i < nl - 1 ? 2 * C : C;
            snprintf(name, sizeof name, "WN.%d.res_skip_layers.%d.weight", k, i);
            const HostTensor* rw = W.get(name);
            if ((rc = expect(rw, {rsC, C}, name))) return rc;
            snprintf(name, sizeof name, "WN.%d.res_skip_layers.%d.bias", k, i);
            const HostTensor* rb = W.get(name);
            if ((rc = expect(rb, {rsC}, name))) return rc;
            const float* rwd = rw->data.data();
            SrcSpec sa;
            sa.C = h->Cp;
            sa.ktaps = 1;
            sa.dil = 1;
            sa.pad_left = 0;
            sa.w = [=](int m, int, int ci) -> float { return ci < C ? rwd[(size_t)m * C + ci] : 0.f; };
            const int rwide = op16 && rsC % 256 == 0 ? 1 : -1;
            if (!fold) {
                if ((rc = build_layer(&f.res_skip[i], {sa}, rsC, rb->data, prec, EPI_LINEAR, rwide))) return rc;
                f.res_skip[i].alg_macs_per_n = (double)rsC * C;
            } else {
                const int s0 = rsC - C;  // first skip row of res_skip_layers[i]
                float* fw = fold_all.data() + (size_t)i * 2 * n_half * h->Cp;
                for (int j = 0; j < 2 * n_half; ++j) {
                    const float* ej = ew->data.data() + (size_t)j * C;
                    for (int cc = 0; cc < C; ++cc) {
                        double acc = 0;
                        for (int m = 0; m < C; ++m) acc += (double)ej[m] * rwd[(size_t)(s0 + m) * C + cc];
                        fw[(size_t)j * h->Cp + cc] = (float)acc;
                    }
                    for (int m = 0; m < C; ++m) fold_bias[j] += (double)ej[m] * rb->data[s0 + m];
                }
                if (i < nl - 1) {
                    std::vector<float> rbias(rb->data.begin(), rb->data.begin() + C);
                    const int cfg = C % 256 == 0 ? 0 : -1;
                    if ((rc = build_layer(&f.res_only[i], {sa}, C, rbias, MV_PREC_BF16, EPI_LINEAR, cfg))) return rc;
                    f.res_only[i].alg_macs_per_n = (double)C * C;
                    if (h->gg) {
                        unsigned short* rw16 = nullptr;
                        if ((rc = build_gate_gemm_weights({sa}, C, &rw16))) return rc;
                        f.rg_w.push_back(rw16);
                    }
                }
            }
        }
        if (fold) {
            std::vector<float> fb(fold_bias.begin(), fold_bias.end());
            if ((rc = upload(fb, &f.fold_b)) || (rc = upload(fold_all, &f.fold_w))) return rc;
            if (h->gg) {  // the same matrices as (hi, lo) bfloat16 pairs, 16 rows each, for gate_gemm_kernel's epilogue
                std::vector<unsigned short> g16((size_t)nl * 2 * 16 * h->Cp, 0);
                for (int i = 0; i < nl; ++i)
                    for (int j = 0; j < 2 * n_half; ++j)
                        for (int cc = 0; cc < h->Cp; ++cc) {
                            const float w = fold_all[((size_t)i * 2 * n_half + j) * h->Cp + cc];
                            const unsigned short hi = bf16_bits(w);
                            g16[(((size_t)i * 2 + 0) * 16 + j) * h->Cp + cc] = hi;
                            g16[(((size_t)i * 2 + 1) * 16 + j) * h->Cp + cc] = bf16_bits(w - bf16_value(hi));
                        }
                if ((rc = upload(g16, &f.gg_fold))) return rc;
            }
        }
        // (executed work: with the folded skip path the res_skip layers cost C*C (+ 2*n_half*C on the VALU) per layer)
        macs_step += (double)n_half * C + (double)h->ncond * 2 * C * nl + (double)nl * 2 * C * C * ks +
                     (fold ? (double)(nl - 1) * C * C + (double)nl * 2 * n_half * C
                           : (double)(nl - 1) * 2 * C * C + (double)C * C + (double)C * 2 * n_half) +
                     (double)n_rem * n_rem;
    }
    h->n_rem = n_rem;
    h->macs = macs_step * (256 / ng) + (double)nm * nm * 1024;
    return MV_OK;
}

static int waveglow_infer(WaveglowHandle* h, const float* mel, const int32_t* lengths, int B, int Tmax,
                          const float* const* noise, int n_noise, float sigma, void* wav, int out_format,
                          void* workspace, size_t workspace_bytes, hipStream_t stream) {
    const mv_waveglow_cfg& c = h->cfg;
    const int ng = c.n_group, nl = c.n_layers, Cp = h->Cp;
    const int spf = 256 / ng;  // time steps per mel frame
    const int L = Tmax * spf;
    if (workspace_bytes < h->workspace_bytes(B, Tmax) || ((uintptr_t)workspace & 255) != 0) {
        set_error("workspace: need %zu bytes, 256-byte aligned (got %zu)", h->workspace_bytes(B, Tmax),
                  workspace_bytes);
        return MV_ENOMEM;
    }
    int n_early_flows = 0;
    for (int k = 0; k < c.n_flows; ++k)
        if (k % c.n_early_every == 0 && k > 0) ++n_early_flows;
    if (n_noise != 1 + n_early_flows) {
        set_error("mv_waveglow_infer: expected %d noise tensors, got %d", 1 + n_early_flows, n_noise);
        return MV_EINVAL;
    }
    for (int i = 0; i < n_noise; ++i)
        if (!noise[i]) {
            set_error("mv_waveglow_infer: noise[%d] is null", i);
            return MV_EINVAL;
        }
    int rc;
    Profiler* prof = h->prof.on ? &h->prof : nullptr;
    const size_t bl = (size_t)B * L;
    char* ws = (char*)workspace;
    float* X0 = (float*)ws;
    ws += WaveglowHandle::al((size_t)B * Tmax * h->melp * 4);
    void* SPECT = ws;
    ws += WaveglowHandle::al(bl * h->ncond * 4);
    void* H = ws;
    ws += WaveglowHandle::al(bl * Cp * 4);
    void* SKIP = ws;
    ws += WaveglowHandle::al(bl * Cp * 4);
    void* ACTS = ws;
    ws += WaveglowHandle::al(bl * Cp * 4);
    // bf16 mode with the folded skip path: the residual stream H is stored as float16 (11-bit mantissa; it is summed in
    // float32 inside the kernel that updates it) next to its bfloat16 operand copy Hb: -6 % on the residual conv, no
    // change in SNR (47.8 dB worst case over tests/test_gpu_bf16_margin.py).  Otherwise H / SKIP are float32.
    const int io16 = h->precision == MV_PREC_BF16 && h->opt.io16 && h->fold ? 1 : 0;
    // The gated activations and the upsampled mel are read as MFMA operands only: in bf16 mode they are stored as
    // bfloat16, rounded where they are produced instead of where they are staged -- the same values reach the
    // MFMAs, at half the bytes and with plain 16-byte copies into LDS.
    const int opd = h->precision == MV_PREC_BF16 && h->opt.io16 ? 2 : io16;
    // Likewise the residual stream H: it accumulates in float32, and every kernel that updates it also leaves a
    // bfloat16 copy (in the unused half of the ACTS buffer) for the next gated conv to stage by plain copies.
    unsigned short* Hb = opd == 2 ? (unsigned short*)((char*)ACTS + WaveglowHandle::al(bl * Cp * 2)) : nullptr;
    const int Hs = Cp;  // row stride of H / Hb (a padded stride, 64 unused channels per row against power-of-two spacing
                        // of gate_gemm_kernel's 128-byte pieces, measured no different: 594 vs 594 audio-s/s)
    // one fold launch per flow over all its layers' activations (needs them all kept: bf16 operand storage)
    const bool fold_batch = h->fold && opd == 2;
    const size_t slot = WaveglowHandle::al(bl * Cp * 2);
    // a row of zeros right behind ACTS: gate_gemm_kernel addresses the rows of SPECT and Hb and this zero row (what it
    // reads outside a sequence) as ONE scalar base + a 32-bit lane offset, so the row sits above both, within 4 GiB
    unsigned short* ZERO = (unsigned short*)ws;
    ws += h->zero_bytes();
    char* ACTS_REST = ws;  // slots 1 .. n_layers-1
    if (h->fold) ws += (size_t)(nl - 1) * slot;
    float* AUDW = (float*)ws;
    // float output: the flow state lives directly in the caller's wav buffer
    float* AUD = out_format == MV_OUT_F32 ? (float*)wav : AUDW;

    if ((rc = launch_to_channels_last(mel, B, c.n_mel_channels, Tmax, X0, h->melp, stream))) return rc;
    {
        SrcBind sb{X0, (long long)Tmax * h->melp, h->melp, 1, 1.0f};
        LaunchArgs a;
        a.lengths = lengths;
        a.B = B;
        a.Tmax = Tmax;
        a.n_rows_per_frame = 1;
        a.n_extra = 3;
        const long long per_frame = (long long)spf * h->ncond;  // = 256 * n_mel
        a.out[0] = ConvOut{SPECT, (long long)Tmax * per_frame, 0, per_frame, (int)per_frame, ACC_STORE, opd};
        if ((rc = launch_layer(h->up, &sb, 1, a, stream, prof))) return rc;
    }
    {
        dim3 grid((L + 255) / 256, B);
        hipLaunchKernelGGL(wg_noise_kernel, grid, dim3(256), 0, stream, noise[0], h->n_rem, L, sigma, AUD, ng,
                           ng - h->n_rem);
        MV_HIP(hipGetLastError());
    }
    int ni = 1;
    const long long bsC = (long long)L * Cp, bsH = (long long)L * Hs;
    // gate_gemm_kernel works on 256-row tiles, one block per CU: taken when the call has enough of them to fill
    // the chip (or whenever it can run, wg_gemm >= 2); shorter calls keep conv_cl_kernel's 128-row tiles
    const int gg_nx = (L + kGgTile - 1) / kGgTile;
    // (its per-lane row offsets are 32-bit: a sequence's rows x row bytes must stay below 4 GiB)
    const bool gg_fits = (long long)L * std::max(Hs, h->ncond) * 2 < (1ll << 32) &&
                         (long long)((char*)ZERO + h->zero_bytes() - (char*)SPECT) < (1ll << 32);
    const bool use_gg = h->gg && Hb && opd == 2 && gg_fits &&
                        (h->opt.wg_gemm >= 2 || (long long)B * gg_nx * (2 * h->C / kGgTile) >= kGgMinBlocks);
    if (use_gg) MV_HIP(hipMemsetAsync(ZERO, 0, h->zero_bytes(), stream));
    // ... and with it the folded skip path moves into its epilogue (one partial sum per M block, added by wg_tail_kernel)
    const bool gg_fold = use_gg && h->fold;
    const int fold_parts = gg_fold ? 2 * h->C / kGgTile : 1;
    // the residual conv's M block: 256 rows when the call has enough row tiles to fill the chip with those, else 128
    const int res_bm =
        h->C % 256 == 0 && (h->opt.wg_gemm == 3 || (long long)B * gg_nx * (h->C / 256) >= kGgMinBlocks) ? 256 : 128;
    bool started = false;  // flow k's start conv already ran behind flow k+1's tail
    for (int k = c.n_flows - 1; k >= 0; --k) {
        const WgFlow& f = h->flows[k];
        if (!started) {
            const long long rows = (long long)B * L;
            const int q = Cp / 4;
            if (q <= 256 && 256 % q == 0 && f.n_half <= 4) {
                const long long rpb = 256 / q * kStartRowsPerThread;
                hipLaunchKernelGGL(wg_start_kernel, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(256), 0, stream, AUD,
                                   ng, ng - f.Ck, f.n_half, f.start_w, f.start_b, h->C, Cp, rows, H, io16, Hb, Hs);
            } else {
                const long long total = rows * q;
                hipLaunchKernelGGL(wg_start_generic_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                                   AUD, ng, ng - f.Ck, f.n_half, f.start_w, f.start_b, h->C, Cp, rows, H, io16, Hb, Hs);
            }
            MV_HIP(hipGetLastError());
        }
        for (int i = 0; i < nl; ++i) {
            SrcBind sb[2] = {Hb ? SrcBind{Hb, bsH, Hs, spf, 1.0f, 2} : SrcBind{H, bsH, Hs, spf, 1.0f, io16}, {SPECT, (long long)L * h->ncond, h->ncond, spf, 1.0f, opd}};
            LaunchArgs a;
            a.lengths = lengths;
            a.B = B;
            a.Tmax = Tmax;
            a.n_rows_per_frame = spf;
            void* acts_i = fold_batch && !gg_fold && i > 0 ? (void*)(ACTS_REST + (size_t)(i - 1) * slot) : ACTS;
            a.out[0] = ConvOut{acts_i, bsC, 0, (long long)spf * Cp, Cp, ACC_STORE, opd};
            if (use_gg) {
                GateGemmParams g = {};
                g.w = f.gg_w[i];
                g.bias = f.in_layers[i].bias;
                g.h = Hb;
                g.cond = (const unsigned short*)SPECT;
                g.out = (unsigned short*)acts_i;
                g.zero = ZERO;
                g.lengths = lengths;
                g.Tmax = Tmax;
                g.spf = spf;
                g.L = L;
                g.Cp = Cp;
                g.h_stride = Hs;
                g.ncond = h->ncond;
                g.ktaps = c.kernel_size;
                g.dil = 1 << i;
                g.K = c.kernel_size * Cp + h->ncond;
                g.nx = gg_nx;
                g.ny = 2 * h->C / kGgTile;
                g.fold_w = gg_fold ? f.gg_fold + (size_t)i * 2 * 16 * Cp : nullptr;
                g.fold_out = (float*)SKIP;
                g.fold_part_stride = (long long)bl * 8;
                g.fold_nout = 2 * f.n_half;
                g.fold_accumulate = i > 0;
                if (prof) prof->begin(stream, "mv::gate_gemm_kernel(mv::GateGemmParams)", f.in_layers[i].alg_macs_per_n * spf);
                rc = launch_gate_gemm(g, B, stream);
                if (prof) prof->end(stream);
                if (rc) return rc;
            } else if ((rc = launch_layer(f.in_layers[i], sb, 2, a, stream, prof))) {
                return rc;
            }
            SrcBind sa{acts_i, bsC, Cp, spf, 1.0f, opd};
            if (h->fold) {
                const int nout = 2 * f.n_half;
                const long long rows = (long long)B * L;
                // (rows past a sequence's end hold stale activations: their sums are never read, wg_tail masks them)
                if (!gg_fold && (!fold_batch || i == nl - 1)) {
                    const int nfold = fold_batch ? nl : 1;
                    if (prof) prof->begin(stream, "mv::wg_skipfold_kernel(void const*, void const*, long long, int, int, int, float const*, int, long long, float*, int)", (double)nfold * nout * h->C * spf);
                    hipLaunchKernelGGL(wg_skipfold_kernel, dim3((unsigned)((rows + 15) / 16)), dim3(256),
                                       (size_t)nout * Cp * 4, stream, (const void*)ACTS, (const void*)ACTS_REST,
                                       (long long)(slot / 2), nfold, opd, Cp,
                                       f.fold_w + (fold_batch ? 0 : (size_t)i * nout * Cp), nout, rows, (float*)SKIP,
                                       fold_batch || i == 0 ? 0 : 1);
                    if (prof) prof->end(stream);
                    MV_HIP(hipGetLastError());
                }
                if (i < nl - 1 && gg_fold && io16) {
                    // the residual conv on the ring kernel as well (128 x 256 tiles): H (float16) += W acts + b
                    GateGemmParams g = {};
                    g.w = f.rg_w[i];
                    g.bias = f.res_only[i].bias;
                    g.h = (const unsigned short*)acts_i;
                    g.cond = (const unsigned short*)acts_i;
                    g.zero = ZERO;
                    g.lengths = lengths;
                    g.Tmax = Tmax;
                    g.spf = spf;
                    g.L = L;
                    g.Cp = Cp;
                    g.ncond = 0;
                    g.ktaps = 1;
                    g.dil = 1;
                    g.h_stride = Cp;
                    g.K = Cp;
                    g.nx = gg_nx;
                    g.ny = h->C / res_bm;
                    g.res_h = H;
                    g.res_hb = Hb;
                    g.res_stride = Hs;
                    if (prof) prof->begin(stream, res_bm == 256 ? "mv::res_gemm256_kernel(mv::GateGemmParams)" : "mv::res_gemm_kernel(mv::GateGemmParams)", f.res_only[i].alg_macs_per_n * spf);
                    rc = launch_res_gemm(g, B, res_bm, stream);
                    if (prof) prof->end(stream);
                    if (rc) return rc;
                } else if (i < nl - 1) {
                    LaunchArgs r;
                    r.lengths = lengths;
                    r.B = B;
                    r.Tmax = Tmax;
                    r.n_rows_per_frame = spf;
                    r.out[0] = ConvOut{H, bsH, 0, (long long)spf * Hs, Hs, ACC_ADD, io16, Hb};
                    if ((rc = launch_layer(f.res_only[i], &sa, 1, r, stream, prof))) return rc;
                }
                continue;
            }
            LaunchArgs r;
            r.lengths = lengths;
            r.B = B;
            r.Tmax = Tmax;
            r.n_rows_per_frame = spf;
            const int skip_mode = i == 0 ? ACC_STORE : ACC_ADD;
            if (i < nl - 1) {
                r.split = h->C;
                r.out[0] = ConvOut{H, bsH, 0, (long long)spf * Hs, Hs, ACC_ADD, io16, Hb};
                r.out[1] = ConvOut{SKIP, bsC, 0, (long long)spf * Cp, Cp, skip_mode, io16};
            } else {
                r.out[0] = ConvOut{SKIP, bsC, 0, (long long)spf * Cp, Cp, skip_mode, io16};
            }
            if ((rc = launch_layer(f.res_skip[i], &sa, 1, r, stream, prof))) return rc;
        }
        const bool early = k % c.n_early_every == 0 && k > 0;
        {
            dim3 grid((L + 63) / 64, B);
            const size_t lds = (size_t)2 * f.n_half * Cp * 4 + 64 * 8 * 4;
            // the next flow's start conv rides along where wg_start_kernel's thread layout applies (else its own launch)
            WgNextStart nxs = {};
            started = false;
            if (k > 0) {
                const WgFlow& g = h->flows[k - 1];
                const int q = Cp / 4;
                if (ng <= 8 && q <= 256 && 256 % q == 0 && g.n_half <= 4 && g.Ck == f.Ck + (early ? c.n_early_size : 0)) {
                    nxs = WgNextStart{g.start_w, g.start_b, g.n_half, h->C, H, io16, Hb, Hs};
                    started = true;
                }
            }
            hipLaunchKernelGGL(wg_tail_kernel, grid, dim3(256), lds, stream, (const void*)SKIP, io16, Cp, f.end_w,
                               h->fold ? f.fold_b : f.end_b, f.n_half, f.Ck, f.winv, AUD, ng, lengths, Tmax, L, spf,
                               early ? noise[ni] : nullptr, early ? c.n_early_size : 0, sigma,
                               h->fold ? (const float*)SKIP : nullptr, fold_parts, (long long)bl * 8, nxs);
            MV_HIP(hipGetLastError());
        }
        if (early) ++ni;
    }
    if (out_format == MV_OUT_PCM16) return launch_pcm16(AUD, bl * ng, (int16_t*)wav, stream);
    return MV_OK;
}

}  // namespace mv

using namespace mv;

extern "C" {

int mv_waveglow_create(const mv_waveglow_cfg* cfg, const mv_tensor* weights, int n_weights, int device, int precision,
                       const mv_options* options, mv_handle* out) {
    if (!cfg || !weights || !out || n_weights <= 0) {
        set_error("mv_waveglow_create: null argument");
        return MV_EINVAL;
    }
    *out = nullptr;
    if (precision != MV_PREC_FP32 && precision != MV_PREC_BF16 && precision != MV_PREC_BF16X3 && precision != MV_PREC_F16) {
        set_error("mv_waveglow_create: unknown precision %d", precision);
        return MV_EINVAL;
    }
    if (cfg->n_flows < 1 || cfg->n_flows > 64 || cfg->n_layers < 1 || cfg->n_layers > 16 || cfg->n_early_every < 1 ||
        cfg->n_early_size < 0 || cfg->n_early_size % 2 != 0 || cfg->n_mel_channels < 1) {
        set_error("mv_waveglow_create: bad config");
        return MV_EINVAL;
    }
    mv_options opt;
    int rc = resolve_options(options, &opt);
    if (rc) return rc;
    DeviceGuard g(device);
    if (!g.ok) {
        set_error("mv_waveglow_create: cannot select HIP device %d", device);
        return MV_EHIP;
    }
    rc = init_kernel_attributes();
    if (rc) return rc;
    if ((rc = init_gate_gemm_attributes())) return rc;
    WeightMap W;
    if ((rc = W.load(weights, n_weights))) return rc;
    WaveglowHandle* h = new WaveglowHandle();
    h->kind = 2;
    h->device = device;
    h->precision = precision;
    h->opt = opt;
    h->cfg = *cfg;
    if ((rc = waveglow_build(h, W))) {
        delete h;
        return rc;
    }
    *out = h;
    return MV_OK;
}

int mv_waveglow_infer(mv_handle hh, const float* mel, const int32_t* lengths, int B, int Tmax,
                      const float* const* noise, int n_noise, float sigma, void* wav, int out_format, void* workspace,
                      size_t workspace_bytes, void* stream) {
    if (!hh || hh->kind != 2) {
        set_error("mv_waveglow_infer: not a WaveGlow handle");
        return MV_EINVAL;
    }
    if (!mel || !wav || !workspace || !noise || B < 1 || Tmax < 1) {
        set_error("mv_waveglow_infer: bad argument (B=%d, Tmax=%d)", B, Tmax);
        return MV_EINVAL;
    }
    if (out_format != MV_OUT_F32 && out_format != MV_OUT_PCM16) {
        set_error("mv_waveglow_infer: unknown output format %d", out_format);
        return MV_EINVAL;
    }
    if ((long long)Tmax * 256 > (1LL << 30)) {
        set_error("mv_waveglow_infer: Tmax too large");
        return MV_EINVAL;
    }
    DeviceGuard g(hh->device);
    if (!g.ok) {
        set_error("mv_waveglow_infer: cannot select HIP device %d", hh->device);
        return MV_EHIP;
    }
    return waveglow_infer(static_cast<WaveglowHandle*>(hh), mel, lengths, B, Tmax, noise, n_noise, sigma, wav,
                          out_format, workspace, workspace_bytes, (hipStream_t)stream);
}

}  // extern "C"
