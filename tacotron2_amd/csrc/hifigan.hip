// hifigan.hip -- HiFi-GAN generator (hifigan/models.py:112-174 of the reference).
//
// Data in HBM: channels-last [B][rows][Cp] (Cp = channels rounded up to 32, padding channels are exact zeros
// because their weights and biases are zero); float32 in fp32 mode, float16 between kernels in bf16 mode.
//
// fp32 mode, and the fallback of bf16 mode: one conv_cl_kernel launch per conv (config_v1: 79 launches) --
//   to_channels_last  mel [B][80][T] -> [B][T][96]
//   conv_pre          models.py:150
//   per stage i:      ups[i] as a polyphase conv with the leaky_relu(0.1) of models.py:152 applied
//                     while staging; then for each ResBlock j and dilation m (models.py:96-103)
//                       c1: lrelu -> dilated conv            -> T1
//                       c2: lrelu -> conv, + residual        -> P / Q, the last one accumulates
//                           the MRF mean (models.py:154-160) into ACC (store / add / add-then-divide,
//                           the reference's own order of operations)
//   conv_post         lrelu(0.01) -> Conv1d(C,1,7) -> tanh   (models.py:161-163), fp32 VALU
//
// bf16 mode: the ResBlocks run in the persistent fused kernels of fused_rb.h (config_v1: 27 launches) --
//   C256, C128:  (conv, conv) pairs one launch each (launch_fused_rb), a whole k = 3 ResBlock where its halo leaves
//                >= 80 % of the window
//   C64:         one stage launch for the k = 3 and k = 7 ResBlocks (launch_fused_stage), k = 11 pair by pair
//   C32:         one stage launch for all three ResBlocks + the MRF mean + conv_post
// with the weights of those layers packed a second time in the pair-interleaved channel order the fused kernels
// use (RbPair::c1p / c2p).
#include <cmath>
#include <mutex>

#include "fused_rb.h"

#ifdef MV_TRACE
static constexpr int kFusedMaxLds = 160 * 1024 - 4608;  // window table + the tracer's static LDS
#else
static constexpr int kFusedMaxLds = 160 * 1024 - 4352;  // the kernels' static LDS (window table) comes off the top
#endif
#include "handle.h"

namespace mv {

// ---- mv_hifigan_forward_static: the workspace of callers that cannot allocate -----------------
// A file-scope device array stands in for the caller's workspace argument.  Every GPU the library runs on holds its
// own copy (at its own address) from the moment the code object is loaded until the process ends.  In host code the
// array's NAME is only a stand-in for the symbol, never a device address: static_scratch_ptr() asks the runtime for
// the address once per device (with that device current) and everything else -- kernel arguments, memsets -- uses
// the pointer it returns.
static __device__ __align__(256) unsigned char scratch[MV_STATIC_SCRATCH_BYTES];

constexpr int kMaxDevices = 64;
static void* g_scratch_ptr[kMaxDevices];  // per device; written once under g_scratch_mu
static std::mutex g_scratch_mu;

// The current device's copy of `scratch`; MV_EHIP (and *out = nullptr) if the runtime cannot resolve it.
static int static_scratch_ptr(int device, void** out) {
    *out = nullptr;
    if (device < 0 || device >= kMaxDevices) {
        set_error("static scratch: device %d out of range", device);
        return MV_EINVAL;
    }
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[device]) {
        void* ptr = nullptr;
        MV_HIP(hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch)));  // caller made `device` current
        if (!ptr || ((uintptr_t)ptr & 255) != 0) {
            set_error("static scratch: bad symbol address %p on device %d", ptr, device);
            return MV_EHIP;
        }
        g_scratch_ptr[device] = ptr;
    }
    *out = g_scratch_ptr[device];
    return MV_OK;
}

// ---- fused ResBlock launches (bf16 mode) ------------------------------------------------------
typedef void (*FusedKernel)(const FusedRbParams);
struct FusedCfg {
    int C, WM, WN, TN, npair;
    FusedKernel k;        // float32 window IO
    FusedKernel k16;      // float16 window IO through LDS
    FusedKernel k16_one;  // ... specialised for one pair per launch
    bool x3;              // split bf16 operands (MV_PREC_BF16X3): `k` only, tiles twice as wide per row
    int W() const { return 16 * TN * WN; }
    int threads() const { return 64 * WM * WN; }
    int RS() const { return C * (x3 ? 4 : 2) + kLdsRowPad; }
};
// 8 waves (two per SIMD): the second wave hides LDS / L2 latency and the epilogue phases
#define MV_FUSED_CFG(C, WM, WN, TN, NPAIR)                                                  \
    {                                                                                       \
        C, WM, WN, TN, NPAIR, fused_rb_kernel<C, WM, WN, TN>, fused_rb_kernel<C, WM, WN, TN, true>, \
            fused_rb_kernel<C, WM, WN, TN, true, true>, false                               \
    }
static const FusedCfg kFused[] = {
    MV_FUSED_CFG(32, 1, 8, 6, 3),
    MV_FUSED_CFG(64, 2, 4, 7, 3),
    MV_FUSED_CFG(128, 4, 2, 7, 3),
    MV_FUSED_CFG(256, 8, 1, 7, 1),
};
// float16 operands (MV_PREC_F16): the same kernels, tiles and launch shapes
#define MV_FUSED_CFG_F16(C, WM, WN, TN, NPAIR)                                                                     \
    {                                                                                                              \
        C, WM, WN, TN, NPAIR, fused_rb_kernel<C, WM, WN, TN, false, false, false, OP_F16>,                          \
            fused_rb_kernel<C, WM, WN, TN, true, false, false, OP_F16>,                                             \
            fused_rb_kernel<C, WM, WN, TN, true, true, false, OP_F16>, false                                        \
    }
static const FusedCfg kFusedF16[] = {
    MV_FUSED_CFG_F16(32, 1, 8, 6, 3),
    MV_FUSED_CFG_F16(64, 2, 4, 7, 3),
    MV_FUSED_CFG_F16(128, 4, 2, 7, 3),
    MV_FUSED_CFG_F16(256, 8, 1, 7, 1),
};
// split operands: the (hi, lo) tiles take twice the LDS per row, so the windows are half as long; C = 256 does not
// fit a useful window and stays layer by layer
#define MV_FUSED_X3(C, WM, WN, TN, NPAIR) \
    { C, WM, WN, TN, NPAIR, fused_rb_kernel<C, WM, WN, TN, false, false, true>, nullptr, nullptr, true }
static const FusedCfg kFusedX3[] = {
    MV_FUSED_X3(32, 1, 8, 3, 3),
    MV_FUSED_X3(64, 2, 4, 3, 3),
    MV_FUSED_X3(128, 4, 2, 3, 3),
};

// launch-shape thresholds (measured on the bench batch)
constexpr int kWholeMinPct = 80;  // a whole ResBlock in one launch while NT / W is at least this
constexpr int kPairMinPct = 60;   // a lone (conv, conv) pair below this recomputes too much halo: layer by layer
constexpr int kStageMinPct = 55;  // likewise for a whole-stage launch

static const FusedCfg* fused_cfg_for(int Cp, int precision) {
    if (precision == MV_PREC_BF16X3) {
        for (const FusedCfg& f : kFusedX3)
            if (f.C == Cp) return &f;
        return nullptr;
    }
    for (const FusedCfg& f : precision == MV_PREC_F16 ? kFusedF16 : kFused)
        if (f.C == Cp) return &f;
    return nullptr;
}

// A whole ResBlock (3 pairs) in one launch pays for itself while the halo it recomputes stays small
// next to the window; past that the pairs run as three launches.
static bool fuse_whole(const FusedCfg& fc, int K, const int* dils, int io16) {
    if (fc.npair < 3) return false;
    int H = 0;
    for (int m = 0; m < 3; ++m) H += (K - 1) / 2 * (dils[m] + 1);
    if (((io16 && fc.k16) || fc.x3) && dils[0] * (K - 1) / 2 <= kMaxApron) H -= dils[0] * (K - 1) / 2;  // see launch_fused_rb
    return (fc.W() - 2 * H) * 100 >= fc.W() * kWholeMinPct;
}

// whole-stage launches (all ResBlocks + MRF mean [+ conv_post]) for the HBM-bound low-channel stages
typedef void (*StageKernel)(const FusedStageParams);
struct StageCfg {
    int C, WM, WN, TN;
    StageKernel k[2][2];  // [io16][with conv_post]
    int W() const { return 16 * TN * WN; }
    int threads() const { return 64 * WM * WN; }
};
static const StageCfg kStage[] = {
#define MV_STAGE_CFG(C, WM, WN, TN)                                                                        \
    {                                                                                                      \
        C, WM, WN, TN, {                                                                                   \
            {fused_stage_kernel<C, WM, WN, TN, false, false>, fused_stage_kernel<C, WM, WN, TN, false, true>}, \
                {fused_stage_kernel<C, WM, WN, TN, true, false>, fused_stage_kernel<C, WM, WN, TN, true, true>}, \
        }                                                                                                  \
    }
    MV_STAGE_CFG(32, 1, 8, 6),
    MV_STAGE_CFG(64, 2, 4, 7),
};
static const StageCfg kStageF16[] = {
#define MV_STAGE_CFG_F16(C, WM, WN, TN)                                                    \
    {                                                                                      \
        C, WM, WN, TN, {                                                                   \
            {fused_stage_kernel<C, WM, WN, TN, false, false, OP_F16>,                      \
             fused_stage_kernel<C, WM, WN, TN, false, true, OP_F16>},                      \
                {fused_stage_kernel<C, WM, WN, TN, true, false, OP_F16>,                   \
                 fused_stage_kernel<C, WM, WN, TN, true, true, OP_F16>},                   \
        }                                                                                  \
    }
    MV_STAGE_CFG_F16(32, 1, 8, 6),
    MV_STAGE_CFG_F16(64, 2, 4, 7),
};
static const StageCfg* stage_cfg_for(int Cp, int precision) {
    for (const StageCfg& f : precision == MV_PREC_F16 ? kStageF16 : kStage)
        if (f.C == Cp) return &f;
    return nullptr;
}
// the 16-bit operand modes that share kernels, float16 activation storage and launch structure
static bool is_op16(int precision) { return precision == MV_PREC_BF16 || precision == MV_PREC_F16; }

int init_fused_attributes() {
    for (const FusedCfg& f : kFusedX3)
        MV_HIP(hipFuncSetAttribute((const void*)f.k, hipFuncAttributeMaxDynamicSharedMemorySize, kFusedMaxLds));
    for (const FusedCfg* tab : {kFused, kFusedF16})
        for (int i = 0; i < (int)(sizeof(kFused) / sizeof(kFused[0])); ++i) {
            const FusedCfg& f = tab[i];
            MV_HIP(hipFuncSetAttribute((const void*)f.k, hipFuncAttributeMaxDynamicSharedMemorySize, kFusedMaxLds));
            if (f.k16)
                MV_HIP(hipFuncSetAttribute((const void*)f.k16, hipFuncAttributeMaxDynamicSharedMemorySize, kFusedMaxLds));
            if (f.k16_one)
                MV_HIP(hipFuncSetAttribute((const void*)f.k16_one, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           kFusedMaxLds));
        }
    static_assert(sizeof(kFused) == sizeof(kFusedF16) && sizeof(kStage) == sizeof(kStageF16), "one entry per format");
    for (const StageCfg* tab : {kStage, kStageF16})
        for (int i = 0; i < (int)(sizeof(kStage) / sizeof(kStage[0])); ++i)
            for (int a = 0; a < 2; ++a)
                for (int b = 0; b < 2; ++b)
                    MV_HIP(hipFuncSetAttribute((const void*)tab[i].k[a][b], hipFuncAttributeMaxDynamicSharedMemorySize,
                                               kFusedMaxLds));
    return MV_OK;
}

struct RbPair {
    Layer c1, c2;
    // bf16 only: the same weights with the output channels pair-interleaved, for the fused kernels (fused_rb.h)
    Layer c1p, c2p;
};

struct HifiganHandle : mv_handle_s {
    mv_hifigan_cfg cfg;
    int nstage = 0;
    int C[9];   // channels entering stage i (C[0] = upsample_initial_channel) .. C[nstage]
    int Cp[9];  // padded
    int melp = 0;
    Layer pre;
    std::vector<Layer> ups;
    std::vector<RbPair> rb;  // [(stage*num_kernels + j)*3 + m]
    float* post_w = nullptr;  // device [7][Cp_last]
    float* post_b = nullptr;  // device [1]
    double macs = 0;

    ~HifiganHandle() override {
        free_layer(&pre);
        for (auto& l : ups) free_layer(&l);
        for (auto& p : rb) {
            free_layer(&p.c1);
            free_layer(&p.c2);
            free_layer(&p.c1p);
            free_layer(&p.c2p);
        }
        if (post_w) (void)hipFree(post_w);
        if (post_b) (void)hipFree(post_b);
    }
    int hop() const override {
        int h = 1;
        for (int i = 0; i < nstage; ++i) h *= cfg.upsample_rates[i];
        return h;
    }
    double macs_per_frame() const override { return macs; }
    size_t stage_elems_per_frame() const {
        size_t mx = 0, rows = 1;
        for (int i = 0; i < nstage; ++i) {
            rows *= cfg.upsample_rates[i];
            mx = std::max(mx, rows * (size_t)Cp[i + 1]);
        }
        return mx;
    }
    static size_t al(size_t bytes) { return (bytes + 255) / 256 * 256; }
    size_t workspace_bytes(int B, int Tmax) const override {
        const size_t bt = (size_t)B * Tmax;
        return al(bt * melp * 4) + al(bt * Cp[0] * 4) + 5 * al(bt * stage_elems_per_frame() * 4);
    }
};

// ---- conv_post: lrelu(slope) -> Conv1d(C -> 1, k=7, pad=3) -> tanh -> f32 or PCM16 -----------
template <bool PCM>
__global__ __launch_bounds__(256) void conv_post_kernel(const void* __restrict__ x, int x_f16, long long batch_stride, int Cp,
                                                        const float* __restrict__ w, const float* __restrict__ bias,
                                                        const int32_t* __restrict__ lengths, int Tmax, int hop,
                                                        float slope, void* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ws = (float*)smem;  // [7][Cp]
    for (int i = threadIdx.x; i < 7 * Cp; i += 256) ws[i] = w[i];
    __syncthreads();
    const int b = blockIdx.y;
    const long long L = (long long)Tmax * hop;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= L) return;
    const long long len = (long long)seq_frames(lengths, b, Tmax) * hop;
    float y = 0.f;
    if (t < len) {
        const long long xoff = (long long)b * batch_stride;
        float acc = bias[0];
        for (int k = 0; k < 7; ++k) {
            const long long r = t + k - 3;
            if (r < 0 || r >= len) continue;
            const f32x4* wr = (const f32x4*)(ws + k * Cp);
            for (int c = 0; c < Cp / 4; ++c) {
                const f32x4 v = ld4(x, xoff + r * Cp + c * 4, x_f16), q = wr[c];
                acc += lrelu(v[0], slope) * q[0];
                acc += lrelu(v[1], slope) * q[1];
                acc += lrelu(v[2], slope) * q[2];
                acc += lrelu(v[3], slope) * q[3];
            }
        }
        y = tanhf(acc);
    }
    if constexpr (PCM)
        ((int16_t*)out)[(long long)b * L + t] = pcm16_of(y);
    else
        ((float*)out)[(long long)b * L + t] = y;
}

static int conv_shape_check(const HostTensor* w, int64_t d0, int64_t d1, int64_t d2, const char* name) {
    if (!w) return MV_ENOTFOUND;
    if (w->shape.size() != 3 || w->shape[0] != d0 || w->shape[1] != d1 || w->shape[2] != d2) {
        set_error("%s: expected shape [%lld,%lld,%lld]", name, (long long)d0, (long long)d1, (long long)d2);
        return MV_EINVAL;
    }
    return MV_OK;
}

static int bias_check(const HostTensor* b, int64_t n, const char* name) {
    if (!b) return MV_ENOTFOUND;
    if ((int64_t)b->data.size() != n) {
        set_error("%s: expected %lld entries", name, (long long)n);
        return MV_EINVAL;
    }
    return MV_OK;
}

// Conv1d [Cout][Cin][K] as a layer; output/ input channels padded with zeros.
// pair_interleave: weight row 32g + 16t + r holds output channel 32g + 8(r>>2) + 4t + (r&3), the order in which the
// fused kernels' waves (two M tiles each) own 8 consecutive channels per lane; the bias stays in channel order.
int build_conv1d_layer(Layer* L, const HostTensor* w, const HostTensor* b, int Cin, int Cout, int K, int dil,
                       int cin_pad, int cout_pad, int prec, int force_cfg, bool pair_interleave) {
    const float* wd = w->data.data();
    SrcSpec s;
    s.C = cin_pad;
    s.ktaps = K;
    s.dil = dil;
    s.pad_left = dil * (K - 1) / 2;
    s.w = [=](int row, int tap, int c) -> float {
        const int r = row & 15;
        const int m = pair_interleave ? (row & ~31) + 8 * (r >> 2) + 4 * ((row >> 4) & 1) + (r & 3) : row;
        return (m < Cout && c < Cin) ? wd[((size_t)m * Cin + c) * K + tap] : 0.f;
    };
    std::vector<float> bias(cout_pad, 0.f);
    for (int i = 0; i < Cout; ++i) bias[i] = b ? b->data[i] : 0.f;
    return build_layer(L, {s}, cout_pad, bias, prec, EPI_LINEAR, force_cfg, pair_interleave);
}

// ConvTranspose1d [Cin][Cout][K], stride u, as a J-tap conv with M = u*cout_pad (phase-major).
//   out[t = q*u + p - pad][co] = sum_j sum_ci x[q - j][ci] * w[ci][co][p + j*u]
int build_convT_layer(Layer* L, const HostTensor* w, const HostTensor* b, int Cin, int Cout, int K, int u, int cin_pad,
                      int cout_pad, int prec, int force_cfg = -1) {
    const float* wd = w->data.data();
    const int J = (K + u - 1) / u;
    SrcSpec s;
    s.C = cin_pad;
    s.ktaps = J;
    s.dil = 1;
    s.pad_left = J - 1;
    s.w = [=](int m, int tap, int c) -> float {
        const int p = m / cout_pad, co = m - p * cout_pad;
        const int k = p + (J - 1 - tap) * u;
        return (co < Cout && c < Cin && k < K) ? wd[((size_t)c * Cout + co) * K + k] : 0.f;
    };
    std::vector<float> bias((size_t)u * cout_pad, 0.f);
    for (int p = 0; p < u; ++p)
        for (int i = 0; i < Cout; ++i) bias[(size_t)p * cout_pad + i] = b ? b->data[i] : 0.f;
    return build_layer(L, {s}, u * cout_pad, bias, prec, EPI_LINEAR, force_cfg);
}

// Launch a transposed-conv layer built by build_convT_layer.
int launch_convT(const Layer& L, const void* x, long long x_bs, int cin_stride, int rows_per_frame, float slope,
                 void* y, long long y_bs, int cout_pad, int K, int u, int pad, const int32_t* lengths, int B,
                 int Tmax, hipStream_t stream, Profiler* prof, int in_f16, int out_f16) {
    const int J = (K + u - 1) / u;
    SrcBind sb{x, x_bs, cin_stride, rows_per_frame, slope, in_f16};
    LaunchArgs a;
    a.lengths = lengths;
    a.B = B;
    a.Tmax = Tmax;
    a.n_rows_per_frame = rows_per_frame;
    a.n_extra = J - 1;
    a.out[0].ptr = y;
    a.out[0].batch_stride = y_bs;
    a.out[0].row_stride = u * cout_pad;
    a.out[0].off = -(long long)pad * cout_pad;
    a.out[0].valid_per_frame = (long long)rows_per_frame * u * cout_pad;
    a.out[0].mode = ACC_STORE;
    a.out[0].f16 = out_f16;
    return launch_layer(L, &sb, 1, a, stream, prof);
}

#ifdef MV_SWEEP_CFG
// Diagnostic builds only (tools/sweep_cfg.sh): tile configuration of conv_pre (index 0) and of upsampler i (index i + 1)
// from the environment, "MV_SWEEP_CFG=pre,u0,u1,u2,u3" (-1 = the default).  The product library reads no environment.
static int sweep_cfg(int idx) {
    const char* e = getenv("MV_SWEEP_CFG");
    if (!e) return -1;
    int v[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
    sscanf(e, "%d,%d,%d,%d,%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3], &v[4], &v[5], &v[6], &v[7]);
    return idx < 8 ? v[idx] : -1;
}
#else
static int sweep_cfg(int) { return -1; }
#endif

static int hifigan_build(HifiganHandle* h, WeightMap& W) {
    const mv_hifigan_cfg& c = h->cfg;
    const int bf16 = h->precision;  // MV_PREC_* as the layer builders take it
    char name[160];
    int rc;
    h->nstage = c.num_upsamples;
    h->melp = round_up(c.num_mels, 32);
    for (int i = 0; i <= h->nstage; ++i) {
        h->C[i] = c.upsample_initial_channel >> i;
        h->Cp[i] = round_up(h->C[i], 32);
        if (h->C[i] < 1 || (h->C[i] << i) != c.upsample_initial_channel) {
            set_error("upsample_initial_channel=%d is not divisible by 2^%d", c.upsample_initial_channel, i);
            return MV_EINVAL;
        }
    }
    double macs = 0;
    {
        const HostTensor* w = W.get("conv_pre.weight");
        if ((rc = conv_shape_check(w, h->C[0], c.num_mels, 7, "conv_pre.weight"))) return rc;
        const HostTensor* b = W.get("conv_pre.bias");
        if ((rc = bias_check(b, h->C[0], "conv_pre.bias"))) return rc;
        if ((rc = build_conv1d_layer(&h->pre, w, b, c.num_mels, h->C[0], 7, 1, h->melp, h->Cp[0], bf16, sweep_cfg(0)))) return rc;
        h->pre.alg_macs_per_n = (double)h->C[0] * c.num_mels * 7;
        macs += (double)h->C[0] * c.num_mels * 7;
    }
    h->ups.resize(h->nstage);
    h->rb.resize((size_t)h->nstage * c.num_kernels * 3);
    double rows = 1;
    for (int i = 0; i < h->nstage; ++i) {
        const int u = c.upsample_rates[i], K = c.upsample_kernel_sizes[i];
        if (u < 1 || K < u || (K - u) % 2 != 0) {
            set_error("ups.%d: kernel %d / stride %d unsupported (needs K >= stride, K - stride even)", i, K, u);
            return MV_EUNSUPPORTED;
        }
        snprintf(name, sizeof name, "ups.%d.weight", i);
        const HostTensor* w = W.get(name);
        if ((rc = conv_shape_check(w, h->C[i], h->C[i + 1], K, name))) return rc;
        snprintf(name, sizeof name, "ups.%d.bias", i);
        const HostTensor* b = W.get(name);
        if ((rc = bias_check(b, h->C[i + 1], name))) return rc;
        if ((rc = build_convT_layer(&h->ups[i], w, b, h->C[i], h->C[i + 1], K, u, h->Cp[i], h->Cp[i + 1], bf16, sweep_cfg(i + 1))))
            return rc;
        h->ups[i].alg_macs_per_n = (double)h->C[i] * h->C[i + 1] * K;
        macs += rows * (double)h->C[i] * h->C[i + 1] * K;  // every input row meets all K taps
        rows *= u;
        const int Cc = h->C[i + 1], Ccp = h->Cp[i + 1];
        for (int j = 0; j < c.num_kernels; ++j) {
            const int k = c.resblock_kernel_sizes[j];
            if (k % 2 != 1) {
                set_error("resblock kernel size %d must be odd", k);
                return MV_EUNSUPPORTED;
            }
            for (int m = 0; m < 3; ++m) {
                const int n = i * c.num_kernels + j;
                RbPair& pr = h->rb[(size_t)n * 3 + m];
                const int d = c.resblock_dilations[j][m];
                for (int which = 0; which < 2; ++which) {
                    snprintf(name, sizeof name, "resblocks.%d.convs%d.%d.weight", n, which + 1, m);
                    const HostTensor* cw = W.get(name);
                    if ((rc = conv_shape_check(cw, Cc, Cc, k, name))) return rc;
                    snprintf(name, sizeof name, "resblocks.%d.convs%d.%d.bias", n, which + 1, m);
                    const HostTensor* cb = W.get(name);
                    if ((rc = bias_check(cb, Cc, name))) return rc;
                    if ((rc = build_conv1d_layer(which ? &pr.c2 : &pr.c1, cw, cb, Cc, Cc, k, which ? 1 : d, Ccp, Ccp,
                                                 bf16)))
                        return rc;
                    (which ? pr.c2 : pr.c1).alg_macs_per_n = (double)Cc * Cc * k;
                    if (bf16 != MV_PREC_FP32 && (rc = build_conv1d_layer(which ? &pr.c2p : &pr.c1p, cw, cb, Cc, Cc, k, which ? 1 : d, Ccp,
                                                         Ccp, bf16, -1, true)))
                        return rc;
                    macs += rows * (double)Cc * Cc * k;
                }
            }
        }
    }
    {
        const int Cl = h->C[h->nstage], Clp = h->Cp[h->nstage];
        const HostTensor* w = W.get("conv_post.weight");
        if ((rc = conv_shape_check(w, 1, Cl, 7, "conv_post.weight"))) return rc;
        const HostTensor* b = W.get("conv_post.bias");
        if ((rc = bias_check(b, 1, "conv_post.bias"))) return rc;
        std::vector<float> pw((size_t)7 * Clp, 0.f);
        for (int cc = 0; cc < Cl; ++cc)
            for (int k = 0; k < 7; ++k) pw[(size_t)k * Clp + cc] = w->data[(size_t)cc * 7 + k];
        if ((rc = upload(pw, &h->post_w))) return rc;
        if ((rc = upload(b->data, &h->post_b))) return rc;
        macs += rows * (double)Cl * 7;
    }
    h->macs = macs;
    return MV_OK;
}

static int cu_count() {
    static int n[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (n[dev] == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        n[dev] = v;
    }
    return n[dev];
}

// Runs pairs [m0, m0+npair) of one ResBlock fused; returns MV_EUNSUPPORTED if the window is too small.
static int launch_fused_rb(const FusedCfg& fc, const RbPair* pairs, int m0, int npair, int K, const int* dils,
                           const void* x, void* out, long long bs, int rpf, int mode, float div,
                           const int32_t* lengths, int B, int Tmax, hipStream_t stream, Profiler* prof, int io16) {
    FusedRbParams p;
    memset(&p, 0, sizeof p);
    p.x = x;
    p.out = out;
    p.io16 = io16;
    p.batch_stride = bs;
    p.lengths = lengths;
    p.B = B;
    p.Tmax = Tmax;
    p.rows_per_frame = rpf;
    p.K = K;
    p.npair = npair;
    p.mode = mode;
    p.div = div;
    const int pad2 = (K - 1) / 2;
    int H = 0, gA = 0;
    double macs = 0;
    for (int i = 0; i < npair; ++i) {
        const RbPair& pr = pairs[m0 + i];
        if (pr.c1p.npass != 1 || pr.c2p.npass != 1) return MV_EUNSUPPORTED;  // fused kernel wants [tap][kc] order
        p.dil[i] = dils[m0 + i];
        p.w1[i] = pr.c1p.w;
        p.w2[i] = pr.c2p.w;
        p.b1[i] = pr.c1.bias;
        p.b2[i] = pr.c2.bias;
        // the float16-IO kernels give the first pair's conv1 real rows either side of the window (apron)
        if (i == 0 && ((io16 && fc.k16) || fc.x3) && dils[m0] * pad2 <= kMaxApron) {
            p.g1 = dils[m0] * pad2;
            H += pad2;
        } else {
            H += pad2 * (dils[m0 + i] + 1);
        }
        gA = std::max(gA, dils[m0 + i] * pad2);
        macs += pr.c1.alg_macs_per_n + pr.c2.alg_macs_per_n;
    }
    p.H = H;
    p.NT = fc.W() - 2 * H;
    p.guardA = gA;
    p.guardY = pad2;
    const int RS = fc.RS();
    const size_t lds = (size_t)(2 * fc.W() + 2 * gA + 2 * pad2) * RS + (size_t)2 * npair * fc.C * 4;  // tiles + biases
    if (p.NT < 16 || lds > kFusedMaxLds || p.NT * 100 < fc.W() * kPairMinPct || B > kMaxSeq) return MV_EUNSUPPORTED;
    const long long rows = (long long)Tmax * rpf;
    // persistent blocks, one per CU (LDS allows no more), each walking every gridDim.x-th existing window
    const long long vmax = ((rows + p.NT - 1) / p.NT) * B;
    dim3 grid((unsigned)std::min<long long>(vmax, (long long)cu_count()));
    const bool one = io16 && npair == 1;
    FusedKernel kern = one ? fc.k16_one : io16 ? fc.k16 : fc.k;
    if (prof) {
        char nm[128], dt[32];  // the symbol as rocprofv3 prints it + which of its launches
        const int opf = &fc >= kFusedF16 && &fc < kFusedF16 + sizeof(kFusedF16) / sizeof(kFusedF16[0]) ? OP_F16 : OP_BF16;
        snprintf(nm, sizeof nm, "void mv::fused_rb_kernel<%d, %d, %d, %d, %s, %s, %s, %d>(mv::FusedRbParams)", fc.C, fc.WM,
                 fc.WN, fc.TN, io16 ? "true" : "false", one ? "true" : "false", fc.x3 ? "true" : "false", opf);
        snprintf(dt, sizeof dt, "k=%d", K);
        prof->begin(stream, nm, macs * rpf, dt);
    }
    hipLaunchKernelGGL(kern, grid, dim3(fc.threads()), lds, stream, p);
    if (prof) prof->end(stream);
    MV_HIP(hipGetLastError());
    return MV_OK;
}

// All ResBlocks of stage `i` (+ conv_post when `wav` is given) in one launch.
// The first `nrb` ResBlocks of stage `i` (+ conv_post when `wav` is given, which needs all of them) in one launch:
// out = their sum, divided by num_kernels when the launch covers them all.
static int launch_fused_stage(const StageCfg& sc, const HifiganHandle* h, int i, int nrb, const void* x, void* out,
                              long long bs, int rpf, void* wav, int out_format, const int32_t* lengths, int B, int Tmax,
                              hipStream_t stream, Profiler* prof, int io16) {
    const mv_hifigan_cfg& c = h->cfg;
    if (c.num_kernels > 3 || nrb < 1 || nrb > c.num_kernels || (wav && nrb != c.num_kernels)) return MV_EUNSUPPORTED;
    FusedStageParams p;
    memset(&p, 0, sizeof p);
    p.x = x;
    p.out = out;
    p.io16 = io16;
    p.batch_stride = bs;
    p.lengths = lengths;
    p.B = B;
    p.Tmax = Tmax;
    p.rows_per_frame = rpf;
    p.nrb = nrb;
    p.div = nrb == c.num_kernels ? (float)c.num_kernels : 1.f;
    int H = 0, gA = 0, gY = 0;
    double macs = 0;
    for (int j = 0; j < nrb; ++j) {
        const int K = c.resblock_kernel_sizes[j];
        const int pad2 = (K - 1) / 2;
        p.K[j] = K;
        int Hj = 0;
        for (int m = 0; m < 3; ++m) {
            const RbPair& pr = h->rb[((size_t)i * c.num_kernels + j) * 3 + m];
            if (pr.c1p.npass != 1 || pr.c2p.npass != 1) return MV_EUNSUPPORTED;
            const int d = c.resblock_dilations[j][m];
            p.dil[j][m] = d;
            p.w1[j][m] = pr.c1p.w;
            p.w2[j][m] = pr.c2p.w;
            p.b1[j][m] = pr.c1.bias;
            p.b2[j][m] = pr.c2.bias;
            if (m == 0 && io16 && d * pad2 <= kMaxApron) {
                p.g1[j] = d * pad2;
                Hj += pad2;
            } else {
                Hj += pad2 * (d + 1);
            }
            gA = std::max(gA, d * pad2);
            macs += pr.c1.alg_macs_per_n + pr.c2.alg_macs_per_n;
        }
        H = std::max(H, Hj);
        gY = std::max(gY, pad2);
    }
    if (wav) {
        H += 3;
        p.post_w = h->post_w;
        p.post_b = h->post_b;
        p.wav = wav;
        p.wav_pcm16 = out_format == MV_OUT_PCM16;
        macs += (double)h->C[h->nstage] * 7;
    }
    p.H = H;
    p.NT = sc.W() - 2 * H;
    p.guardA = gA;
    p.guardY = gY;
    const int RS = sc.C * 2 + kLdsRowPad;
    const size_t lds = (size_t)(2 * sc.W() + 2 * gA + 2 * gY) * RS + (size_t)6 * nrb * sc.C * 4;  // + biases
    if (lds > kFusedMaxLds || p.NT * 100 < sc.W() * kStageMinPct || B > kMaxSeq) return MV_EUNSUPPORTED;
    const long long rows = (long long)Tmax * rpf;
    const long long vmax = ((rows + p.NT - 1) / p.NT) * B;
    dim3 grid((unsigned)std::min<long long>(vmax, cu_count()));
    // the persistent blocks visit only windows that exist; the waveform past a sequence's end is zero
    if (wav) MV_HIP(hipMemsetAsync(wav, 0, (size_t)B * rows * (out_format == MV_OUT_PCM16 ? 2 : 4), stream));
    if (prof) {
        char nm[128];
        snprintf(nm, sizeof nm, "void mv::fused_stage_kernel<%d, %d, %d, %d, %s, %s, %d>(mv::FusedStageParams)", sc.C, sc.WM,
                 sc.WN, sc.TN, io16 ? "true" : "false", wav ? "true" : "false",
                 h->precision == MV_PREC_F16 ? OP_F16 : OP_BF16);
        prof->begin(stream, nm, macs * rpf, wav ? "+conv_post" : nrb < c.num_kernels ? "first ResBlocks" : "");
    }
    hipLaunchKernelGGL(sc.k[io16 ? 1 : 0][wav ? 1 : 0], grid, dim3(sc.threads()), lds, stream, p);
    if (prof) prof->end(stream);
    MV_HIP(hipGetLastError());
    return MV_OK;
}

static int hifigan_forward(HifiganHandle* h, const float* mel, const int32_t* lengths, int B, int Tmax, void* wav,
                           int out_format, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    const mv_hifigan_cfg& c = h->cfg;
    if (workspace_bytes < h->workspace_bytes(B, Tmax) || ((uintptr_t)workspace & 255) != 0) {
        set_error("workspace: need %zu bytes, 256-byte aligned (got %zu)", h->workspace_bytes(B, Tmax),
                  workspace_bytes);
        return MV_ENOMEM;
    }
    int rc;
    Profiler* prof = h->prof.on ? &h->prof : nullptr;
    // bf16 mode: activations between kernels are float16 (see ld4/st4 in conv_mfma.h); fp32 mode: float32
    const int io16 = is_op16(h->precision) ? h->opt.io16 : 0;
    const size_t bt = (size_t)B * Tmax;
    char* ws = (char*)workspace;
    float* X0 = (float*)ws;
    ws += HifiganHandle::al(bt * h->melp * 4);
    float* PRE = (float*)ws;
    ws += HifiganHandle::al(bt * h->Cp[0] * 4);
    const size_t stage_bytes = HifiganHandle::al(bt * h->stage_elems_per_frame() * 4);
    float* buf[5];
    for (int i = 0; i < 5; ++i) buf[i] = (float*)(ws + i * stage_bytes);

    if ((rc = launch_to_channels_last(mel, B, c.num_mels, Tmax, X0, h->melp, stream))) return rc;
    {
        SrcBind sb{X0, (long long)Tmax * h->melp, h->melp, 1, 1.0f};
        LaunchArgs a;
        a.lengths = lengths;
        a.B = B;
        a.Tmax = Tmax;
        a.out[0] = ConvOut{PRE, (long long)Tmax * h->Cp[0], 0, h->Cp[0], h->Cp[0], ACC_STORE, io16};
        if ((rc = launch_layer(h->pre, &sb, 1, a, stream, prof))) return rc;
    }
    const float* cur_in = PRE;  // input of the next upsampling
    int rpf = 1;                // rows per mel frame of cur_in
    int acc_idx = 1;            // buf[acc_idx] receives the MRF mean of this stage
    bool post_done = false;     // conv_post already fused into the last stage's launch
    for (int i = 0; i < h->nstage; ++i) {
        const int u = c.upsample_rates[i], K = c.upsample_kernel_sizes[i];
        const int Cin_p = h->Cp[i], Cp = h->Cp[i + 1];
        const int rpf_out = rpf * u;
        // buffers of this stage: A (ups output), ACC (MRF mean), T1, P, Q
        float* A = buf[(acc_idx + 1) % 5];
        float* ACC = buf[(acc_idx + 2) % 5];
        float* T1 = buf[(acc_idx + 3) % 5];
        float* PQ[2] = {buf[(acc_idx + 4) % 5], buf[acc_idx % 5]};  // the previous ACC is free once A is built
        const long long bs_in = (long long)Tmax * rpf * Cin_p;
        const long long bs = (long long)Tmax * rpf_out * Cp;
        if ((rc = launch_convT(h->ups[i], cur_in, bs_in, Cin_p, rpf, 0.1f, A, bs, Cp, K, u, (K - u) / 2, lengths, B,
                               Tmax, stream, prof, io16, io16)))
            return rc;
        const FusedCfg* fc = h->precision != MV_PREC_FP32 && h->opt.fuse >= 1 ? fused_cfg_for(Cp, h->precision) : nullptr;
        const StageCfg* sc = is_op16(h->precision) && h->opt.fuse >= 2 ? stage_cfg_for(Cp, h->precision) : nullptr;
        int j0 = 0;  // ResBlocks [0, j0) are done by a stage launch
        if (sc) {
            const bool last = i == h->nstage - 1;
            // The window of a stage launch pays the halo of its widest ResBlock for all of them.  Where HBM is not
            // the bound (C64), the widest (last) one runs pair by pair afterwards with each pair's own, smaller halo.
            int nrb = c.num_kernels;
            if (!last && Cp == 64 && fc && nrb == 3 && io16) nrb = 2;
            rc = launch_fused_stage(*sc, h, i, nrb, A, ACC, bs, rpf_out, last ? wav : nullptr, out_format, lengths, B,
                                    Tmax, stream, prof, io16);
            if (rc == MV_OK) {
                j0 = nrb;
                if (nrb == c.num_kernels) {
                    cur_in = ACC;
                    rpf = rpf_out;
                    acc_idx = (acc_idx + 2) % 5;
                    post_done = last;
                    continue;
                }
            } else if (rc != MV_EUNSUPPORTED) {
                return rc;
            }
        }
        for (int j = j0; j < c.num_kernels; ++j) {
            const float* cur = A;
            const RbPair* pairs = &h->rb[((size_t)i * c.num_kernels + j) * 3];
            const int acc_mode =
                c.num_kernels == 1 || j == 0 ? ACC_STORE : (j == c.num_kernels - 1 ? ACC_ADD_DIV : ACC_ADD);
            const int krb = c.resblock_kernel_sizes[j];
            if (fc && fuse_whole(*fc, krb, c.resblock_dilations[j], io16)) {
                rc = launch_fused_rb(*fc, pairs, 0, 3, krb, c.resblock_dilations[j], A, ACC, bs, rpf_out, acc_mode,
                                     (float)c.num_kernels, lengths, B, Tmax, stream, prof, io16);
                if (rc == MV_OK) continue;
                if (rc != MV_EUNSUPPORTED) return rc;
            }
            for (int m = 0; m < 3; ++m) {
                const RbPair& pr = pairs[m];
                float* dst = m < 2 ? PQ[m] : ACC;
                const int mode = m < 2 ? ACC_STORE : acc_mode;
                if (fc) {
                    rc = launch_fused_rb(*fc, pairs, m, 1, krb, c.resblock_dilations[j], cur, dst, bs, rpf_out, mode,
                                         (float)c.num_kernels, lengths, B, Tmax, stream, prof, io16);
                    if (rc == MV_OK) {
                        cur = dst;
                        continue;
                    }
                    if (rc != MV_EUNSUPPORTED) return rc;
                }
                LaunchArgs a;
                a.lengths = lengths;
                a.B = B;
                a.Tmax = Tmax;
                a.n_rows_per_frame = rpf_out;
                a.out[0] = ConvOut{T1, bs, 0, (long long)rpf_out * Cp, Cp, ACC_STORE, io16};
                SrcBind s1{cur, bs, Cp, rpf_out, 0.1f, io16};
                if ((rc = launch_layer(pr.c1, &s1, 1, a, stream, prof))) return rc;
                SrcBind s2{T1, bs, Cp, rpf_out, 0.1f, io16};
                a.resid = cur;
                a.resid_batch_stride = bs;
                a.out[0].ptr = dst;
                a.out[0].mode = mode;
                a.div = (float)c.num_kernels;
                if ((rc = launch_layer(pr.c2, &s2, 1, a, stream, prof))) return rc;
                cur = dst;
            }
        }
        cur_in = ACC;
        rpf = rpf_out;
        acc_idx = (acc_idx + 2) % 5;
    }
    if (!post_done) {
        const int Clp = h->Cp[h->nstage];
        const long long L = (long long)Tmax * rpf;
        dim3 grid((unsigned)((L + 255) / 256), (unsigned)B);
        const size_t lds = (size_t)7 * Clp * 4;
        if (out_format == MV_OUT_PCM16)
            hipLaunchKernelGGL(conv_post_kernel<true>, grid, dim3(256), lds, stream, (const void*)cur_in, io16, L * Clp, Clp, h->post_w,
                               h->post_b, lengths, Tmax, rpf, 0.01f, wav);
        else
            hipLaunchKernelGGL(conv_post_kernel<false>, grid, dim3(256), lds, stream, (const void*)cur_in, io16, L * Clp, Clp, h->post_w,
                               h->post_b, lengths, Tmax, rpf, 0.01f, wav);
        MV_HIP(hipGetLastError());
    }
    return MV_OK;
}

}  // namespace mv

using namespace mv;

extern "C" {

int mv_hifigan_create(const mv_hifigan_cfg* cfg, const mv_tensor* weights, int n_weights, int device, int precision,
                      const mv_options* options, mv_handle* out) {
    if (!cfg || !weights || !out || n_weights <= 0) {
        set_error("mv_hifigan_create: null argument");
        return MV_EINVAL;
    }
    *out = nullptr;
    if (precision != MV_PREC_FP32 && precision != MV_PREC_BF16 && precision != MV_PREC_BF16X3 && precision != MV_PREC_F16) {
        set_error("mv_hifigan_create: unknown precision %d", precision);
        return MV_EINVAL;
    }
    if (cfg->num_upsamples < 1 || cfg->num_upsamples > 8 || cfg->num_kernels < 1 || cfg->num_kernels > 8 ||
        cfg->num_mels < 1 || cfg->upsample_initial_channel < 1) {
        set_error("mv_hifigan_create: bad config");
        return MV_EINVAL;
    }
    mv_options opt;
    int rc = resolve_options(options, &opt);
    if (rc) return rc;
    DeviceGuard g(device);
    if (!g.ok) {
        set_error("mv_hifigan_create: cannot select HIP device %d", device);
        return MV_EHIP;
    }
    rc = init_kernel_attributes();
    if (rc) return rc;
    if ((rc = init_fused_attributes())) return rc;
    // this device's static scratch is resolved here (device current, no stream work pending) so that
    // mv_hifigan_forward_static finds it in a table; a failure is NOT this call's failure -- callers that never use
    // the static entry point lose nothing, and mv_hifigan_forward_static reports MV_EHIP itself
    void* sp = nullptr;
    (void)static_scratch_ptr(device, &sp);
    WeightMap W;
    if ((rc = W.load(weights, n_weights))) return rc;
    HifiganHandle* h = new HifiganHandle();
    h->kind = 1;
    h->device = device;
    h->precision = precision;
    h->opt = opt;
    h->cfg = *cfg;
    if ((rc = hifigan_build(h, W))) {
        delete h;
        return rc;
    }
    *out = h;
    return MV_OK;
}

int mv_hifigan_forward(mv_handle hh, const float* mel, const int32_t* lengths, int B, int Tmax, void* wav,
                       int out_format, void* workspace, size_t workspace_bytes, void* stream) {
    if (!hh || hh->kind != 1) {
        set_error("mv_hifigan_forward: not a HiFi-GAN handle");
        return MV_EINVAL;
    }
    if (!mel || !wav || !workspace || B < 1 || Tmax < 1) {
        set_error("mv_hifigan_forward: bad argument (B=%d, Tmax=%d)", B, Tmax);
        return MV_EINVAL;
    }
    if (out_format != MV_OUT_F32 && out_format != MV_OUT_PCM16) {
        set_error("mv_hifigan_forward: unknown output format %d", out_format);
        return MV_EINVAL;
    }
    if ((long long)Tmax * hh->hop() > (1LL << 30)) {
        set_error("mv_hifigan_forward: Tmax too large");
        return MV_EINVAL;
    }
    DeviceGuard g(hh->device);
    if (!g.ok) {
        set_error("mv_hifigan_forward: cannot select HIP device %d", hh->device);
        return MV_EHIP;
    }
    return hifigan_forward(static_cast<HifiganHandle*>(hh), mel, lengths, B, Tmax, wav, out_format, workspace,
                           workspace_bytes, (hipStream_t)stream);
}

int mv_hifigan_forward_static(mv_handle hh, const float* mel, const int32_t* lengths, int B, int Tmax, void* wav,
                              int out_format, void* stream) {
    if (!hh || hh->kind != 1) {
        set_error("mv_hifigan_forward_static: not a HiFi-GAN handle");
        return MV_EINVAL;
    }
    if (!mel || !wav || B < 1 || Tmax < 1) {
        set_error("mv_hifigan_forward_static: bad argument (B=%d, Tmax=%d)", B, Tmax);
        return MV_EINVAL;
    }
    if (out_format != MV_OUT_F32 && out_format != MV_OUT_PCM16) {
        set_error("mv_hifigan_forward_static: unknown output format %d", out_format);
        return MV_EINVAL;
    }
    HifiganHandle* h = static_cast<HifiganHandle*>(hh);
    // sizes first (in 64 bits: B * Tmax may be anything here), before anything is launched
    const unsigned long long per_frame = h->workspace_bytes(1, 1);
    if ((unsigned long long)B * (unsigned long long)Tmax > MV_STATIC_SCRATCH_BYTES / per_frame ||
        h->workspace_bytes(B, Tmax) > MV_STATIC_SCRATCH_BYTES) {
        set_error("mv_hifigan_forward_static: B * Tmax = %lld frames need %llu bytes of scratch, the static buffer "
                  "holds %llu (about %llu frames); use mv_hifigan_forward with a workspace of your own",
                  (long long)B * Tmax, (unsigned long long)B * Tmax * per_frame,
                  (unsigned long long)MV_STATIC_SCRATCH_BYTES, (unsigned long long)(MV_STATIC_SCRATCH_BYTES / per_frame));
        return MV_ENOMEM;
    }
    DeviceGuard g(hh->device);
    if (!g.ok) {
        set_error("mv_hifigan_forward_static: cannot select HIP device %d", hh->device);
        return MV_EHIP;
    }
    void* sp = nullptr;
    int rc = static_scratch_ptr(hh->device, &sp);  // resolved at create: a table read here
    if (rc) return rc;
    return hifigan_forward(h, mel, lengths, B, Tmax, wav, out_format, sp, MV_STATIC_SCRATCH_BYTES, (hipStream_t)stream);
}

}  // extern "C"

#ifdef MV_TRACE
// Diagnostic builds only (tools/trace.sh): fetch and reset the per-block phase records of fused_rb.h.
extern "C" __attribute__((visibility("default"))) long long mv_trace_read(unsigned long long* host, long long max_records) {
    unsigned int n = 0;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(mv::g_trace_n), sizeof(n)) != hipSuccess) return -1;
    long long take = n < (unsigned)mv::kTraceCap ? n : mv::kTraceCap;
    if (take > max_records) take = max_records;
    if (host && take > 0 &&
        hipMemcpyFromSymbol(host, HIP_SYMBOL(mv::g_trace), (size_t)take * mv::kTraceSlots * 8) != hipSuccess)
        return -1;
    const unsigned int zero = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(mv::g_trace_n), &zero, sizeof(zero)) != hipSuccess) return -1;
    return take;
}
#endif
