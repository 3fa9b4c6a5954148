// common.h -- host-side plumbing shared by the libmelvoc translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "../../include/melvoc.h"
#include "conv_mfma.h"

namespace mv {

void set_error(const char* fmt, ...);

#define MV_HIP(call)                                                                           \
    do {                                                                                       \
        hipError_t _e = (call);                                                                \
        if (_e != hipSuccess) {                                                                \
            mv::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
            return MV_EHIP;                                                                    \
        }                                                                                      \
    } while (0)

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// ---- kernel tile configurations -----------------------------------------------------------
struct TileCfg {
    int TM, TN, WM, WN;
    int MT() const { return 16 * TM * WM; }
    int NT() const { return 16 * TN * WN; }
};
extern const TileCfg kTileCfgs[];
extern const int kNumTileCfgs;

// A convolution repacked for conv_cl_kernel.
struct LayerPass {
    int src;  // index into the caller's source bindings
    int cin_off, cb, ktaps, dil, pad_left;
    int tap0;  // first tap of the source this pass covers (taps are split when the dilation exceeds the tile)
};
struct Layer {
    void* w = nullptr;      // device, packed fragments
    float* bias = nullptr;  // device [M]
    int M = 0;
    int steps_total = 0;
    int npass = 0;
    LayerPass pass[kMaxPass];
    int cfg = 0;  // index into kTileCfgs
    bool bf16 = false;  // bf16 MFMA operands ...
    bool x3 = false;    // ... each float32 value as a (hi, lo) pair of them (MV_PREC_BF16X3, conv_mfma.h split_bf16x2)
    int opf = OP_BF16;  // format of the 16-bit operands: OP_BF16, or OP_F16 (MV_PREC_F16)
    int epi = EPI_LINEAR;
    size_t lds_bytes = 0;
    double macs_per_n = 0;  // padded MACs per output row (diagnostics)
    double alg_macs_per_n = 0;  // algorithmic (un-padded) MACs per output row, set by the model builders
};

// Per-launch HIP-event timing, off unless mv_profile_begin() was called on the handle.
struct Profiler {
    struct Rec {
        std::string name, detail;
        double macs_per_frame;
        hipEvent_t e0, e1;
    };
    bool on = false;
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    hipEvent_t get();
    void begin(hipStream_t s, const std::string& name, double macs_per_frame, const std::string& detail = "");
    void end(hipStream_t s);
    void release();
};
std::string layer_kernel_name(const Layer& L);

// One logical input of a layer: C channels (already padded to the k-chunk), taps, and the weight
// as a function (m, tap, c) -> float in the *kernel's* tap order (input row = n + tap*dil - pad_left).
struct SrcSpec {
    int C;
    int ktaps;
    int dil;
    int pad_left;
    std::function<float(int, int, int)> w;
};

// `prec` is MV_PREC_* (0 fp32, 1 bf16, 2 bf16x3, 3 f16) here and in the builders below
// one_pass: pack every source as a single [tap][k-chunk] pass whatever its LDS footprint (weights for the fused
// kernels of fused_rb.h, which keep their own tiles; such a layer is never launched through launch_layer)
int build_layer(Layer* L, const std::vector<SrcSpec>& srcs, int M, const std::vector<float>& bias, int prec, int epi,
                int force_cfg = -1, bool one_pass = false);
int build_gate_gemm_weights(const std::vector<SrcSpec>& srcs, int M, unsigned short** dev);
void free_layer(Layer* L);

struct SrcBind {
    const void* ptr;
    long long batch_stride;
    int row_stride;
    int rows_per_frame;
    float slope;
    int f16 = 0;  // tensor holds float16
};

struct LaunchArgs {
    const int32_t* lengths = nullptr;
    int B = 1;
    int Tmax = 0;
    int n_rows_per_frame = 1;
    int n_extra = 0;
    int split = 1 << 30;
    ConvOut out[2] = {};
    const void* resid = nullptr;  // typed like out[0]
    long long resid_batch_stride = 0;
    float div = 1.f;
    int act = 0;  // 1: tanh on the linear epilogue
};

int launch_layer(const Layer& L, const SrcBind* srcs, int nsrc, const LaunchArgs& a, hipStream_t stream,
                 Profiler* prof = nullptr);
int init_kernel_attributes();

// WaveGlow's gated in-layer conv as a 256 x 256 LDS-DMA GEMM (gate_gemm.h / gate_gemm.hip)
struct GateGemmParams;
int init_gate_gemm_attributes();
bool gate_gemm_supported(int C, int Cp, int ncond, int ktaps);
// (build_gate_gemm_weights below: one layer's weights as the kernel reads them, [M][K] bfloat16 row-major with
// K = (tap, channel) of srcs[0] followed by the channels of srcs[1])
int launch_gate_gemm(const GateGemmParams& p, int B, hipStream_t stream);
int launch_res_gemm(const GateGemmParams& p, int B, int bm, hipStream_t stream);

// ---- weight lookup (state-dict names, weight-norm folding) --------------------------------
struct HostTensor {
    std::vector<float> data;
    std::vector<int64_t> shape;
    int64_t dim(int i) const { return i < (int)shape.size() ? shape[i] : 1; }
};
class WeightMap {
   public:
    int load(const mv_tensor* t, int n);
    // folded weight ("name.weight" or g/v pair) or plain tensor; nullptr + error if missing
    const HostTensor* get(const std::string& name);

   private:
    std::map<std::string, HostTensor> m_;
};

// Conv1d [Cout][Cin][K] as a layer, channels zero-padded to cin_pad / cout_pad (hifigan.hip)
int build_conv1d_layer(Layer* L, const HostTensor* w, const HostTensor* b, int Cin, int Cout, int K, int dil,
                       int cin_pad, int cout_pad, int prec, int force_cfg = -1, bool pair_interleave = false);

template <typename T>
int upload(const std::vector<T>& host, T** dev) {
    MV_HIP(hipMalloc((void**)dev, host.size() * sizeof(T)));
    MV_HIP(hipMemcpy(*dev, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    return MV_OK;
}

}  // namespace mv
