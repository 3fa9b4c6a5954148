// postnet.hip -- Tacotron2 Postnet (model.py:140-174 of the reference) on conv_cl_kernel: the step right
// before the vocoder.  n Conv1d(k, pad (k-1)/2) + BatchNorm1d (eval: folded into the
// conv at create) with tanh after all but the last; model.py:717 adds the result to its input.
//
// Data in HBM: float32 channels-last.  X0 [B][T][melp] (the transposed input), two ping-pong buffers
// [B][T][Cp] for the hidden layers, X1 [B][T][melp] for the last conv (+ residual), transposed back into the
// reference's [B][dim][T].  Ragged batches: every conv zero-pads at each sequence's own end (lengths), so row b
// equals the reference's batch-1 result for x[b, :, :lengths[b]]; frames past lengths[b] are written as 0.
#include <cmath>

#include "handle.h"

namespace mv {

struct PostnetHandle : mv_handle_s {
    mv_postnet_cfg cfg;
    int melp = 0, Cp = 0;
    std::vector<Layer> convs;
    double macs = 0;

    ~PostnetHandle() override {
        for (auto& l : convs) free_layer(&l);
    }
    int hop() const override { return 1; }
    double macs_per_frame() const override { return macs; }
    static size_t al(size_t bytes) { return (bytes + 255) / 256 * 256; }
    size_t workspace_bytes(int B, int Tmax) const override {
        const size_t bt = (size_t)B * Tmax;
        return 2 * al(bt * melp * 4) + 2 * al(bt * Cp * 4);
    }
};

// [B][T][Cp] channels-last -> [B][C][T]; frames at or past lengths[b] are written as zeros
__global__ __launch_bounds__(256) void from_channels_last_kernel(const float* __restrict__ x, int Cp, int C, int T,
                                                                 const int32_t* __restrict__ lengths,
                                                                 float* __restrict__ y) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int len = seq_frames(lengths, b, T);
    for (int i = ty; i < 32; i += 8) {
        const int t = t0 + i, c = c0 + tx;
        tile[i][tx] = (t < len && c < Cp) ? x[((long long)b * T + t) * Cp + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, t = t0 + tx;
        if (c < C && t < T) y[((long long)b * C + c) * T + t] = tile[tx][i];
    }
}

static int postnet_build(PostnetHandle* h, WeightMap& W) {
    const mv_postnet_cfg& c = h->cfg;
    const int bf16 = h->precision;  // MV_PREC_* as the layer builders take it
    const int n = c.n_convolutions, K = c.kernel_size, D = c.dim_data, E = c.embedding_dim;
    h->melp = round_up(D, 32);
    h->Cp = round_up(E, 32);
    h->convs.resize(n);
    char name[160];
    int rc;
    double macs = 0;
    for (int i = 0; i < n; ++i) {
        const int Cin = i == 0 ? D : E, Cout = i == n - 1 ? D : E;
        const int cin_p = i == 0 ? h->melp : h->Cp, cout_p = i == n - 1 ? h->melp : h->Cp;
        snprintf(name, sizeof name, "convolutions.%d.0.conv.weight", i);
        const HostTensor* w = W.get(name);
        if (!w) return MV_ENOTFOUND;
        if (w->shape.size() != 3 || w->shape[0] != Cout || w->shape[1] != Cin || w->shape[2] != K) {
            set_error("%s: expected shape [%d,%d,%d]", name, Cout, Cin, K);
            return MV_EINVAL;
        }
        snprintf(name, sizeof name, "convolutions.%d.0.conv.bias", i);
        const HostTensor* b = W.get(name);
        const char* bn_keys[4] = {"weight", "bias", "running_mean", "running_var"};
        const HostTensor* bn[4];
        for (int q = 0; q < 4; ++q) {
            snprintf(name, sizeof name, "convolutions.%d.1.%s", i, bn_keys[q]);
            bn[q] = W.get(name);
            if (!bn[q]) return MV_ENOTFOUND;
            if ((int)bn[q]->data.size() != Cout) {
                set_error("%s: expected %d entries", name, Cout);
                return MV_EINVAL;
            }
        }
        if (!b) return MV_ENOTFOUND;
        if ((int)b->data.size() != Cout) {
            set_error("convolutions.%d.0.conv.bias: expected %d entries", i, Cout);
            return MV_EINVAL;
        }
        // BatchNorm1d in eval mode (model.py:153,160,166): y = (conv + b - mean) / sqrt(var + eps) * gamma + beta
        HostTensor wf = *w, bf;
        bf.data.resize(Cout);
        bf.shape = {Cout};
        for (int m = 0; m < Cout; ++m) {
            const double s = (double)bn[0]->data[m] / std::sqrt((double)bn[3]->data[m] + (double)c.bn_eps);
            for (size_t q = 0; q < (size_t)Cin * K; ++q)
                wf.data[(size_t)m * Cin * K + q] = (float)((double)w->data[(size_t)m * Cin * K + q] * s);
            bf.data[m] = (float)(((double)b->data[m] - (double)bn[2]->data[m]) * s + (double)bn[1]->data[m]);
        }
        if ((rc = build_conv1d_layer(&h->convs[i], &wf, &bf, Cin, Cout, K, 1, cin_p, cout_p, bf16))) return rc;
        h->convs[i].alg_macs_per_n = (double)Cin * Cout * K;
        macs += (double)Cin * Cout * K;
    }
    h->macs = macs;
    return MV_OK;
}

static int postnet_forward(PostnetHandle* h, const float* x, const int32_t* lengths, int B, int Tmax, float* y,
                           int add_input, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (workspace_bytes < h->workspace_bytes(B, Tmax) || ((uintptr_t)workspace & 255) != 0) {
        set_error("workspace: need %zu bytes, 256-byte aligned (got %zu)", h->workspace_bytes(B, Tmax),
                  workspace_bytes);
        return MV_ENOMEM;
    }
    const int n = h->cfg.n_convolutions, melp = h->melp, Cp = h->Cp, D = h->cfg.dim_data;
    Profiler* prof = h->prof.on ? &h->prof : nullptr;
    const size_t bt = (size_t)B * Tmax;
    char* ws = (char*)workspace;
    float* X0 = (float*)ws;
    ws += PostnetHandle::al(bt * melp * 4);
    float* X1 = (float*)ws;
    ws += PostnetHandle::al(bt * melp * 4);
    float* P[2];
    P[0] = (float*)ws;
    ws += PostnetHandle::al(bt * Cp * 4);
    P[1] = (float*)ws;
    int rc;
    if ((rc = launch_to_channels_last(x, B, D, Tmax, X0, melp, stream))) return rc;
    const float* cur = X0;
    int cur_c = melp;
    for (int i = 0; i < n; ++i) {
        const bool last = i == n - 1;
        float* dst = last ? X1 : P[i & 1];
        const int dst_c = last ? melp : Cp;
        SrcBind sb{cur, (long long)Tmax * cur_c, cur_c, 1, 1.0f};
        LaunchArgs a;
        a.lengths = lengths;
        a.B = B;
        a.Tmax = Tmax;
        a.act = last ? 0 : 1;  // tanh after every conv but the last (model.py:171-173)
        a.out[0] = ConvOut{dst, (long long)Tmax * dst_c, 0, dst_c, dst_c, ACC_STORE};
        if (last && add_input) {  // model.py:717: par_out + postnet(par_out)
            a.resid = X0;
            a.resid_batch_stride = (long long)Tmax * melp;
        }
        if ((rc = launch_layer(h->convs[i], &sb, 1, a, stream, prof))) return rc;
        cur = dst;
        cur_c = dst_c;
    }
    dim3 grid((Tmax + 31) / 32, (D + 31) / 32, B);
    hipLaunchKernelGGL(from_channels_last_kernel, grid, dim3(256), 0, stream, (const float*)X1, melp, D, Tmax, lengths, y);
    MV_HIP(hipGetLastError());
    return MV_OK;
}

}  // namespace mv

using namespace mv;

extern "C" {

int mv_postnet_create(const mv_postnet_cfg* cfg, const mv_tensor* weights, int n_weights, int device, int precision,
                      const mv_options* options, mv_handle* out) {
    if (!cfg || !weights || !out || n_weights <= 0) {
        set_error("mv_postnet_create: null argument");
        return MV_EINVAL;
    }
    *out = nullptr;
    if (precision != MV_PREC_FP32 && precision != MV_PREC_BF16 && precision != MV_PREC_BF16X3 && precision != MV_PREC_F16) {
        set_error("mv_postnet_create: unknown precision %d", precision);
        return MV_EINVAL;
    }
    if (cfg->dim_data < 1 || cfg->embedding_dim < 1 || cfg->n_convolutions < 2 || cfg->n_convolutions > 16 ||
        cfg->kernel_size < 1 || cfg->kernel_size % 2 != 1 || !(cfg->bn_eps > 0.f)) {
        set_error("mv_postnet_create: bad config");
        return MV_EINVAL;
    }
    mv_options opt;
    int rc = resolve_options(options, &opt);
    if (rc) return rc;
    DeviceGuard g(device);
    if (!g.ok) {
        set_error("mv_postnet_create: cannot select HIP device %d", device);
        return MV_EHIP;
    }
    rc = init_kernel_attributes();
    if (rc) return rc;
    WeightMap W;
    if ((rc = W.load(weights, n_weights))) return rc;
    PostnetHandle* h = new PostnetHandle();
    h->kind = 3;
    h->device = device;
    h->precision = precision;
    h->opt = opt;
    h->cfg = *cfg;
    if ((rc = postnet_build(h, W))) {
        delete h;
        return rc;
    }
    *out = h;
    return MV_OK;
}

int mv_postnet_forward(mv_handle hh, const float* x, const int32_t* lengths, int B, int Tmax, float* y, int add_input,
                       void* workspace, size_t workspace_bytes, void* stream) {
    if (!hh || hh->kind != 3) {
        set_error("mv_postnet_forward: not a Postnet handle");
        return MV_EINVAL;
    }
    if (!x || !y || !workspace || B < 1 || Tmax < 1) {
        set_error("mv_postnet_forward: bad argument (B=%d, Tmax=%d)", B, Tmax);
        return MV_EINVAL;
    }
    DeviceGuard g(hh->device);
    if (!g.ok) {
        set_error("mv_postnet_forward: cannot select HIP device %d", hh->device);
        return MV_EHIP;
    }
    return postnet_forward(static_cast<PostnetHandle*>(hh), x, lengths, B, Tmax, y, add_input, workspace,
                           workspace_bytes, (hipStream_t)stream);
}

}  // extern "C"
