// api.hip -- handle-independent entry points of libmelvoc and the single-layer test hooks.
#include "handle.h"

namespace mv {
const char* last_error_cstr();
int build_convT_layer(Layer* L, const HostTensor* w, const HostTensor* b, int Cin, int Cout, int K, int u, int cin_pad,
                      int cout_pad, int prec, int force_cfg);
int launch_convT(const Layer& L, const void* x, long long x_bs, int cin_stride, int rows_per_frame, float slope,
                 void* y, long long y_bs, int cout_pad, int K, int u, int pad, const int32_t* lengths, int B,
                 int Tmax, hipStream_t stream, Profiler* prof, int in_f16, int out_f16);
}  // namespace mv

namespace mv {
int resolve_options(const mv_options* in, mv_options* out) {
    mv_default_options(out);
    if (!in) return MV_OK;
    if (in->fuse < 0 || in->fuse > 2 || in->io16 < 0 || in->io16 > 1 || in->wg_fold < 0 || in->wg_fold > 1 ||
        in->wg_gemm < 0 || in->wg_gemm > 3) {
        set_error("mv_options: fuse must be 0..2, wg_gemm 0..3, io16 and wg_fold 0 or 1 (got %d, %d, %d, %d)", in->fuse,
                  in->io16, in->wg_fold, in->wg_gemm);
        return MV_EINVAL;
    }
    *out = *in;
    return MV_OK;
}
}  // namespace mv

using namespace mv;

extern "C" {

int mv_backend(void) { return 1; }

void mv_default_options(mv_options* opt) {
    if (!opt) return;
    opt->fuse = 2;
    opt->io16 = 1;
    opt->wg_fold = 1;
    opt->wg_gemm = 1;
}

const char* mv_last_error(void) { return last_error_cstr(); }

size_t mv_workspace_bytes(mv_handle h, int B, int Tmax) {
    if (!h || B < 1 || Tmax < 1) return 0;
    return h->workspace_bytes(B, Tmax);
}

int mv_hop(mv_handle h) { return h ? h->hop() : 0; }

double mv_macs_per_frame(mv_handle h) { return h ? h->macs_per_frame() : 0.0; }

void mv_destroy(mv_handle h) {
    if (!h) return;
    DeviceGuard g(h->device);
    delete h;
}

int mv_pcm16(const float* wav, size_t n, int16_t* pcm, void* stream) {
    if (!wav || !pcm) {
        set_error("mv_pcm16: null pointer");
        return MV_EINVAL;
    }
    return launch_pcm16(wav, n, pcm, (hipStream_t)stream);
}

int mv_profile_begin(mv_handle h) {
    if (!h) {
        set_error("mv_profile_begin: null handle");
        return MV_EINVAL;
    }
    DeviceGuard g(h->device);
    h->prof.release();
    h->prof.on = true;
    return MV_OK;
}

int mv_profile_end(mv_handle h, mv_profile_row* rows, int max_rows, int* n_rows) {
    if (!h || !rows || !n_rows || max_rows < 1) {
        set_error("mv_profile_end: bad argument");
        return MV_EINVAL;
    }
    DeviceGuard g(h->device);
    std::map<std::string, mv_profile_row> acc;
    for (auto& r : h->prof.recs) {
        MV_HIP(hipEventSynchronize(r.e1));
        float ms = 0.f;
        MV_HIP(hipEventElapsedTime(&ms, r.e0, r.e1));
        mv_profile_row& row = acc[r.name + "|" + r.detail];
        if (row.launches == 0) {
            memset(&row, 0, sizeof row);
            snprintf(row.name, sizeof row.name, "%s", r.name.c_str());
            snprintf(row.detail, sizeof row.detail, "%s", r.detail.c_str());
        }
        row.launches += 1;
        row.ms += ms;
        row.macs_per_frame += r.macs_per_frame;
    }
    int n = 0;
    for (auto& kv : acc)
        if (n < max_rows) rows[n++] = kv.second;
    *n_rows = n;
    h->prof.release();
    return MV_OK;
}

int mv_test_conv1d(const float* x, const int32_t* lengths, int B, int L, int Cin, const float* w, const float* bias,
                   int Cout, int K, int dilation, float pre_slope, int precision, float* y, void* stream) {
    if (!x || !w || !y || B < 1 || L < 1 || Cin % 32 || Cout % 32 || K < 1 || K % 2 != 1 || dilation < 1) {
        set_error("mv_test_conv1d: needs Cin, Cout multiples of 32 and an odd K");
        return MV_EINVAL;
    }
    int rc = init_kernel_attributes();
    if (rc) return rc;
    HostTensor hw, hb;
    hw.data.assign(w, w + (size_t)Cout * Cin * K);
    if (bias) hb.data.assign(bias, bias + Cout);
    Layer Lr;
    rc = build_conv1d_layer(&Lr, &hw, bias ? &hb : nullptr, Cin, Cout, K, dilation, Cin, Cout, precision);
    if (rc) return rc;
    SrcBind sb{x, (long long)L * Cin, Cin, 1, pre_slope};
    LaunchArgs a;
    a.lengths = lengths;
    a.B = B;
    a.Tmax = L;
    a.out[0] = ConvOut{y, (long long)L * Cout, 0, Cout, Cout, ACC_STORE};
    rc = launch_layer(Lr, &sb, 1, a, (hipStream_t)stream);
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    free_layer(&Lr);
    if (rc) return rc;
    MV_HIP(e);
    return MV_OK;
}

int mv_test_conv_transpose1d(const float* x, const int32_t* lengths, int B, int L, int Cin, const float* w,
                             const float* bias, int Cout, int K, int stride, int padding, float pre_slope,
                             int precision, float* y, void* stream) {
    if (!x || !w || !y || B < 1 || L < 1 || Cin % 32 || Cout % 32 || stride < 1 || K < stride ||
        (K - stride) != 2 * padding) {
        set_error("mv_test_conv_transpose1d: needs Cin, Cout multiples of 32 and K - stride == 2*padding");
        return MV_EINVAL;
    }
    int rc = init_kernel_attributes();
    if (rc) return rc;
    HostTensor hw, hb;
    hw.data.assign(w, w + (size_t)Cout * Cin * K);
    if (bias) hb.data.assign(bias, bias + Cout);
    Layer Lr;
    rc = build_convT_layer(&Lr, &hw, bias ? &hb : nullptr, Cin, Cout, K, stride, Cin, Cout, precision, -1);
    if (rc) return rc;
    rc = launch_convT(Lr, x, (long long)L * Cin, Cin, 1, pre_slope, y, (long long)L * stride * Cout, Cout, K, stride,
                      padding, lengths, B, L, (hipStream_t)stream, nullptr, 0, 0);
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    free_layer(&Lr);
    if (rc) return rc;
    MV_HIP(e);
    return MV_OK;
}

}  // extern "C"
