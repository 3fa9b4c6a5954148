// layers.hip -- weight repacking and launch of conv_cl_kernel (see conv_mfma.h).
#include <cmath>
#include <cstdlib>

#include "common.h"

namespace mv {

static thread_local std::string g_err;

void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}
const char* last_error_cstr() { return g_err.c_str(); }

//                              TM TN WM WN      MT   NT
const TileCfg kTileCfgs[] = {
    {4, 4, 4, 1},  // 0:  256 x  64
    {4, 8, 4, 1},  // 1:  256 x 128
    {4, 4, 2, 2},  // 2:  128 x 128
    {4, 4, 1, 4},  // 3:   64 x 256
    {2, 4, 1, 4},  // 4:   32 x 256
    {2, 8, 1, 4},  // 5:   32 x 512
    {4, 8, 1, 4},  // 6:   64 x 512
    {4, 8, 2, 2},  // 7:  128 x 256
    {4, 8, 4, 2},  // 8:  256 x 256, 8 waves (one block per CU)
};
const int kNumTileCfgs = sizeof(kTileCfgs) / sizeof(kTileCfgs[0]);

// tile shape by output-channel count (measured best of kTileCfgs per class on the two vocoders' layers)
static int default_cfg(int M) {
    if (M % 256 == 0) return 0;
    if (M % 128 == 0) return 2;
    if (M % 64 == 0) return 3;
    return 4;
}

static inline uint16_t f2bf(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// float32 -> float16 bits, round to nearest even, saturating at +-65504 (the kernels' own conversions saturate too)
static inline uint16_t f2h(float x) {
    if (x != x) return 0x7e00;
    x = x > 65504.f ? 65504.f : (x < -65504.f ? -65504.f : x);
    const _Float16 h = (_Float16)x;
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
}

int build_layer(Layer* L, const std::vector<SrcSpec>& srcs, int M, const std::vector<float>& bias, int prec, int epi,
                int force_cfg, bool one_pass) {
    const bool bf16 = prec != MV_PREC_FP32, x3 = prec == MV_PREC_BF16X3, f16 = prec == MV_PREC_F16;
    const int CH = bf16 ? 32 : 16;
    const int ES = bf16 && !x3 ? 2 : 4;  // LDS bytes per staged element (x3: a bf16 pair)
    const int KV = x3 ? 3 : 1;           // K steps per real k-chunk (x3: hi*hi, hi*lo, lo*hi)
    if (M % 32 != 0 || (int)bias.size() != M) {
        set_error("build_layer: M=%d must be a multiple of 32 with a bias of the same size", M);
        return MV_EINVAL;
    }
    L->M = M;
    L->bf16 = bf16;
    L->x3 = x3;
    L->opf = f16 ? OP_F16 : OP_BF16;
    L->epi = epi;
    L->cfg = force_cfg >= 0 ? force_cfg : default_cfg(M);
    const TileCfg& tc = kTileCfgs[L->cfg];
    if (M % tc.MT() != 0 || (epi == EPI_GATE && tc.TM % 2 != 0)) {
        set_error("build_layer: tile config %d does not fit M=%d", L->cfg, M);
        return MV_EINVAL;
    }
    // 4-wave tiles: two blocks per CU; the 8-wave tile has the CU to itself
    const bool big = tc.WM * tc.WN >= 8;
#ifndef MV_LDS_KB_FP32
#define MV_LDS_KB_FP32 50
#endif
    const size_t lds_budget = (size_t)(bf16 ? (big ? 128 : 72) : (big ? 128 : MV_LDS_KB_FP32)) * 1024;  // 4-wave tiles: two blocks per CU (2 x 72 KB of 160)
    L->npass = 0;
    L->lds_bytes = 0;
    L->steps_total = 0;
    L->macs_per_n = 0;
    for (size_t si = 0; si < srcs.size(); ++si) {
        const SrcSpec& s = srcs[si];
        if (s.C % CH != 0) {
            set_error("build_layer: source %zu has %d channels, not a multiple of %d", si, s.C, CH);
            return MV_EINVAL;
        }
        // taps further apart than half a tile share few rows: stage each tap on its own
        const bool per_tap = !one_pass && s.ktaps > 1 && 2 * s.dil >= tc.NT();
        const int groups = per_tap ? s.ktaps : 1;
        const int gtaps = per_tap ? 1 : s.ktaps;
        const int rows = tc.NT() + (gtaps - 1) * s.dil;
        // split the channels until one staged tile fits the LDS budget (cb stays a multiple of CH)
        int nsplit = 1;
        while (true) {
            const int cb = round_up((s.C + nsplit - 1) / nsplit, CH);
            if (one_pass || (size_t)rows * (cb * ES + kLdsRowPad) <= lds_budget || cb == CH) break;
            ++nsplit;
        }
        const int cb0 = round_up((s.C + nsplit - 1) / nsplit, CH);
        for (int g = 0; g < groups; ++g)
            for (int off = 0; off < s.C; off += cb0) {
                if (L->npass >= kMaxPass) {
                    set_error("build_layer: more than %d passes needed", kMaxPass);
                    return MV_EINVAL;
                }
                LayerPass& p = L->pass[L->npass++];
                p.src = (int)si;
                p.cin_off = off;
                p.cb = std::min(cb0, s.C - off);
                p.ktaps = gtaps;
                p.dil = s.dil;
                p.tap0 = g;
                p.pad_left = s.pad_left - g * s.dil;
                L->steps_total += p.ktaps * (p.cb / CH) * KV;
                L->lds_bytes = std::max(L->lds_bytes, (size_t)rows * (p.cb * ES + kLdsRowPad));
            }
        L->macs_per_n += (double)M * s.C * s.ktaps;
    }
    // room for the coalescing epilogue's float16 [NT][MT] tile (conv_cl_kernel, epi_lds)
    if (bf16 && !x3) L->lds_bytes = std::max(L->lds_bytes, (size_t)tc.NT() * (tc.MT() * (epi == EPI_GATE ? 1 : 2) + 16));
    if (one_pass) L->lds_bytes = 0;
    if (L->lds_bytes > 160 * 1024) {
        set_error("build_layer: needs %zu bytes of LDS", L->lds_bytes);
        return MV_EINVAL;
    }
    // pack: [mtile][pass][tap][kc][lane][16 B]
    const int mtiles = M / 16;
    const size_t units = (size_t)mtiles * L->steps_total * 64;
    std::vector<uint32_t> packed(units * 4);
    const int per_lane = CH / 4;  // 8 bf16 or 4 f32
    for (int mt = 0; mt < mtiles; ++mt) {
        size_t step = 0;
        for (int pi = 0; pi < L->npass; ++pi) {
            const LayerPass& p = L->pass[pi];
            const SrcSpec& s = srcs[p.src];
            const int nk = p.cb / CH;
            for (int tap = 0; tap < p.ktaps; ++tap)
                for (int kv = 0; kv < nk * KV; ++kv, ++step)
                    for (int lane = 0; lane < 64; ++lane) {
                        // x3: steps [0, nk) and [nk, 2nk) carry w_hi (against x_hi, then x_lo), [2nk, 3nk) w_lo
                        const int kc = kv % nk;
                        const bool want_lo = x3 && kv >= 2 * nk;
                        const int m = mt * 16 + (lane & 15);
                        const int c0 = p.cin_off + kc * CH + (lane >> 4) * per_lane;
                        uint32_t* dst = &packed[(((size_t)mt * L->steps_total + step) * 64 + lane) * 4];
                        if (bf16) {
                            auto part = [&](float w) -> uint32_t {
                                if (f16) return f2h(w);
                                const uint16_t h = f2bf(w);
                                if (!want_lo) return h;
                                uint32_t u = (uint32_t)h << 16;
                                float hf;
                                memcpy(&hf, &u, 4);
                                return f2bf(w - hf);
                            };
                            for (int j = 0; j < 4; ++j) {
                                const uint32_t lo = part(s.w(m, p.tap0 + tap, c0 + 2 * j));
                                const uint32_t hi = part(s.w(m, p.tap0 + tap, c0 + 2 * j + 1));
                                dst[j] = lo | (hi << 16);
                            }
                        } else {
                            for (int j = 0; j < 4; ++j) {
                                const float f = s.w(m, p.tap0 + tap, c0 + j);
                                memcpy(&dst[j], &f, 4);
                            }
                        }
                    }
        }
    }
    uint32_t* dw = nullptr;
    int rc = upload(packed, &dw);
    if (rc) return rc;
    L->w = dw;
    return upload(bias, &L->bias);
}

void free_layer(Layer* L) {
    if (L->w) (void)hipFree(L->w);
    if (L->bias) (void)hipFree(L->bias);
    L->w = nullptr;
    L->bias = nullptr;
}

// ---- kernel table ---------------------------------------------------------------------------
typedef void (*ConvKernel)(const ConvParams);

template <bool BF16, int EPI, bool X3, int OPF = OP_BF16>
static ConvKernel kernel_for(int cfg) {
    switch (cfg) {
        case 0: return conv_cl_kernel<BF16, 4, 4, 4, 1, EPI, X3, OPF>;
        case 1: return conv_cl_kernel<BF16, 4, 8, 4, 1, EPI, X3, OPF>;
        case 2: return conv_cl_kernel<BF16, 4, 4, 2, 2, EPI, X3, OPF>;
        case 3: return conv_cl_kernel<BF16, 4, 4, 1, 4, EPI, X3, OPF>;
        case 4: return conv_cl_kernel<BF16, 2, 4, 1, 4, EPI, X3, OPF>;
        case 5: return conv_cl_kernel<BF16, 2, 8, 1, 4, EPI, X3, OPF>;
        case 6: return conv_cl_kernel<BF16, 4, 8, 1, 4, EPI, X3, OPF>;
        case 7: return conv_cl_kernel<BF16, 4, 8, 2, 2, EPI, X3, OPF>;
        case 8: return conv_cl_kernel<BF16, 4, 8, 4, 2, EPI, X3, OPF>;
    }
    return nullptr;
}

// prec: MV_PREC_*
static ConvKernel pick_kernel(int prec, int epi, int cfg) {
    if (prec == MV_PREC_BF16X3)
        return epi == EPI_GATE ? kernel_for<true, EPI_GATE, true>(cfg) : kernel_for<true, EPI_LINEAR, true>(cfg);
    if (prec == MV_PREC_BF16)
        return epi == EPI_GATE ? kernel_for<true, EPI_GATE, false>(cfg) : kernel_for<true, EPI_LINEAR, false>(cfg);
    if (prec == MV_PREC_F16)
        return epi == EPI_GATE ? kernel_for<true, EPI_GATE, false, OP_F16>(cfg)
                               : kernel_for<true, EPI_LINEAR, false, OP_F16>(cfg);
    return epi == EPI_GATE ? kernel_for<false, EPI_GATE, false>(cfg) : kernel_for<false, EPI_LINEAR, false>(cfg);
}

int init_kernel_attributes() {
    // per device: allow the full 160 KiB of dynamic LDS
    for (int prec = 0; prec < 4; ++prec)
        for (int epi = 0; epi < 2; ++epi)
            for (int c = 0; c < kNumTileCfgs; ++c) {
                ConvKernel k = pick_kernel(prec, epi, c);
                MV_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            }
    return MV_OK;
}

hipEvent_t Profiler::get() {
    if (!pool.empty()) {
        hipEvent_t e = pool.back();
        pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
void Profiler::begin(hipStream_t s, const std::string& name, double macs_per_frame, const std::string& detail) {
    if (!on) return;
    Rec r{name, detail, macs_per_frame, get(), get()};
    (void)hipEventRecord(r.e0, s);
    recs.push_back(r);
}
void Profiler::end(hipStream_t s) {
    if (!on || recs.empty()) return;
    (void)hipEventRecord(recs.back().e1, s);
}
void Profiler::release() {
    for (auto& r : recs) {
        pool.push_back(r.e0);
        pool.push_back(r.e1);
    }
    recs.clear();
    for (hipEvent_t e : pool)
        if (e) (void)hipEventDestroy(e);
    pool.clear();
    on = false;
}

std::string layer_kernel_name(const Layer& L) {
    const TileCfg& tc = kTileCfgs[L.cfg];
    char buf[128];
    // the symbol as rocprofv3 prints it
    snprintf(buf, sizeof buf, "void mv::conv_cl_kernel<%s, %d, %d, %d, %d, %d, %s, %d>(mv::ConvParams)",
             L.bf16 ? "true" : "false", tc.TM, tc.TN, tc.WM, tc.WN, L.epi, L.x3 ? "true" : "false", L.opf);
    return buf;
}

int launch_layer(const Layer& L, const SrcBind* srcs, int nsrc, const LaunchArgs& a, hipStream_t stream,
                 Profiler* prof) {
    const TileCfg& tc = kTileCfgs[L.cfg];
    ConvParams p;
    memset(&p, 0, sizeof p);
    p.npass = L.npass;
    for (int i = 0; i < L.npass; ++i) {
        const LayerPass& lp = L.pass[i];
        if (lp.src >= nsrc) {
            set_error("launch_layer: pass %d wants source %d of %d", i, lp.src, nsrc);
            return MV_EINVAL;
        }
        const SrcBind& s = srcs[lp.src];
        if (lp.cin_off + lp.cb > s.row_stride) {
            set_error("launch_layer: source %d rows hold %d channels, pass needs %d", lp.src, s.row_stride,
                      lp.cin_off + lp.cb);
            return MV_EINVAL;
        }
        ConvPass& cp = p.pass[i];
        cp.src = s.ptr;
        cp.batch_stride = s.batch_stride;
        cp.row_stride = s.row_stride;
        cp.cin_off = lp.cin_off;
        cp.cb = lp.cb;
        cp.ktaps = lp.ktaps;
        cp.dil = lp.dil;
        cp.pad_left = lp.pad_left;
        cp.rows_per_frame = s.rows_per_frame;
        cp.slope = s.slope;
        cp.src_f16 = s.f16;
    }
    p.w = L.w;
    p.steps_total = L.steps_total;
    p.bias = L.bias;
    p.lengths = a.lengths;
    p.Tmax = a.Tmax;
    p.n_rows_per_frame = a.n_rows_per_frame;
    p.n_extra = a.n_extra;
    p.split = a.split;
    p.out[0] = a.out[0];
    p.out[1] = a.out[1];
    p.resid = a.resid;
    p.resid_batch_stride = a.resid_batch_stride;
    p.div = a.div;
    p.act = a.act;
    // the coalescing epilogue (conv_cl_kernel) wants 8-element units that never straddle a validity bound, a
    // block's M range on one side of the split, and a residual typed and laid out like out[0]
    const int mo = L.epi == EPI_GATE ? tc.MT() / 2 : tc.MT();
    const bool two = a.split < L.M;
    auto unit_ok = [&](const ConvOut& o) {
        return o.row_stride % 8 == 0 && o.off % 8 == 0 && o.valid_per_frame % 8 == 0 && o.batch_stride % 8 == 0;
    };
    p.epi_lds = L.bf16 && !L.x3 && unit_ok(a.out[0]) && (!two || (a.split % mo == 0 && unit_ok(a.out[1]))) &&
                (L.epi == EPI_LINEAR || !two) && a.resid_batch_stride % 8 == 0;
    // bfloat16 storage (dtype code 2) is for tensors that are only ever MFMA operands: a plain copy on the way in,
    // a plain store on the way out
    for (int i = 0; i < L.npass; ++i) {
        const SrcBind& sb = srcs[L.pass[i].src];
        if (sb.f16 == 2 && !(L.bf16 && sb.slope == 1.0f)) {
            set_error("launch_layer: a bfloat16 source needs a bf16 layer without input activation");
            return MV_EINVAL;
        }
    }
    if (a.out[0].f16 == 2 && !(!two && !a.resid && a.out[0].mode == ACC_STORE)) {
        set_error("launch_layer: a bfloat16 output must be a plain store (single output, no residual, no accumulate)");
        return MV_EINVAL;
    }
    const long long n_max = (long long)a.Tmax * a.n_rows_per_frame + a.n_extra;
    if (n_max <= 0 || a.B <= 0) return MV_OK;
    const unsigned nx = (unsigned)((n_max + tc.NT() - 1) / tc.NT()), ny = (unsigned)(L.M / tc.MT());
    dim3 grid(nx, ny, (unsigned)a.B);
    // M-blocks of a row tile on one XCD (conv_cl_kernel) -- when the activations are the big operand.  The price is
    // that every XCD's L2 then cycles through all ny weight slices instead of sitting on one, so layers with few
    // rows and many output channels (conv_pre, the first upsampler) keep the natural order.
    double act_row_bytes = 0;
    for (int i = 0; i < L.npass; ++i)
        if (L.pass[i].tap0 == 0) act_row_bytes += (double)L.pass[i].cb * (L.bf16 ? 2 : 4);
    const double act_bytes = (double)a.B * (double)n_max * act_row_bytes;
    const double w_bytes = (double)L.M * L.steps_total * (L.bf16 ? 64 : 64);  // 16 rows x 1 KiB per (M tile, K step)
    if (ny > 1 && act_bytes >= 8.0 * w_bytes) {
        p.xcd_ny = (int)ny;
        p.nx = (int)nx;
        grid = dim3((nx + 7) / 8 * 8 * ny, 1, (unsigned)a.B);
    }
    dim3 block(tc.WM * tc.WN * 64);
    ConvKernel k = pick_kernel(L.x3 ? MV_PREC_BF16X3 : L.opf == OP_F16 ? MV_PREC_F16 : L.bf16 ? MV_PREC_BF16 : MV_PREC_FP32,
                               L.epi, L.cfg);
    if (prof) prof->begin(stream, layer_kernel_name(L), L.alg_macs_per_n * a.n_rows_per_frame);
    hipLaunchKernelGGL(k, grid, block, L.lds_bytes, stream, p);
    if (prof) prof->end(stream);
    MV_HIP(hipGetLastError());
    return MV_OK;
}

// ---- WeightMap ------------------------------------------------------------------------------
int WeightMap::load(const mv_tensor* t, int n) {
    for (int i = 0; i < n; ++i) {
        if (!t[i].name || !t[i].data || t[i].ndim < 1 || t[i].ndim > 4) {
            set_error("weights[%d]: bad tensor descriptor", i);
            return MV_EINVAL;
        }
        HostTensor h;
        size_t cnt = 1;
        for (int d = 0; d < t[i].ndim; ++d) {
            if (t[i].shape[d] <= 0) {
                set_error("weights[%d] (%s): bad shape", i, t[i].name);
                return MV_EINVAL;
            }
            h.shape.push_back(t[i].shape[d]);
            cnt *= (size_t)t[i].shape[d];
        }
        h.data.assign(t[i].data, t[i].data + cnt);
        m_[t[i].name] = std::move(h);
    }
    return MV_OK;
}

const HostTensor* WeightMap::get(const std::string& name) {
    auto it = m_.find(name);
    if (it != m_.end()) return &it->second;
    const std::string suffix = ".weight";
    if (name.size() > suffix.size() && name.compare(name.size() - suffix.size(), suffix.size(), suffix) == 0) {
        auto g = m_.find(name + "_g"), v = m_.find(name + "_v");
        if (g != m_.end() && v != m_.end()) {
            // w = g * v / ||v||, norm over all dims but 0 (torch weight_norm, dim=0)
            HostTensor w = v->second;
            const size_t d0 = (size_t)w.shape[0];
            const size_t inner = w.data.size() / d0;
            if (g->second.data.size() != d0) {
                set_error("%s_g has %zu entries, expected %zu", name.c_str(), g->second.data.size(), d0);
                return nullptr;
            }
            for (size_t i = 0; i < d0; ++i) {
                double ss = 0;
                for (size_t j = 0; j < inner; ++j) ss += (double)w.data[i * inner + j] * w.data[i * inner + j];
                const float scale = g->second.data[i] / (float)std::sqrt(ss);
                for (size_t j = 0; j < inner; ++j) w.data[i * inner + j] *= scale;
            }
            auto ins = m_.emplace(name, std::move(w));
            return &ins.first->second;
        }
    }
    set_error("missing tensor '%s'", name.c_str());
    return nullptr;
}

}  // namespace mv
