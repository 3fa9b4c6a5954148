// gate_gemm.hip -- launch side of gate_gemm.h (WaveGlow's gated in-layer conv and residual 1x1 conv as ring GEMMs).
#define MV_GATE_GEMM_KERNEL
#include "common.h"
#include "gate_gemm.h"

namespace mv {

int init_gate_gemm_attributes() {
    MV_HIP(hipFuncSetAttribute((const void*)gate_gemm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kGgLds));
    MV_HIP(hipFuncSetAttribute((const void*)res_gemm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, GgGeom<4>::Lds));
    MV_HIP(hipFuncSetAttribute((const void*)res_gemm256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kGgLds));
    return MV_OK;
}

// Shapes the kernel assumes (checked here, on the host, before anything is launched).
bool gate_gemm_supported(int C, int Cp, int ncond, int ktaps) {
    return C == Cp && C % 128 == 0 && ncond % kGgBK == 0 && Cp % kGgBK == 0 && ktaps % 2 == 1 && ktaps >= 1 &&
           ktaps * (Cp / kGgBK) + ncond / kGgBK >= 3;  // (the prologue fills a ring of five half-stages)
}

static inline uint16_t f2bf(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

int build_gate_gemm_weights(const std::vector<SrcSpec>& srcs, int M, unsigned short** dev) {
    size_t K = 0;
    for (const SrcSpec& s : srcs) K += (size_t)s.ktaps * s.C;
    std::vector<unsigned short> w((size_t)M * K);
    for (int m = 0; m < M; ++m) {
        unsigned short* row = &w[(size_t)m * K];
        size_t k = 0;
        for (const SrcSpec& s : srcs)
            for (int tap = 0; tap < s.ktaps; ++tap)
                for (int c = 0; c < s.C; ++c) row[k++] = f2bf(s.w(m, tap, c));
    }
    return upload(w, dev);
}

static bool aligned16(const void* a, const void* b, const void* c, const void* d, const void* e) {
    return (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d | (uintptr_t)e) & 15) == 0;
}

int launch_gate_gemm(const GateGemmParams& p, int B, hipStream_t stream) {
    const long long row_bytes = 2ll * (p.h_stride > p.ncond ? p.h_stride : p.ncond);
    if (p.K != p.ktaps * p.Cp + p.ncond || p.nx < 1 || p.ny < 1 || B < 1 || p.h_stride < p.Cp || p.ny * 128 != p.Cp ||
        p.Cp % kGgBK != 0 || p.ncond % kGgBK != 0 || p.h_stride % 8 != 0 || !aligned16(p.w, p.h, p.cond, p.out, p.zero) ||
        (long long)p.ny * kGgTile * p.K * 2 > 0xffffffffll || p.L * row_bytes > 0xffffffffll ||
        (long long)p.nx * kGgTile < p.L || !p.zero || (p.fold_w && (!p.fold_out || p.fold_nout < 1 || p.fold_nout > 8))) {
        set_error("gate_gemm: inconsistent launch (K=%d, nx=%d, ny=%d)", p.K, p.nx, p.ny);
        return MV_EINVAL;
    }
    const unsigned blocks = (unsigned)((p.nx + 7) / 8) * 8u * (unsigned)p.ny;
    hipLaunchKernelGGL(gate_gemm_kernel, dim3(blocks, (unsigned)B), dim3(512), kGgLds, stream, p);
    MV_HIP(hipGetLastError());
    return MV_OK;
}

// The residual 1x1 conv on the same kernel body: M = C rows in blocks of `bm` = 128 or 256, K = Cp, one input (the gated
// activations).
int launch_res_gemm(const GateGemmParams& p, int B, int bm, hipStream_t stream) {
    if ((bm != 128 && bm != 256) || p.ktaps != 1 || p.ncond != 0 || p.K != p.Cp || p.Cp % kGgBK != 0 ||
        p.Cp / kGgBK < 3 || p.nx < 1 || p.ny < 1 || B < 1 || p.h_stride < p.Cp || p.h_stride % 8 != 0 ||
        p.res_stride < p.ny * bm || p.res_stride % 8 != 0 || !p.res_h || !p.res_hb ||
        !aligned16(p.w, p.h, p.res_h, p.res_hb, p.zero) || !p.zero || (long long)p.ny * bm * p.K * 2 > 0xffffffffll ||
        p.L * 2ll * p.h_stride > 0xffffffffll || (long long)p.nx * kGgTile < p.L) {
        set_error("res_gemm: inconsistent launch (K=%d, Cp=%d, nx=%d, ny=%d, bm=%d)", p.K, p.Cp, p.nx, p.ny, bm);
        return MV_EINVAL;
    }
    const unsigned blocks = (unsigned)((p.nx + 7) / 8) * 8u * (unsigned)p.ny;
    if (bm == 128)
        hipLaunchKernelGGL(res_gemm_kernel, dim3(blocks, (unsigned)B), dim3(512), GgGeom<4>::Lds, stream, p);
    else
        hipLaunchKernelGGL(res_gemm256_kernel, dim3(blocks, (unsigned)B), dim3(512), kGgLds, stream, p);
    MV_HIP(hipGetLastError());
    return MV_OK;
}

}  // namespace mv

#ifdef MV_GG_STAMP
extern "C" __attribute__((visibility("default"))) int mv_debug_gate_gemm_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mv::gg_stamps), sizeof(unsigned long long) * 4096 * 8);
}
#endif
