// small_kernels.hip -- layout and sample-format kernels (HBM-bound, no MFMA).
#include "handle.h"

namespace mv {

// [B][C][T] (the reference's channel-first mel) -> [B][T][Cp] channels-last, channels >= C zeroed.
__global__ __launch_bounds__(256) void to_channels_last_kernel(const float* __restrict__ x, int C, int T,
                                                               float* __restrict__ y, int Cp) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, t = t0 + tx;
        tile[i][tx] = (c < C && t < T) ? x[((long long)b * C + c) * T + t] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int t = t0 + i, c = c0 + tx;
        if (t < T && c < Cp) y[((long long)b * T + t) * Cp + c] = tile[tx][i];
    }
}

int launch_to_channels_last(const float* x, int B, int C, int T, float* y, int Cp, hipStream_t s) {
    dim3 grid((T + 31) / 32, (Cp + 31) / 32, B);
    hipLaunchKernelGGL(to_channels_last_kernel, grid, dim3(256), 0, s, x, C, T, y, Cp);
    MV_HIP(hipGetLastError());
    return MV_OK;
}

// do_syn.py:11,389-394: 32768*y, then numpy astype('int16') = truncation toward zero, no clip,
// wrap-around on overflow.
__global__ __launch_bounds__(256) void pcm16_kernel(const float* __restrict__ wav, size_t n, int16_t* __restrict__ pcm) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) pcm[i] = pcm16_of(wav[i]);
}

int launch_pcm16(const float* wav, size_t n, int16_t* pcm, hipStream_t s) {
    if (n == 0) return MV_OK;
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(pcm16_kernel, dim3(blocks), dim3(256), 0, s, wav, n, pcm);
    MV_HIP(hipGetLastError());
    return MV_OK;
}

}  // namespace mv
