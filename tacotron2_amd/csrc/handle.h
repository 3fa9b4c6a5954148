// handle.h -- what an mv_handle points at.
#pragma once
#include "common.h"

struct mv_handle_s {
    int kind = 0;  // 1 = HiFi-GAN generator, 2 = WaveGlow, 3 = Tacotron2 Postnet
    int device = 0;
    int precision = MV_PREC_FP32;
    mv_options opt = {2, 1, 1, 1};  // see mv_default_options
    mv::Profiler prof;
    virtual ~mv_handle_s() { prof.release(); }
    virtual size_t workspace_bytes(int B, int Tmax) const = 0;
    virtual int hop() const = 0;
    virtual double macs_per_frame() const = 0;
};

namespace mv {

// options of a create call: the defaults, overridden by what the caller passed; MV_EINVAL on a value out of range
int resolve_options(const mv_options* in, mv_options* out);

// small kernels shared by both vocoders (small_kernels.hip)
int launch_to_channels_last(const float* x, int B, int C, int T, float* y, int Cp, hipStream_t s);
int launch_pcm16(const float* wav, size_t n, int16_t* pcm, hipStream_t s);

// RAII-free device guard: remembers and restores the current device
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) ok = false;
        if (ok && prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
        target = dev;
    }
    ~DeviceGuard() {
        if (prev >= 0 && prev != target) (void)hipSetDevice(prev);
    }
    int target = 0;
};

}  // namespace mv
