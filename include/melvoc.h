/*
 * melvoc.h -- C ABI of libmelvoc.so: MI355X (gfx950) mel-spectrogram -> waveform vocoding.
 *
 * The reference (gerard-bailly/tacotron2) has no FFI: its vocoder path is two Python
 * nn.Module forward functions called by duck typing.  Each entry point below names the
 * reference interface it stands in for (paths relative to the reference repository); the
 * Python classes in tacotron2_amd/ keep those call signatures and bind these symbols with
 * ctypes.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / C++ types cross this boundary.
 *   - every function returns 0 or a negative MV_E* code; mv_last_error() gives the message of
 *     the calling thread's last failure.  Kernels never abort().
 *   - `*_forward/_infer` are asynchronous on the given hipStream_t (void* stream; NULL = the
 *     default stream) and perform no allocation, no host<->device copy and no synchronisation.
 *     The caller owns mel, lengths, noise, output and workspace (device memory).
 *   - a handle owns only its repacked weights; it is bound to the device it was created on and
 *     is not re-entrant (callers serialise calls on one handle; distinct handles are independent).
 *     Process-global state: the per-thread error string, and ONE exception to "distinct handles are
 *     independent" -- the static scratch of mv_hifigan_forward_static, a MV_STATIC_SCRATCH_BYTES device
 *     array that is part of the library's code object (every GPU a process runs the library on holds a
 *     copy from load to exit, used or not) and is shared by every handle and stream on that GPU: calls
 *     of mv_hifigan_forward_static on one GPU must be serialised by the caller.  The library reads no
 *     environment variables: what can be tuned is in mv_options, per handle.
 *   - the library needs a gfx950 device: there is no CPU fallback (mv_backend() == 1 always).
 */
#ifndef MELVOC_H
#define MELVOC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MV_API __attribute__((visibility("default")))

#define MV_OK 0
#define MV_EINVAL (-1)   /* bad argument / shape / config */
#define MV_ENOTFOUND (-2) /* a required weight tensor is missing */
#define MV_EHIP (-3)     /* a HIP runtime call failed */
#define MV_ENOMEM (-4)   /* workspace too small */
#define MV_EUNSUPPORTED (-5)

/* arithmetic of the MFMA-mapped convolutions */
#define MV_PREC_FP32 0 /* f32 operands, v_mfma_f32_16x16x4_f32 (exact fp32 fma chains) */
#define MV_PREC_BF16 1 /* bf16 operands, fp32 accumulate, fp32 residual stream (v_mfma_f32_16x16x32_bf16).  This mode
                          may also store activations between kernels as float16 / bfloat16 and regroup linear
                          operations (WaveGlow: end(sum_i skip_i) as sum_i (W_end W_skip_i) acts_i); results stay
                          within SNR >= 40 dB of the fp32 reference.  MV_PREC_FP32 keeps the reference's order of
                          operations (max-abs error <= 1e-3, ~1e-5 measured). */

#define MV_PREC_BF16X3 2 /* float32-grade results on the bf16 MFMA: every float32 operand travels as a (hi, lo) pair of
                            bf16 values and a product is w_hi*x_hi + w_hi*x_lo + w_lo*x_hi, accumulated in fp32; activations
                            between kernels stay float32, the order of operations is MV_PREC_FP32's.  Meets the fp32 bar
                            (max-abs <= 1e-3; ~5e-5 measured) at a third of the bf16 MFMA rate instead of a sixteenth. */

#define MV_PREC_F16 3 /* float16 operands (v_mfma_f32_16x16x32_f16: the bf16 instruction's cycles), fp32 accumulate, fp32
                         residual stream, float16 activations between kernels: MV_PREC_BF16's kernels and speed with 11
                         significant bits per operand instead of 8 (HiFi-GAN: ~18 dB more SNR than bf16).  Every
                         conversion to an operand saturates at +-65504.  WaveGlow runs its layer-by-layer structure in
                         this mode (no ring GEMMs, no folded skip path). */

/* output sample formats of mv_hifigan_forward / mv_waveglow_infer */
#define MV_OUT_F32 0   /* float32 in (-1,1), what Generator.forward / WaveGlow.infer return */
#define MV_OUT_PCM16 1 /* int16 = (int16)(int32)(32768.0f * y): do_syn.py:11,389-394 (truncate, no clip) */

typedef struct mv_handle_s* mv_handle;

/* One named host tensor (float32, C-contiguous) in the reference's own layout:
 * Conv1d weight [Cout][Cin][K], ConvTranspose1d weight [Cin][Cout][K], bias [Cout].
 * Names are the reference's state_dict keys; a weight may be given folded ("x.weight") or as
 * the weight-norm pair ("x.weight_g" [dim0,1,1] + "x.weight_v"), folded here as
 * w = g * v / ||v|| over dims (1,2) per dim-0 index (torch.nn.utils.remove_weight_norm,
 * hifigan/models.py:167-174, waveglow/glow.py:297-313). */
typedef struct mv_tensor {
    const char* name;
    const float* data;
    int ndim;
    int64_t shape[4];
} mv_tensor;

/* hifigan/config.json keys read by Generator.__init__ (hifigan/models.py:113-147). */
typedef struct mv_hifigan_cfg {
    int num_mels;                 /* 80 */
    int upsample_initial_channel; /* 512 */
    int num_upsamples;            /* 4 */
    int upsample_rates[8];        /* 8,8,2,2 */
    int upsample_kernel_sizes[8]; /* 16,16,4,4 */
    int num_kernels;              /* 3 */
    int resblock_kernel_sizes[8]; /* 3,7,11 */
    int resblock_dilations[8][3]; /* 1,3,5 each */
} mv_hifigan_cfg;

/* waveglow/config.json "waveglow_config" (WaveGlow.__init__, waveglow/glow.py:179-205). */
typedef struct mv_waveglow_cfg {
    int n_mel_channels; /* 80 */
    int n_flows;        /* 12 */
    int n_group;        /* 8 */
    int n_early_every;  /* 4 */
    int n_early_size;   /* 2 */
    int n_layers;       /* WN_config.n_layers 8 */
    int n_channels;     /* WN_config.n_channels 256 (repo config) / 512 (waveglow_NEB.pt) */
    int kernel_size;    /* WN_config.kernel_size 3 */
} mv_waveglow_cfg;

/* Postnet hyper-parameters of one decoder (model.py:146-168; tc2_french.yaml:15,87-89). */
typedef struct mv_postnet_cfg {
    int dim_data;        /* hps['dim_data'][i] 80 */
    int embedding_dim;   /* hps['postnet_embedding_dim'][i] 512 */
    int kernel_size;     /* hps['postnet_kernel_size'][i] 5 */
    int n_convolutions;  /* hps['postnet_n_convolutions'][i] 5 */
    float bn_eps;        /* nn.BatchNorm1d default 1e-5 */
} mv_postnet_cfg;

/* Per-handle options of mv_*_create (NULL = the defaults, which are what bench.py measures).  They select
 * other kernels for the same arithmetic; every setting meets the same parity bars.  No counterpart in the
 * reference.  Zero-initialise, then call mv_default_options() before changing single fields. */
typedef struct mv_options {
    int fuse;    /* HiFi-GAN, MV_PREC_BF16: 2 (default) fused ResBlock + whole-stage kernels, 1 fused ResBlock
                    kernels only, 0 one convolution per launch (the MV_PREC_FP32 structure) */
    int io16;    /* MV_PREC_BF16: 1 (default) activations between kernels stored as float16, 0 as float32 */
    int wg_fold; /* WaveGlow, MV_PREC_BF16: 1 (default) skip halves of res_skip folded with the end conv
                    (end(sum_i skip_i) = sum_i (W_end W_skip_i) acts_i), 0 the reference's split (glow.py:168-173) */
    int wg_gemm; /* WaveGlow, MV_PREC_BF16 with io16: the gated in-layer conv (+ the folded skip path) and the
                  * residual 1x1 conv as LDS-DMA ring GEMMs (gate_gemm_kernel, res_gemm_kernel): 1 (default) where
                  * the call has enough row tiles to fill the chip (the residual conv then on 128- or 256-row blocks
                  * by the number of tiles), 2 always (where the shapes allow it), 3 always and with the 256-row
                  * residual blocks wherever the channel count allows them, 0 never (conv_cl_kernel +
                  * wg_skipfold_kernel) */
} mv_options;
MV_API void mv_default_options(mv_options* opt);

/* 1 = gfx950 HIP kernels.  (0 would be a CPU backend; this library has none.) */
MV_API int mv_backend(void);

/* Message of the calling thread's last error ("" if none). */
MV_API const char* mv_last_error(void);

/* Replaces: Generator(h) + load_state_dict(ckpt["generator"]) + eval() + remove_weight_norm()
 * + .to(device)   (do_syn.py:155-169, hifigan/models.py:113-147,167-174).
 * Folds weight-norm, repacks every convolution into MFMA fragment order and uploads it to
 * HIP device `device`.  `precision` is MV_PREC_*. */
MV_API int mv_hifigan_create(const mv_hifigan_cfg* cfg, const mv_tensor* weights, int n_weights, int device,
                      int precision, const mv_options* options, mv_handle* out);

/* Replaces: Generator.forward (hifigan/models.py:149-165) as called at do_syn.py:278,391 and
 * do_tts.py:351.
 *   mel      device float32 [B][num_mels][Tmax], C-contiguous (the reference's layout)
 *   lengths  device int32 [B] true frame counts or NULL (all Tmax); every kernel clamps the entries it reads
 *            into [0, Tmax], so a bad entry cannot take an access out of the buffers.  Every layer
 *            zero-pads at each sequence's own end, so row b equals the reference's batch-1
 *            output for mel[b, :, :lengths[b]]; samples past lengths[b]*hop are written as 0.
 *   wav      device [B][Tmax*hop], float32 (MV_OUT_F32) or int16 (MV_OUT_PCM16); hop = prod(rates)
 *   workspace  device scratch of at least mv_workspace_bytes(h, B, Tmax) bytes, 256-byte aligned */
MV_API int mv_hifigan_forward(mv_handle h, const float* mel, const int32_t* lengths, int B, int Tmax, void* wav,
                       int out_format, void* workspace, size_t workspace_bytes, void* stream);

/* mv_hifigan_forward for callers that cannot allocate device memory (code under HIP graph capture, tight serving
 * loops such as do_tts.py-style streaming of short chunks, harnesses that forbid hipMalloc).  It has NO counterpart
 * in the reference.  Same arguments and results as mv_hifigan_forward minus the workspace: the scratch is a
 * file-scope device array of MV_STATIC_SCRATCH_BYTES inside the library (each GPU the library runs on holds its own
 * copy until the process ends; its address is resolved once per device, in mv_hifigan_create, and if the runtime could
 * not resolve it this call -- not the create -- returns MV_EHIP).  The call allocates
 * nothing (no hipMalloc / hipFree, no allocating library routine) and returns MV_ENOMEM, launching nothing, when
 * mv_workspace_bytes(h, B, Tmax) exceeds the buffer: B * Tmax <= ~403 frames for config_v1 (~166 KB per frame),
 * i.e. 4.7 s of audio per call.  All calls on one GPU share that GPU's array and it keeps the last call's
 * contents (every kernel writes what it later reads, nothing needs to start at zero): calls that overlap on one
 * GPU -- other streams, other threads, other handles -- race; serialise them. */
#define MV_STATIC_SCRATCH_BYTES (64ull << 20)
MV_API int mv_hifigan_forward_static(mv_handle h, const float* mel, const int32_t* lengths, int B, int Tmax,
                                     void* wav, int out_format, void* stream);

/* Replaces: torch.load(path)['model'] + WaveGlow.remove_weightnorm + .cuda().eval()
 * (do_syn.py:147-150, waveglow/glow.py:179-205,297-313).  W^-1 of every Invertible1x1Conv is
 * computed here once in fp32 (glow.py:89-95 caches it on first use). */
MV_API int mv_waveglow_create(const mv_waveglow_cfg* cfg, const mv_tensor* weights, int n_weights, int device,
                       int precision, const mv_options* options, mv_handle* out);

/* Replaces: WaveGlow.infer (waveglow/glow.py:251-295) as called at do_syn.py:276,389 and
 * do_tts.py:349.  The kernels never draw random numbers: the caller passes the unit-variance
 * normal tensors in the reference's draw order (glow.py:262-271, 287-292):
 *   noise[0]      device float32 [B][n_remaining_channels][L]   (L = Tmax*256/n_group)
 *   noise[1..]    device float32 [B][n_early_size][L], one per early-output flow in the order the
 *                 inverse loop meets them (k = 8 then k = 4 for the 12-flow model)
 *   n_noise       number of entries (1 + number of early outputs)
 *   wav           device [B][Tmax*256], sample = l*n_group + channel (glow.py:294) */
MV_API int mv_waveglow_infer(mv_handle h, const float* mel, const int32_t* lengths, int B, int Tmax,
                      const float* const* noise, int n_noise, float sigma, void* wav, int out_format,
                      void* workspace, size_t workspace_bytes, void* stream);

/* Replaces: Postnet(hps, i_decoder) + load_state_dict + eval() + .to(device) (model.py:140-168,614-621).
 * Weights by the module's state_dict keys: "convolutions.<i>.0.conv.weight|bias" and the BatchNorm1d tensors
 * "convolutions.<i>.1.weight|bias|running_mean|running_var", folded into the conv here (eval mode; dropout
 * is inactive in eval, model.py:172-173). */
MV_API int mv_postnet_create(const mv_postnet_cfg* cfg, const mv_tensor* weights, int n_weights, int device,
                      int precision, const mv_options* options, mv_handle* out);

/* Replaces: Postnet.forward (model.py:170-174) and, with add_input != 0, the call site
 * `par_out + self.postnet(par_out)` (model.py:717) whose result do_syn.py:380-391 hands to the vocoder.
 *   x, y     device float32 [B][dim_data][Tmax] (the reference's layout); y may not alias x
 *   lengths  device int32 [B] or NULL; frames past lengths[b] are written as 0 */
MV_API int mv_postnet_forward(mv_handle h, const float* x, const int32_t* lengths, int B, int Tmax, float* y,
                       int add_input, void* workspace, size_t workspace_bytes, void* stream);

/* Scratch bytes mv_*_forward/_infer need for a [B][*][Tmax] batch (either handle kind). */
MV_API size_t mv_workspace_bytes(mv_handle h, int B, int Tmax);

/* Samples produced per mel frame (256 for both reference vocoders). */
MV_API int mv_hop(mv_handle h);

/* Algorithmic multiply-accumulates per mel frame of the model behind `h` (SURVEY.md §8a totals). */
MV_API double mv_macs_per_frame(mv_handle h);

/* Replaces: garbage collection of the Python module. */
MV_API void mv_destroy(mv_handle h);

/* Replaces: `32768.0 * y` -> .cpu().numpy().astype('int16') (do_syn.py:11,389-394) for a float32
 * device buffer of n samples, on device. */
MV_API int mv_pcm16(const float* wav, size_t n, int16_t* pcm, void* stream);

/* ---- measurement: per-launch HIP-event timing of the MFMA kernels (bench.py's roofline figure).
 * No counterpart in the reference.  After mv_profile_begin(h) every convolution launch of the
 * following forward/infer calls is bracketed by hipEventRecord on the call's stream;
 * mv_profile_end(h, ...) synchronises those events, sums them per (kernel symbol, detail) and
 * switches profiling off.  macs_per_frame is the ALGORITHMIC (un-padded) multiply-accumulate count
 * per mel frame summed over the recorded launches: flops = 2 * macs_per_frame * frames of the batch. */
typedef struct mv_profile_row {
    char name[192];  /* the kernel's symbol as rocprofv3 --kernel-trace --stats prints it */
    char detail[32]; /* which of that symbol's launches ("k=11", "+conv_post", ...; "" if there is one kind) */
    int launches;
    double ms;
    double macs_per_frame;
} mv_profile_row;
MV_API int mv_profile_begin(mv_handle h);
MV_API int mv_profile_end(mv_handle h, mv_profile_row* rows, int max_rows, int* n_rows);

/* ---- test hooks: single layers through the same kernels the forward path launches ----------
 * No counterpart in the reference; they let tests/ compare one convolution against the oracle.
 *   x   device float32 channels-last [B][L][Cin]   y  device float32 channels-last [B][Lout][Cout]
 *   w/bias host float32 in the reference layout (Conv1d [Cout][Cin][K]; ConvTranspose1d
 *   [Cin][Cout][K]).  `pre_slope` is the leaky_relu slope applied to x first (1.0 = none). */
MV_API int mv_test_conv1d(const float* x, const int32_t* lengths, int B, int L, int Cin, const float* w,
                   const float* bias, int Cout, int K, int dilation, float pre_slope, int precision, float* y,
                   void* stream);
MV_API int mv_test_conv_transpose1d(const float* x, const int32_t* lengths, int B, int L, int Cin, const float* w,
                             const float* bias, int Cout, int K, int stride, int padding, float pre_slope,
                             int precision, float* y, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MELVOC_H */
