/*
 * kws.h - C ABI of the MI355X-native keyword-spotting inference path (libkws_hip.so).
 *
 * This is the drop-in boundary for honk2's batched inference hot path
 * (SURVEY.md section 8b).  Every entry point names the reference interface it
 * replaces (paths relative to the honk2 repository):
 *
 *   kws_mfcc          <- AudioProcessor.compute_mfccs        utils/audio_processor.py:18-30
 *                        looped per clip by collate_fn        data_loader/audio_data_loader.py:23-35
 *   kws_forward       <- ResNet.forward / CNN.forward        model/resnet.py:38-60, model/cnn.py:79-107
 *   kws_forward_wav   <- collate_fn + model(data)            run/test.py:22-26
 *   kws_create        <- ResNet.__init__ / CNN.__init__      model/resnet.py:11-36, model/cnn.py:12-77
 *                        + AudioProcessor.__init__            utils/audio_processor.py:8-16
 *   kws_load_weights  <- model.load_state_dict               utils/workspace.py:58-61
 *   kws_*_pcm16       <- librosa.load int16->float + `data += noise * noise_pct`   dataset/gsc_dataset.py:163-174
 *   kws_*_windows     <- StreamingDataset.__getitem__ (sliding windows of one long stream)   dataset/dataset_utils.py:20-98
 *   kws_eval_batch    <- loss_fn + metric.accumulate          run/test.py:28-33, loss_function.py:6-9,
 *                                                             metric/acc.py:14-24, metric/per_class_acc.py:14-45
 *
 * Conventions
 *   - plain C types only; device pointers are raw `void*`/`float*` (the caller owns them: the
 *     library BORROWS them for the duration of the call and never frees them).
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Every compute call is
 *     asynchronous on that stream and performs no host synchronisation and no allocation, so a call
 *     sequence can be captured into a hipGraph and replayed back to back without synchronisation (a captured
 *     call is kernel nodes only: the persistent kernels' work queues re-arm themselves; tested, not
 *     guaranteed: INTEGRATION.md section 3) -- with three exceptions, each on the FIRST compute call that
 *     needs it (so: run one un-captured warm-up call per clip length AND at the largest batch size first):
 *     the re-packed parameters are uploaded when the first call after kws_load_weights finalises them; the
 *     tiled ResNet plan (res15 / res26 / narrow models, res8 on clips that are not one second long) builds
 *     and uploads a per-layer position table the first time it sees a clip length T (blocking copies into
 *     buffers of their own; tables of earlier clip lengths are kept and never rewritten); and kws_forward on
 *     the fused res8 plan keeps 4 bytes per clip in a buffer of the handle that is allocated (blocking) the
 *     first time a LARGER batch arrives -- the outgrown block is kept until kws_destroy, because a graph
 *     captured at the smaller size still names it.  A call that would have to allocate while `stream` is
 *     being captured returns KWS_ENOWORKSPACE instead of allocating.
 *     (A cnn-* handle owns a second stream: a kws_forward* call of more than one chunk forks part of each chunk's work
 *     onto it with events and has joined it again before the call returns -- ordered on `stream` at both ends, no host
 *     synchronisation, captured along with the call; kernel and event nodes then.  KWS_CNN_STREAMS=0 at kws_create: off.)
 *   - LIFETIME UNDER GRAPHS: every buffer baked into a captured graph -- the caller's input / output
 *     buffers, the workspace given to kws_set_workspace, and the handle itself (weights, tables, queue
 *     words) -- must outlive every replay of that graph.  Synchronise with the replaying stream before
 *     freeing any of them, replacing the workspace, or calling kws_destroy: a replay that runs after its
 *     buffers were unmapped is a GPU memory fault (INTEGRATION.md section 3 has the one seen in round 3).
 *   - every function returns 0 (KWS_OK) or a negative KWS_E* code; kws_last_error() returns a
 *     thread-local human-readable message for the last failure on the calling thread.
 *   - a handle is bound to the HIP device that was current at kws_create and is not re-entrant: its calls share one workspace,
 *     so they must be ordered on ONE stream (consecutive launches and graph replays there need nothing else), never run
 *     concurrently on two; distinct handles are independent.
 *   - KWS_DTYPE_F32: results are fp32-accurate (fp32 accumulation everywhere).  Products are formed on the 16-bit matrix cores
 *     from split fp32 operands: three exact fp16 x fp16 terms of two-part fp16 splits (weights pre-scaled by a power of two per
 *     layer; error <= 3 * 2^-22 |ab|, measured as close to a float64 evaluation as an fp32 implementation), or --
 *     KWS_RES8_IMPL=bf16x6 -- six bf16 x bf16 terms of three-part bf16 splits.  fp16's range (65504) is no restriction on the
 *     caller: features and activations take any finite fp32 value, as in the reference -- the fused res8 kernel stages
 *     out-of-range feature clips and maps scaled by a power of two (exact), the layer-wise plans recompute a chunk whose
 *     activations leave fp16's range on three-part bf16 operands; both without a host round trip.
 *     The front end's DFT uses the same three-term fp16 products with the samples scaled by a power of two per clip chunk
 *     (any finite sample magnitude is fine); KWS_FRONTEND_IMPL=fp32 selects its fp32-input matrix-core form, which also
 *     serves waveform rows that are not 16-byte aligned.
 *   - Environment: the library reads a few IMPLEMENTATION SELECTORS at kws_create (or per call where noted), each choosing between
 *     implementations that are parity-tested against the same oracle -- KWS_RES8_IMPL=bf16x6|fp32, KWS_FRONTEND_IMPL=fp32,
 *     KWS_LAYERWISE_IMPL=nchw|fp32, KWS_FORCE_LAYERWISE=1, KWS_MATRIX_PARTS=bf16 (per call), KWS_CNN_BAND=0, KWS_CNN_IN1=0,
 *     KWS_CNN_LIN_F16=0, KWS_CNN_COLS=0, KWS_CNN_STREAMS=0, KWS_T3_PAIR=0, KWS_T3_TRIPLE=0|2, KWS_T3_STREAM=0|2|3, KWS_WINDOWS_NO_SHARE (per call) -- and NOTHING ELSE: the ablation bits,
 *     phase stamps, grid-size knobs and the fault-injection hook of the measurement rig exist only in `make EXPERIMENTS=1`
 *     (libkws_hip_exp.so); no environment variable can make this library compute wrong results, synchronise inside a compute
 *     call or throw.
 */
#ifndef KWS_H_
#define KWS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KWS_ABI_VERSION 2

enum {
    KWS_OK = 0,
    KWS_EINVAL = -1,       /* bad argument / shape mismatch                         */
    KWS_ENOMEM = -2,       /* host or device allocation failed (create/load only)  */
    KWS_EUNSUPPORTED = -3, /* configuration outside what the kernels implement     */
    KWS_ENOWORKSPACE = -4, /* workspace missing or too small for this call         */
    KWS_ENOWEIGHTS = -5,   /* forward called before every tensor was loaded        */
    KWS_EHIP = -6          /* a HIP runtime call failed (message has the hipError) */
};

enum { KWS_MODEL_NONE = 0, KWS_MODEL_RESNET = 1, KWS_MODEL_CNN = 2 };
/* KWS_DTYPE_F32: fp32-accurate results (see Conventions).  KWS_DTYPE_BF16X3: opt-in reduced precision -- products are the
 * three leading terms of two-way bf16 splits (relative error ~2^-16 per product, fp32 accumulate); about 2x the matrix
 * throughput of the 6-term form.  Not argmax-exact against the fp32 reference on near-ties.
 * KWS_DTYPE_BF16: plain bf16 operands (weights and activations rounded to bf16 at the matrix operand, fp32 accumulate,
 * fp32 activations between layers, fp32 front end and first conv): BASELINE's bf16 configuration; logits agree with the fp32
 * reference to ~1e-2 at |logit| ~ 1.
 * KWS_DTYPE_F16: plain fp16 operands (BASELINE's fp16 configuration), same structure as KWS_DTYPE_BF16 with 11-bit operands:
 * logits agree with the fp32 reference to ~1e-3 at |logit| ~ 1. */
enum { KWS_DTYPE_F32 = 0, KWS_DTYPE_BF16X3 = 1, KWS_DTYPE_BF16 = 2, KWS_DTYPE_F16 = 3 };

typedef struct kws_conv_desc {
    int32_t out_channels;
    int32_t kernel_h, kernel_w; /* (time, frequency) */
    int32_t stride_h, stride_w;
} kws_conv_desc;

/* Mirrors the reference's JSON `model.config` (config/resnet/res8.json:3-14, config/cnn/cnn-trad-pool2.json)
 * plus the AudioProcessor constructor arguments (utils/audio_processor.py:8). */
typedef struct kws_model_desc {
    int32_t struct_size; /* sizeof(kws_model_desc), ABI check */
    int32_t family;      /* KWS_MODEL_*; KWS_MODEL_NONE = front end only */
    int32_t dtype;       /* KWS_DTYPE_* */
    int32_t n_labels;
    int32_t time;        /* frames of the feature map (CNN: config["time"]; ResNet: nominal, any T accepted) */
    int32_t freq;        /* config["frequency"] / n_mels (40) */
    /* ResNet */
    int32_t n_layers, n_feature_maps, use_dilation;
    int32_t pool_h, pool_w; /* 0,0 = no "pool" key */
    /* CNN */
    int32_t n_conv; /* 1 or 2 */
    kws_conv_desc conv[2];
    int32_t pool_kh[2], pool_kw[2];
    int32_t lin0_out, dnn0_out, dnn1_out; /* 0 = layer absent */
    /* front end */
    int32_t sample_rate, n_fft, hop_length, n_mels;
    float f_min, f_max;
} kws_model_desc;

typedef struct kws_handle kws_handle;

/* Build the execution plan and allocate the (empty) device-side parameter store. */
int kws_create(const kws_model_desc* desc, kws_handle** out);
void kws_destroy(kws_handle* h);

/* Copy one state-dict tensor (fp32, contiguous, reference key name such as "layers.conv_3.weight",
 * "layers.bn_3.running_var", "layers.output.bias"; a leading "module." is ignored) from HOST memory and
 * re-pack it into the kernel's layout.  `bytes` must equal the tensor's size.  "…num_batches_tracked" is
 * accepted and ignored.  Synchronous. */
int kws_load_weights(kws_handle* h, const char* name, const void* host_ptr, size_t bytes);

/* Device scratch the compute calls need for a batch of B clips whose feature maps have T frames.
 * kws_set_workspace zero-fills the block once (synchronous, on the null stream); the compute calls never do. */
size_t kws_workspace_bytes(const kws_handle* h, int B, int T);
int kws_set_workspace(kws_handle* h, void* d_ptr, size_t bytes);

/* Clips per CHUNK of a kws_forward* call of B clips (T frames) on this handle's plan: the layer-wise plans walk a batch in chunks
 * (launch group by launch group), and a chunk is the unit the fp16 range guard recomputes; the fused res8 plan has no chunks
 * (returns B).  Informational (tests, capacity planning); valid once the weights are loaded. */
int kws_chunk_clips(const kws_handle* h, int B, int T);

/* Number of feature frames for n_samples input samples: 1 + n_samples / hop_length. */
int kws_num_frames(const kws_handle* h, int n_samples);

/* wav (B, n_samples) fp32 -> feat (B, T, n_mels) fp32, feat[b,t,f] = 2*ln(mel[f,t]) (0 where mel == 0). */
int kws_mfcc(kws_handle* h, const float* d_wav, int B, int n_samples, float* d_feat, void* stream);

/* Same from 16-bit PCM: sample = pcm / 32768 (+ d_noise[b, i] * noise_pct when d_noise != NULL; d_noise is (B, n_samples)
 * fp32).  The conversion and the mix happen inside the front end's staging load. */
int kws_mfcc_pcm16(kws_handle* h, const int16_t* d_pcm, const float* d_noise, float noise_pct, int B, int n_samples,
                   float* d_feat, void* stream);

/* feat (B, T, freq) fp32 -> logits (B, n_labels) fp32.  Features may be any finite fp32 values (reference model/resnet.py:39-41,
 * model/cnn.py:79-81); needs kws_workspace_bytes(h, B, T) of workspace on every plan (the fused res8 plan keeps 4 bytes per clip
 * there: the power of two an out-of-range clip's features are staged down by). */
int kws_forward(kws_handle* h, const void* d_feat, int B, int T, void* d_logits, void* stream);

/* wav (B, n_samples) fp32 -> logits (B, n_labels) fp32 (feature maps stay in the workspace). */
int kws_forward_wav(kws_handle* h, const float* d_wav, int B, int n_samples, float* d_logits, void* stream);

/* pcm16 (B, n_samples) [+ noise] -> logits. */
int kws_forward_pcm16(kws_handle* h, const int16_t* d_pcm, const float* d_noise, float noise_pct, int B, int n_samples,
                      float* d_logits, void* stream);

/* Streaming evaluation: window i of a long stream is d_stream[i*shift : i*shift + window] (window_size_ms / shift_size_ms of
 * the reference's streaming datasets, in samples); every window takes the usual path (its own reflect padding included,
 * exactly as if it had been copied out), but the windows are read in place -- no (n_windows, window) batch is built.
 * (n_windows - 1) * shift + window must not exceed n_stream.
 * When window and shift are multiples of the hop and the windows overlap, all but the first two and last two frames of a
 * window are frames of the stream itself: they are computed once for the whole stream and only the edge frames per window
 * (needs kws_workspace_bytes_windows() of workspace; with less, every window is transformed on its own). */
size_t kws_workspace_bytes_windows(const kws_handle* h, int window, int shift, int n_windows);
int kws_mfcc_windows(kws_handle* h, const float* d_stream, size_t n_stream, int window, int shift, int n_windows,
                     float* d_feat, void* stream);
int kws_forward_windows(kws_handle* h, const float* d_stream, size_t n_stream, int window, int shift, int n_windows,
                        float* d_logits, void* stream);

/* Evaluation tail, fused: adds to d_stats (int64[3 + 2*n_labels] = correct, total, per-class correct[n],
 * per-class total[n], clips skipped because their target lies outside [0, n_labels)) and to d_loss_sum (double[1]: sum
 * over the counted clips of the cross-entropy, natural log).  A bad target is never used as an index; the caller reads
 * the last counter after its device-to-host copy and treats a non-zero value as the error the reference's
 * F.cross_entropy raises (loss_function.py:6-9). */
int kws_eval_batch(kws_handle* h, const float* d_logits, const int64_t* d_target, int B,
                   int64_t* d_stats, double* d_loss_sum, void* stream);

/* Which execution plan the handle uses, e.g. "res8_fused" or "layerwise". */
const char* kws_plan_name(const kws_handle* h);
/* What the last kws_forward* call launched per chunk on the tiled ResNet plan, layer by layer -- e.g. "conv0 stream(1,2,3) stream(4,5,6)
 * stream(7,8,9) stream(10,11,12) stream(13) mean+linear" --, on the cnn band plan -- e.g. "conv_in1 conv_cols linear | two streams" --, or the
 * plan name on the other plans (diagnostics and tests; the pointer stays valid until the next compute call on the handle). */
const char* kws_plan_detail(const kws_handle* h);

/* Optional in-library timing of the dominant kernel: when enabled every kws_forward* call brackets the model
 * kernel(s) with hipEvents on the caller's stream.  kws_profile_read synchronises those events, returns the
 * accumulated milliseconds and launch count since the last read, and resets them. */
int kws_profile_enable(kws_handle* h, int enable);
int kws_profile_read(kws_handle* h, double* model_ms, double* frontend_ms, int* calls);

const char* kws_last_error(void);
int kws_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* KWS_H_ */
