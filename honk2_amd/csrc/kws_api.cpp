// C ABI of libkws_hip.so (declared in include/kws.h): handle, execution plans, weight packing, dispatch.
// No torch types, no exceptions across the boundary, no allocation or synchronisation in the compute calls.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <new>
#include <set>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include "kws_internal.h"

using namespace kws;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
int fail_hip(hipError_t e, const char* what) {
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return KWS_EHIP;
}
#define HIP_TRY(expr)                                   \
    do {                                                \
        hipError_t _e = (expr);                         \
        if (_e != hipSuccess) return fail_hip(_e, #expr); \
    } while (0)

struct DevMem {
    void* p = nullptr;
    size_t bytes = 0;
    DevMem() = default;
    DevMem(const DevMem&) = delete;              // owns its block: never copied (a copy would free it twice)
    DevMem& operator=(const DevMem&) = delete;
    DevMem(DevMem&& o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
    DevMem& operator=(DevMem&& o) noexcept {
        if (this != &o) {
            if (p) (void)hipFree(p);
            p = o.p; bytes = o.bytes;
            o.p = nullptr; o.bytes = 0;
        }
        return *this;
    }
    ~DevMem() { if (p) (void)hipFree(p); }
    int upload(const void* src, size_t n) {
        if (bytes < n) {
            if (p) (void)hipFree(p);
            p = nullptr;
            bytes = 0;
            HIP_TRY(hipMalloc(&p, n));
            bytes = n;
        }
        HIP_TRY(hipMemcpy(p, src, n, hipMemcpyHostToDevice));
        return KWS_OK;
    }
    // Grow-only scratch of a COMPUTE call (contents are not kept).  A block that was ever handed to a kernel may be baked into a hipGraph the
    // caller captured -- a replay would write freed memory if it were released here -- so an outgrown block is PARKED (freed with the handle),
    // never freed; and nothing is allocated while `s` is being captured (hipMalloc would invalidate the capture): that call fails with
    // KWS_ENOWORKSPACE and asks for the un-captured warm-up call at this batch size that include/kws.h prescribes.
    int reserve_parked(size_t n, std::vector<void*>& parked, hipStream_t s, const char* what) {
        if (bytes >= n) return KWS_OK;
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone)
            return fail(KWS_ENOWORKSPACE, std::string(what) + ": this batch size needs a larger internal buffer and the stream is being captured -- "
                                          "make one un-captured warm-up call with this (or a larger) batch size first");
        const size_t want = std::max(n, 2 * bytes);          // geometric: a batch-size sweep parks O(log B) blocks
        void* np = nullptr;
        HIP_TRY(hipMalloc(&np, want));
        if (p) parked.push_back(p);
        p = np;
        bytes = want;
        return KWS_OK;
    }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

enum Plan { PLAN_FRONTEND_ONLY, PLAN_RESNET, PLAN_CNN };
// layer-wise matrix path: fp32-input MFMA / bf16x6 over fp32 NCHW activations / the same plus the LDS-tiled 3x3 kernel
// over channels-last activations for ResNets (default).  KWS_LAYERWISE_IMPL=fp32|nchw selects the first two (A/B
// measurements, tests).
enum LwMode { LW_FP32 = 0, LW_NCHW = 1, LW_TILED = 2 };

struct ConvLayer {          // one conv (or Linear run as a 1 x K conv) of the layer-wise plans
    ConvGeom g{};           // B/H/W/Ho/Wo filled per call for ResNet, at create for CNN
    DevMem apk, apk16, apk16h, apk_t3h, bias, border, border_pad, w9cl;
    // conv3x3_tile position tables, one per clip length seen (never rewritten: an earlier call on a non-blocking stream may
    // still be reading its table when the next clip length arrives)
    struct PosTab { DevMem mem; int cpc[3] = {0, 0, 0}; };
    std::map<int, std::unique_ptr<PosTab>> postabs;
    std::map<int, std::unique_ptr<PosTab>> postabs3;   // the same for a run of three layers starting here (conv3x3_triple_kernel)
    // apk16h: fp16 fragments of the generic kernel; w9cl: conv_0 as [9 taps][C padded to 8] (conv0_cl_kernel)
    float x_scale = 1.f;     // 2^S of apk16h   // apk_t3h: fp16 fragments of the tiled 3x3 kernel
    float t3h_scale = 1.f;   // 2^S of apk_t3h   // border_pad: rows padded to a multiple of 8 channels (tiled kernel)
    bool has_bias = false, has_border = false, use_x = false;   // use_x: bf16x6 kernel available for this layer
    DevMem apk_band;         // conv_band.hip fragments of a cnn-* conv_1 (fp16 parts x band_scale)
    DevMem apk_cols;         // conv_cols.hip fragments of the same layer (`fp16` dtype, where that kernel fits)
    float band_scale = 1.f;
    DevMem apk_in1[2];       // conv_in1.hip fragments of a cnn-* conv_0 for IN1_MH3 / IN1_MH1 channel tiles per pass (x x_scale)
    std::vector<float> w_host;   // ResNet: raw weights kept until finalize() folds the previous BatchNorm in
};

struct BnHost {
    std::vector<float> mean, var;
};

}  // namespace

struct kws_handle {
    kws_model_desc d{};
    int device = 0;
    int n_cu = 256;
    Plan plan = PLAN_FRONTEND_ONLY;
    bool res8_eligible = false;
    bool force_layerwise = false;
    LwMode lw_mode = LW_TILED;

    // front end
    DevMem dft, hann, dft16, consts16, melw, mel_lo, mel_hi, edge_hann, edge_trig, mel_a;
    bool fe_fp32 = false;                  // KWS_FRONTEND_IMPL=fp32: the fp32-input MFMA front end (frontend.hip)
    int mel_maxw = 0;
    int mel_fb[3] = {0, 0, 0}, mel_ns[3] = {0, 0, 0};   // banded mel GEMM of the fp16 front end (kws_internal.h)

    // parameters
    std::set<std::string> required, loaded;
    bool dirty = true;
    // ResNet
    std::vector<ConvLayer> rconv;          // conv_0 .. conv_n  (layer-wise packing)
    std::vector<BnHost> bn;                // bn_1 .. bn_n (index i-1)
    DevMem bn_mean, bn_rstd;               // (n_layers, C) each: the LAST layer's BN is applied after the spatial mean
    DevMem out_w, out_b;
    // fused res8
    DevMem r8_w0a, r8_apk, r8_bn, r8_zcells, r8x_apk, r8h_apk, r8h_bn, r8h_w0;
    float r8h_scale0 = 1.f;   // 2^S of conv_0's fp16 weights
    std::vector<float> r8_apk_host;
    std::vector<unsigned short> r8x_apk_host, r8h_apk_host;
    float r8h_scale[R8_LAYERS] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f};   // 2^S per layer (fp16 path)
    float r8h_kappa[R8_LAYERS] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f};   // value of the constant channel in the map each layer writes
    // fused res8 kernel: 0 = fp16 three-term products (default), 1 = bf16 six-term (KWS_RES8_IMPL=bf16x6; also what the
    // reduced-precision dtypes use), 2 = fp32-input MFMA (KWS_RES8_IMPL=fp32)
    int res8_impl = 0;
    // CNN
    std::vector<ConvLayer> cconv;          // conv_0 [, conv_1]
    std::vector<ConvLayer> clin;           // lin_0, dnn_0, dnn_1, lin_1 (present ones, in order)
    std::vector<std::string> clin_names;
    int cnn_shape[3][3] = {};              // (C,H,W) after conv/pool stage i (index 0 = input)
    size_t cnn_max_elems = 0;              // largest per-clip activation
    // "cnn_band" plan (fp16-part modes): conv_0 writes channels-last cells, conv_1 runs in conv_band.hip, the first Linear reads
    // its channels-last output through column-permuted weights (clin0_cl)
    int cnn_band_R = 0;                    // output rows per band of conv_1, 0: plan not available
    int cnn_band_parts = 2;                // operand parts the band plan was sized for (1: fp16 tensors only)
    bool cnn_in1 = false;                  // conv_0 runs in conv_in1.hip (LDS image of the clip) in the channels-last plans
    bool cnn_cl1 = false;                  // single-conv model: conv_in1 -> clin0_cl ("cnn_in1" plan)
    int cl_last[3] = {0, 0, 0};            // (channels, positions, channels per cell) of the tensor the first Linear reads
    bool cnn_cols = false;                 // `fp16` dtype: conv_1 runs in conv_cols.hip (column tiles, persistent, double-buffered) instead of conv_band.hip (KWS_CNN_COLS=0: off)
    BandPlan cnn_band;                     // conv_band_plan of conv_1
    DevMem cnn_band_tab;                   // its position table
    int cnn_cp[2] = {0, 0};                // channels per cell of conv_0's / conv_1's output (padded to 16)
    ConvLayer clin0_cl;

    // workspace
    void* ws = nullptr;
    size_t ws_bytes = 0;
    std::string plan_detail;               // tiled ResNet plan: what the last call launched per chunk (kws_plan_detail)
    int plan_detail_T = -1;
    bool lin_in_f16 = true;                // cnn band plan, `fp16` dtype: fp16 cells between conv_1 and the first Linear (KWS_CNN_LIN_F16=0: fp32 cells, A/B and tests)
    int t3_triple = 1;                     // runs of three equal-dilation layers in one kernel (KWS_T3_TRIPLE=0: pairs + singles; 2: any three consecutive layers)
    bool t3_pair = true;                   // tiled plan, 16-bit tensors: consecutive layers of equal dilation in one kernel (KWS_T3_PAIR=0: off)
    int t3_stream = 1;                     // tiled plan, 16-bit tensors, 41-48 channels: runs of three layers and odd single layers as persistent weight-stationary streams (conv3x3_stream.hip; launches of >= 7 680 cells per CU; KWS_T3_STREAM=0: the tile kernels only; 2: launches of any size; 3: odd-first runs only)
    DevMem r8_shift;                       // fused res8, kws_forward: per-clip power-of-two shifts of caller-provided features (feat_shift_kernel)
    std::vector<void*> parked;             // outgrown r8_shift blocks: a captured graph may still name them, so they live as long as the handle
    DevMem range_flag;                     // device words: [0] (and [1], [2]: the other chunks in flight on the two-stream cnn plan) fp16 range guard of the layer-wise plans
                                           // (kws_internal.h), [16] / [32] clip / unit counters of the fused res8 and front-end kernels
    // cnn plans, calls of more than one chunk: a chunk's Linears, split-K reduces and gated second pass run on a stream of the handle while the caller's
    // stream goes on with the next chunk's convolutions (run_cnn; KWS_CNN_STREAMS=0: everything on the caller's stream)
    hipStream_t side = nullptr;
    static constexpr int CNN_RING = 3;     // chunks in flight between the two streams: buffers for what the Linear reads, flag words, events
    hipEvent_t ev_fork[CNN_RING] = {}, ev_join[CNN_RING] = {};
    bool cnn_streams = true;

    // profiling
    bool prof = false;
    std::vector<hipEvent_t> ev_pool;              // created by kws_profile_enable, reused: no hipEventCreate in a timed call
    size_t ev_next = 0;
    std::vector<hipEvent_t> ev_model, ev_front;   // start/stop pairs (entries of ev_pool)
    double acc_model_ms = 0, acc_front_ms = 0;
    int acc_calls = 0;
    const char* last_plan = "none";

    ~kws_handle() {
        for (auto e : ev_pool) (void)hipEventDestroy(e);
        for (void* q : parked) (void)hipFree(q);
        for (int i = 0; i < CNN_RING; ++i) {
            if (ev_fork[i]) (void)hipEventDestroy(ev_fork[i]);
            if (ev_join[i]) (void)hipEventDestroy(ev_join[i]);
        }
        if (side) (void)hipStreamDestroy(side);
    }
};

namespace {

// A handle belongs to the device that was current in kws_create (weights, DFT tables, function attributes live there).
// Every entry point that allocates, copies or launches makes that device current for the duration of the call and
// restores the caller's, so a handle keeps working when the caller's current device has moved on (several GPUs driven
// from one process, as the reference's DataParallel does).
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(const kws_handle* h) {
        if (h && hipGetDevice(&prev) == hipSuccess && prev != h->device) switched = hipSetDevice(h->device) == hipSuccess;
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
};

// ---------------------------------------------------------------------------------------------- front end setup
double hz_to_mel(double f) {
    const double f_sp = 200.0 / 3.0, logstep = std::log(6.4) / 27.0;
    return f >= 1000.0 ? 15.0 + std::log(f / 1000.0) / logstep : f / f_sp;
}
double mel_to_hz(double m) {
    const double f_sp = 200.0 / 3.0, logstep = std::log(6.4) / 27.0;
    return m >= 15.0 ? 1000.0 * std::exp(logstep * (m - 15.0)) : m * f_sp;
}

int setup_frontend(kws_handle* h) {
    const kws_model_desc& d = h->d;
    if (d.n_fft != FE_NFFT || d.hop_length != FE_HOP)
        return fail(KWS_EUNSUPPORTED, "front end kernels are built for n_fft=480, hop_length=160");
    if (d.n_mels <= 0 || d.n_mels > 256) return fail(KWS_EINVAL, "n_mels out of range");
    // Slaney filterbank; double arithmetic, cast to fp32 like librosa's dtype=float32
    const int nb = 1 + d.n_fft / 2;
    std::vector<double> edges(d.n_mels + 2);
    const double m0 = hz_to_mel(d.f_min), m1 = hz_to_mel(d.f_max);
    for (int i = 0; i < d.n_mels + 2; ++i) edges[i] = mel_to_hz(m0 + (m1 - m0) * i / (d.n_mels + 1));
    std::vector<float> wts((size_t)d.n_mels * FE_ROWS, 0.f);
    std::vector<int> lo(d.n_mels), hi(d.n_mels);
    for (int i = 0; i < d.n_mels; ++i) {
        lo[i] = FE_ROWS;
        hi[i] = 0;
        const float enorm = (float)(2.0 / (edges[i + 2] - edges[i]));
        for (int k = 0; k < nb; ++k) {
            const double fk = (double)k * d.sample_rate / d.n_fft;
            const double up = (fk - edges[i]) / (edges[i + 1] - edges[i]);
            const double dn = (edges[i + 2] - fk) / (edges[i + 2] - edges[i + 1]);
            const float wv = (float)std::max(0.0, std::min(up, dn)) * enorm;
            if (wv != 0.f) {
                if (k >= FE_ROWS)
                    return fail(KWS_EUNSUPPORTED, "mel filterbank reaches past DFT bin 127 (f_max too high for this build)");
                wts[(size_t)i * FE_ROWS + k] = wv;
                lo[i] = std::min(lo[i], k);
                hi[i] = std::max(hi[i], k + 1);
            }
        }
        if (hi[i] <= lo[i]) lo[i] = hi[i] = 0;
        h->mel_maxw = std::max(h->mel_maxw, hi[i] - lo[i]);
    }
    std::vector<float> tab, hann;
    build_dft_table(tab, hann);
    int rc;
    if ((rc = h->dft.upload(tab.data(), tab.size() * sizeof(float)))) return rc;
    if ((rc = h->hann.upload(hann.data(), hann.size() * sizeof(float)))) return rc;
    std::vector<unsigned> tab16;
    std::vector<float> hann2;
    build_dft_table_f16(tab16, hann2);
    if ((rc = h->dft16.upload(tab16.data(), tab16.size() * sizeof(unsigned)))) return rc;
    // constants of the fp16 front end: the window halves (LDS blob) and the banded mel GEMM's A fragments
    std::vector<float> blob(FE16_CONST_WORDS, 0.f);
    std::copy(hann2.begin(), hann2.end(), blob.begin());
    if ((rc = h->consts16.upload(blob.data(), blob.size() * sizeof(float)))) return rc;
    std::vector<float> mel_tab;
    const bool mel_fits = build_mel_gemm_table(wts, lo, hi, d.n_mels, mel_tab, h->mel_fb, h->mel_ns);
    if (mel_fits && (rc = h->mel_a.upload(mel_tab.data(), mel_tab.size() * sizeof(float)))) return rc;
    const char* fimpl = std::getenv("KWS_FRONTEND_IMPL");
    h->fe_fp32 = (fimpl && std::strcmp(fimpl, "fp32") == 0) || !mel_fits;   // filterbanks the fp16 kernel's mel stage cannot hold
    std::vector<float> ehann, etrig;
    build_edge_tables(ehann, etrig);
    if ((rc = h->edge_hann.upload(ehann.data(), ehann.size() * sizeof(float)))) return rc;
    if ((rc = h->edge_trig.upload(etrig.data(), etrig.size() * sizeof(float)))) return rc;
    if ((rc = h->melw.upload(wts.data(), wts.size() * sizeof(float)))) return rc;
    if ((rc = h->mel_lo.upload(lo.data(), lo.size() * sizeof(int)))) return rc;
    if ((rc = h->mel_hi.upload(hi.data(), hi.size() * sizeof(int)))) return rc;
    return KWS_OK;
}

// ---------------------------------------------------------------------------------------------- geometry helpers
int conv_out(int n, int k, int s, int p, int dil) { return (n + 2 * p - (dil * (k - 1) + 1)) / s + 1; }

void finish_geom(ConvGeom& g) {
    g.kx_inner = g.Cin == 1 ? 1 : 0;
    g.inner_steps = g.kx_inner ? (g.kw + 3) / 4 : (g.Cin + 3) / 4;
    g.ksteps = g.kx_inner ? g.kh * g.inner_steps : g.kh * g.kw * g.inner_steps;
    g.mtiles = (g.Cout + 15) / 16;
    g.MT = choose_mt(g.mtiles);
}

ConvGeom make_geom(int cin, int cout, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int relu) {
    ConvGeom g{};
    g.Cin = cin; g.Cout = cout; g.kh = kh; g.kw = kw; g.sh = sh; g.sw = sw; g.ph = ph; g.pw = pw; g.dh = dh; g.dw = dw;
    g.relu = relu;
    finish_geom(g);
    return g;
}

void set_spatial(ConvGeom& g, int B, int H, int W) {
    g.B = B; g.H = H; g.W = W;
    g.Ho = conv_out(H, g.kh, g.sh, g.ph, g.dh);
    g.Wo = conv_out(W, g.kw, g.sw, g.pw, g.dw);
}

int resnet_dilation(const kws_model_desc& d, int i) { return d.use_dilation ? 1 << ((i - 1) / 3) : 1; }

int build_resnet(kws_handle* h) {
    const kws_model_desc& d = h->d;
    if (d.n_layers < 1 || d.n_feature_maps < 1 || d.n_labels < 1) return fail(KWS_EINVAL, "bad ResNet description");
    // sizes are bounded BEFORE anything is allocated from them (a depth of INT_MAX must come back as a code, not as a terabyte resize)
    if (d.n_layers > 1024 || d.n_feature_maps > 4096 || d.n_labels > 65536 || d.freq < 1 || d.freq > 65536)
        return fail(KWS_EUNSUPPORTED, "ResNet description out of range (n_layers <= 1024, n_feature_maps <= 4096, n_labels <= 65536, freq <= 65536)");
    const int C = d.n_feature_maps;
    h->rconv.resize(d.n_layers + 1);
    h->rconv[0].g = make_geom(1, C, 3, 3, 1, 1, 1, 1, 1, 1, 1);
    h->required.insert("layers.conv_0.weight");
    h->bn.resize(d.n_layers);
    for (int i = 1; i <= d.n_layers; ++i) {
        const int dil = resnet_dilation(d, i);
        h->rconv[i].g = make_geom(C, C, 3, 3, 1, 1, dil, dil, dil, dil, 1);
        h->rconv[i].g.accumulate = (i % 2 == 0);
        const std::string s = std::to_string(i);
        h->required.insert("layers.conv_" + s + ".weight");
        h->required.insert("layers.bn_" + s + ".running_mean");
        h->required.insert("layers.bn_" + s + ".running_var");
    }
    h->required.insert("layers.output.weight");
    h->required.insert("layers.output.bias");
    h->res8_eligible = d.n_layers == R8_LAYERS && C == R8_C && !d.use_dilation && d.pool_h == 4 && d.pool_w == 3 &&
                       d.freq == 40 && d.n_labels <= 256;
    if (h->res8_eligible) {
        h->r8_apk_host.assign(R8_APK_FLOATS, 0.f);
        h->r8x_apk_host.assign(R8X_APK_SHORTS, 0);
        h->r8h_apk_host.assign(R8H_APK_SHORTS, 0);
        const char* impl = std::getenv("KWS_RES8_IMPL");
        h->res8_impl = impl && std::strcmp(impl, "fp32") == 0 ? 2 : (impl && std::strcmp(impl, "bf16x6") == 0 ? 1 : 0);
        if ((d.dtype == KWS_DTYPE_BF16X3 || d.dtype == KWS_DTYPE_BF16) && h->res8_impl == 0) h->res8_impl = 1;   // bf16 kernel
    }
    return KWS_OK;
}

int build_cnn(kws_handle* h) {
    const kws_model_desc& d = h->d;
    if (d.n_conv < 1 || d.n_conv > 2 || d.time < 1 || d.freq < 1) return fail(KWS_EINVAL, "bad CNN description");
    if (d.time > (1 << 20) || d.freq > 65536 || d.n_labels < 1 || d.n_labels > 65536 || d.lin0_out < 0 || d.lin0_out > (1 << 20) ||
        d.dnn0_out < 0 || d.dnn0_out > (1 << 20) || d.dnn1_out < 0 || d.dnn1_out > (1 << 20))
        return fail(KWS_EUNSUPPORTED, "CNN description out of range (time <= 2^20, freq <= 65536, n_labels <= 65536, linear widths <= 2^20)");
    for (int i = 0; i < d.n_conv; ++i)
        if (d.conv[i].out_channels > 4096 || d.conv[i].kernel_h > 4096 || d.conv[i].kernel_w > 4096 || d.conv[i].stride_h > 4096 ||
            d.conv[i].stride_w > 4096 || d.pool_kh[i] > 4096 || d.pool_kw[i] > 4096)
            return fail(KWS_EUNSUPPORTED, "conv/pool description out of range (channels, kernel, stride, pool <= 4096)");
    int C = 1, H = d.time, W = d.freq;
    h->cnn_shape[0][0] = C; h->cnn_shape[0][1] = H; h->cnn_shape[0][2] = W;
    h->cconv.resize(d.n_conv);
    size_t mx = 0;
    for (int i = 0; i < d.n_conv; ++i) {
        const kws_conv_desc& c = d.conv[i];
        if (c.out_channels < 1 || c.kernel_h < 1 || c.kernel_w < 1 || c.stride_h < 1 || c.stride_w < 1 ||
            d.pool_kh[i] < 1 || d.pool_kw[i] < 1)
            return fail(KWS_EINVAL, "bad conv/pool description");
        ConvGeom g = make_geom(C, c.out_channels, c.kernel_h, c.kernel_w, c.stride_h, c.stride_w, 0, 0, 1, 1, 1);
        set_spatial(g, 0, H, W);
        if (g.Ho < 1 || g.Wo < 1) return fail(KWS_EINVAL, "conv kernel larger than its input");
        if (d.pool_kh[i] * d.pool_kw[i] > 4) g.x_mt_cap = 3;   // multi-pass fused pooling exists for <= 3 channel tiles per wave
        h->cconv[i].g = g;
        h->cconv[i].has_bias = true;
        mx = std::max(mx, (size_t)g.Cout * g.Ho * g.Wo);
        C = g.Cout;
        H = g.Ho / d.pool_kh[i];
        W = g.Wo / d.pool_kw[i];
        if (H < 1 || W < 1) return fail(KWS_EINVAL, "pool kernel larger than its input");
        h->cnn_shape[i + 1][0] = C; h->cnn_shape[i + 1][1] = H; h->cnn_shape[i + 1][2] = W;
        const std::string s = std::to_string(i);
        h->required.insert("layers.conv_" + s + ".weight");
        h->required.insert("layers.conv_" + s + ".bias");
    }
    int feat = C * H * W;
    mx = std::max(mx, (size_t)feat);
    const int outs[4] = {d.lin0_out, d.dnn0_out, d.dnn1_out, d.n_labels};
    const char* names[4] = {"lin_0", "dnn_0", "dnn_1", "lin_1"};
    for (int i = 0; i < 4; ++i) {
        if (outs[i] <= 0) continue;
        ConvLayer L;
        L.g = make_geom(1, outs[i], 1, feat, 1, 1, 0, 0, 1, 1, 0);
        set_spatial(L.g, 0, 1, feat);
        L.has_bias = true;
        h->clin.push_back(std::move(L));
        h->clin_names.push_back(names[i]);
        h->required.insert(std::string("layers.") + names[i] + ".weight");
        h->required.insert(std::string("layers.") + names[i] + ".bias");
        feat = outs[i];
        mx = std::max(mx, (size_t)feat);
    }
    // band plan for conv_1: stride 1, identity pool_1, conv_0's pooling fusable, a Linear behind it
    const bool band_off = std::getenv("KWS_CNN_BAND") && std::atoi(std::getenv("KWS_CNN_BAND")) == 0;   // A/B and tests
    if (d.n_conv == 2 && !band_off && h->lw_mode == LW_TILED && d.conv[1].stride_h == 1 && d.conv[1].stride_w == 1 &&
        d.pool_kh[1] * d.pool_kw[1] == 1 && d.pool_kh[0] * d.pool_kw[0] <= 16 && !h->clin.empty()) {
        const int C0 = h->cnn_shape[1][0], H1 = h->cnn_shape[1][1], W1 = h->cnn_shape[1][2];
        const int C1 = d.conv[1].out_channels;
        // (fp16 tensors -- KWS_DTYPE_F16 -- hold one operand part: bands of twice the rows; the guard's second pass does not use this plan)
        h->cnn_band_parts = d.dtype == KWS_DTYPE_F16 ? 1 : 2;
        if (conv_band_plan(C0, C1, H1, W1, d.conv[1].kernel_h, d.conv[1].kernel_w, h->cnn_band_parts, h->cnn_band)) {
            int rcu = h->cnn_band_tab.upload(h->cnn_band.tab.data(), h->cnn_band.tab.size() * sizeof(int));
            if (rcu) return rcu;
            h->cnn_band_R = h->cnn_band.R;
            h->cnn_cp[0] = (C0 + 15) / 16 * 16;
            h->cnn_cp[1] = (C1 + 15) / 16 * 16;
            const ConvGeom& g1 = h->cconv[1].g;
            const int kcl = g1.Ho * g1.Wo * h->cnn_cp[1];
            h->clin0_cl.g = make_geom(1, h->clin[0].g.Cout, 1, kcl, 1, 1, 0, 0, 1, 1, 0);
            set_spatial(h->clin0_cl.g, 0, 1, kcl);
            h->clin0_cl.has_bias = true;
            mx = std::max(mx, std::max((size_t)H1 * W1 * h->cnn_cp[0], (size_t)kcl));
            h->cl_last[0] = C1; h->cl_last[1] = g1.Ho * g1.Wo; h->cl_last[2] = h->cnn_cp[1];
            const bool cols_off = std::getenv("KWS_CNN_COLS") && std::atoi(std::getenv("KWS_CNN_COLS")) == 0;   // A/B and tests
            h->cnn_cols = !cols_off && d.dtype == KWS_DTYPE_F16 && conv_cols_supported(C0, C1, H1, W1, d.conv[1].kernel_h, d.conv[1].kernel_w);
        }
    }
    const bool in1_off = std::getenv("KWS_CNN_IN1") && std::atoi(std::getenv("KWS_CNN_IN1")) == 0;   // A/B and tests
    const bool in1_ok = !in1_off && !band_off && h->lw_mode == LW_TILED && conv_in1_supported(h->cconv[0].g, d.pool_kh[0], d.pool_kw[0]);
    if (h->cnn_band_R > 0) h->cnn_in1 = in1_ok;
    else if (d.n_conv == 1 && in1_ok && !h->clin.empty()) {   // single-conv models: the Linear reads conv_0's channels-last cells
        const int C0 = h->cnn_shape[1][0], H1 = h->cnn_shape[1][1], W1 = h->cnn_shape[1][2];
        h->cnn_in1 = h->cnn_cl1 = true;
        h->cnn_cp[0] = (C0 + 15) / 16 * 16;
        const int kcl = H1 * W1 * h->cnn_cp[0];
        h->clin0_cl.g = make_geom(1, h->clin[0].g.Cout, 1, kcl, 1, 1, 0, 0, 1, 1, 0);
        set_spatial(h->clin0_cl.g, 0, 1, kcl);
        h->clin0_cl.has_bias = true;
        mx = std::max(mx, (size_t)kcl);
        h->cl_last[0] = C0; h->cl_last[1] = H1 * W1; h->cl_last[2] = h->cnn_cp[0];
    }
    h->cnn_max_elems = mx;
    return KWS_OK;
}

int upload_packed(ConvLayer& L, const float* w, LwMode mode) {
    std::vector<float> pk;
    pack_conv_weights(L.g, w, pk);
    int rc = L.apk.upload(pk.data(), pk.size() * sizeof(float));
    if (rc) return rc;
    conv_bf16x6_geometry(L.g);
    L.use_x = mode != LW_FP32 && conv_bf16x6_supported(L.g);
    if (L.use_x) {
        std::vector<unsigned short> pk16;
        pack_conv_weights_bf16x6(L.g, w, pk16);
        if ((rc = L.apk16.upload(pk16.data(), pk16.size() * sizeof(unsigned short)))) return rc;
        L.x_scale = weight_scale_pow2(w, (size_t)L.g.Cout * L.g.Cin * L.g.kh * L.g.kw);
        pack_conv_weights_f16x3(L.g, w, L.x_scale, pk16);
        rc = L.apk16h.upload(pk16.data(), pk16.size() * sizeof(unsigned short));
    }
    return rc;
}

void decode_mode(int mode, int& f16, int& terms);

// one conv launch through whichever kernel the layer supports (geometry fields B/H/W already set in g)
int launch_layer(const ConvLayer& L, const ConvGeom& g_in, ConvArgs a, hipStream_t s, int terms = 6) {
    ConvGeom g = g_in;
    decode_mode(terms, g.x_f16, g.x_terms);
    g.x_inv_scale = g.x_f16 ? 1.0f / L.x_scale : 1.0f;
    if (L.use_x) {
        a.apk16 = g.x_f16 ? L.apk16h.as<unsigned short>() : L.apk16.as<unsigned short>();
        HIP_TRY(launch_conv_bf16x6(g, a, s));
    } else {
        HIP_TRY(launch_conv(g, a, s));
    }
    return KWS_OK;
}

// ---------------------------------------------------------------------------------------------- finalize (lazy)
int finalize(kws_handle* h) {
    if (!h->dirty) return KWS_OK;
    for (const auto& r : h->required)
        if (!h->loaded.count(r)) return fail(KWS_ENOWEIGHTS, "tensor not loaded: " + r);
    if (h->plan == PLAN_RESNET) {
        const int C = h->d.n_feature_maps, n = h->d.n_layers;
        std::vector<float> sc((size_t)n * C), sf((size_t)n * C), mu((size_t)n * C), rs((size_t)n * C);
        for (int i = 0; i < n; ++i)
            for (int c = 0; c < C; ++c) {
                const double r = 1.0 / std::sqrt((double)h->bn[i].var[c] + 1e-5);
                mu[(size_t)i * C + c] = h->bn[i].mean[c];
                rs[(size_t)i * C + c] = (float)r;
                sc[(size_t)i * C + c] = (float)r;
                sf[(size_t)i * C + c] = (float)(-(double)h->bn[i].mean[c] * r);
            }
        int rc;
        if ((rc = h->bn_mean.upload(mu.data(), mu.size() * 4))) return rc;
        if ((rc = h->bn_rstd.upload(rs.data(), rs.size() * 4))) return rc;
        // layer-wise plan: conv_i (i >= 2) reads BN_{i-1}(x).  Fold the BN scale into the weights and turn the BN
        // shift into a border bias: for each of the 16 border classes (top/bottom/left/right tap rows in bounds) the sum
        // over in-bounds taps of sum_ci W[co][ci][tap] * shift[ci]  (zero padding is applied AFTER BN in the reference).
        for (int i = 0; i <= n; ++i) {
            ConvLayer& L = h->rconv[i];
            std::vector<float> wf = L.w_host;
            if (i >= 2) {
                const float* scp = sc.data() + (size_t)(i - 2) * C;
                const float* sfp = sf.data() + (size_t)(i - 2) * C;
                std::vector<float> border((size_t)16 * C, 0.f);
                for (int co = 0; co < C; ++co) {
                    double tapsum[9];
                    for (int t9 = 0; t9 < 9; ++t9) {
                        double acc = 0.0;
                        for (int ci = 0; ci < C; ++ci) acc += (double)L.w_host[((size_t)co * C + ci) * 9 + t9] * sfp[ci];
                        tapsum[t9] = acc;
                    }
                    for (int mask = 0; mask < 16; ++mask) {
                        double acc = 0.0;
                        for (int ky = 0; ky < 3; ++ky)
                            for (int kx = 0; kx < 3; ++kx) {
                                const bool rowok = ky == 1 || (ky == 0 ? (mask & 1) : (mask & 2));
                                const bool colok = kx == 1 || (kx == 0 ? (mask & 4) : (mask & 8));
                                if (rowok && colok) acc += tapsum[ky * 3 + kx];
                            }
                        border[(size_t)mask * C + co] = (float)acc;
                    }
                    for (int ci = 0; ci < C; ++ci)
                        for (int t9 = 0; t9 < 9; ++t9) wf[((size_t)co * C + ci) * 9 + t9] *= scp[ci];
                }
                if ((rc = L.border.upload(border.data(), border.size() * 4))) return rc;
                const int cp = (C + 7) / 8 * 8;
                std::vector<float> bpad((size_t)16 * cp, 0.f);
                for (int mask = 0; mask < 16; ++mask)
                    for (int co = 0; co < C; ++co) bpad[(size_t)mask * cp + co] = border[(size_t)mask * C + co];
                if ((rc = L.border_pad.upload(bpad.data(), bpad.size() * 4))) return rc;
                L.has_border = true;
            }
            if ((rc = upload_packed(L, wf.data(), h->lw_mode))) return rc;
            if (i == 0 && h->lw_mode == LW_TILED && conv3x3_tile_supported(C, C, 1)) {
                const int cp = (C + 7) / 8 * 8;
                std::vector<float> w9((size_t)9 * cp, 0.f);
                for (int co = 0; co < C; ++co)
                    for (int t9 = 0; t9 < 9; ++t9) w9[(size_t)t9 * cp + co] = L.w_host[(size_t)co * 9 + t9];
                if ((rc = L.w9cl.upload(w9.data(), w9.size() * 4))) return rc;
            }
            if (i >= 1 && h->lw_mode == LW_TILED && conv3x3_tile_supported(C, C, 1)) {
                std::vector<unsigned short> pk;
                L.t3h_scale = weight_scale_pow2(wf.data(), wf.size());
                pack_conv3x3_tile_weights_f16(C, wf.data(), L.t3h_scale, pk);
                if ((rc = L.apk_t3h.upload(pk.data(), pk.size() * sizeof(unsigned short)))) return rc;
            }
        }
        if (h->res8_eligible) {
            std::vector<float> tab((size_t)R8_LAYERS * 96, 0.f);
            for (int i = 0; i < R8_LAYERS; ++i)
                for (int c = 0; c < R8_C; ++c) {
                    tab[(size_t)i * 96 + c] = sc[(size_t)i * C + c];
                    tab[(size_t)i * 96 + 48 + c] = sf[(size_t)i * C + c];
                }
            if ((rc = h->r8_bn.upload(tab.data(), tab.size() * 4))) return rc;
            if ((rc = h->r8_apk.upload(h->r8_apk_host.data(), h->r8_apk_host.size() * 4))) return rc;
            if ((rc = h->r8x_apk.upload(h->r8x_apk_host.data(), h->r8x_apk_host.size() * 2))) return rc;
            // res8h_kernel: BatchNorm i - 1 is folded into conv_i (i >= 2).  Its scale goes into the weights; its shift rides on a
            // constant input channel (slot 45 of the 48-channel cells: kappa inside the map, 0 in the halo -- so the reference's zero
            // padding AFTER BatchNorm comes out by itself): w46[co][45][tap] = sum_ci W[co][ci][tap] shift[ci].  The map behind an
            // odd layer holds relu(acc) as it comes (true value x mu, mu = the layer's weight scale 2^S), behind an even layer the
            // residual stream x (mu = 1); 1 / mu goes into the next layer's weights and kappa = mu.  Only the last BatchNorm is left
            // for the tail (45 values per clip).
            {
                double mu_prev = 1.0;   // scale of the map conv_i reads (x_0: 1)
                std::vector<float> w46((size_t)R8_C * 46 * 9);
                for (int i = 1; i <= R8_LAYERS; ++i) {
                    const std::vector<float>& W = h->rconv[i].w_host;
                    const float* scp = i >= 2 ? sc.data() + (size_t)(i - 2) * C : nullptr;
                    const float* sfp = i >= 2 ? sf.data() + (size_t)(i - 2) * C : nullptr;
                    for (int co = 0; co < R8_C; ++co)
                        for (int t9 = 0; t9 < 9; ++t9) {
                            double bias = 0.0;
                            for (int ci = 0; ci < R8_C; ++ci) {
                                const double wv = W[((size_t)co * R8_C + ci) * 9 + t9];
                                w46[((size_t)co * 46 + ci) * 9 + t9] = (float)(wv * (scp ? (double)scp[ci] : 1.0) / mu_prev);
                                if (sfp) bias += wv * (double)sfp[ci];
                            }
                            w46[((size_t)co * 46 + 45) * 9 + t9] = (float)(bias / mu_prev);
                        }
                    h->r8h_scale[i - 1] = weight_scale_pow2(w46.data(), w46.size());
                    pack_res8h_layer(w46.data(), h->r8h_scale[i - 1],
                                     h->r8h_apk_host.data() + (size_t)(i - 1) * R8H_ASTEPS * 3 * 2 * 64 * 8);
                    mu_prev = (i % 2 == 1) ? (double)h->r8h_scale[i - 1] : 1.0;
                    h->r8h_kappa[i - 1] = (float)mu_prev;
                }
            }
            if ((rc = h->r8h_apk.upload(h->r8h_apk_host.data(), h->r8h_apk_host.size() * 2))) return rc;
            std::vector<float> tabh(96, 0.f);   // the LAST BatchNorm only: scale[48], shift[48]
            for (int c = 0; c < R8_C; ++c) {
                tabh[c] = sc[(size_t)(R8_LAYERS - 1) * C + c];
                tabh[48 + c] = sf[(size_t)(R8_LAYERS - 1) * C + c];
            }
            if ((rc = h->r8h_bn.upload(tabh.data(), tabh.size() * 4))) return rc;
            std::vector<int> zc(1024);
            build_res8_zero_cells(zc.data());
            if ((rc = h->r8_zcells.upload(zc.data(), zc.size() * sizeof(int)))) return rc;
        }
    }
    h->dirty = false;
    return KWS_OK;
}

// ---------------------------------------------------------------------------------------------- sizes
size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

size_t feat_bytes(const kws_handle* h, int B, int T) { return align256((size_t)B * T * h->d.freq * sizeof(float)); }

bool use_fused(const kws_handle* h, int T) {
    return h->plan == PLAN_RESNET && h->res8_eligible && !h->force_layerwise && T == 101;
}

struct ResnetShape { int C, T, F, H, W; bool pooled; };
ResnetShape resnet_shape(const kws_handle* h, int T) {
    ResnetShape s{};
    s.C = h->d.n_feature_maps; s.T = T; s.F = h->d.freq;
    s.pooled = h->d.pool_h > 0 && h->d.pool_w > 0;
    s.H = s.pooled ? T / h->d.pool_h : T;
    s.W = s.pooled ? s.F / h->d.pool_w : s.F;
    return s;
}

int pad8(int c) { return (c + 7) / 8 * 8; }
// bf16 product terms per fp32 product: 6 = fp32-accurate, 3 = KWS_DTYPE_BF16X3, 1 = plain bf16 operands
// matrix mode code passed down to the launchers: 6 = fp32-accurate default (two-part fp16 operands, three terms), 16 = plain
// fp16 operands (KWS_DTYPE_F16), 3 = KWS_DTYPE_BF16X3, 1 = KWS_DTYPE_BF16 (bf16 parts), 66 = fp32-accurate on three-part bf16
// operands, six terms (KWS_MATRIX_PARTS=bf16: no fp16 range limit)
int dtype_terms(int dtype) {
    const char* mp = std::getenv("KWS_MATRIX_PARTS");   // read per call on purpose (tests flip it)
    const bool bf16_parts = mp && std::strcmp(mp, "bf16") == 0;
    return dtype == KWS_DTYPE_BF16X3 ? 3 : dtype == KWS_DTYPE_BF16 ? 1 : dtype == KWS_DTYPE_F16 ? 16 : (bf16_parts ? 66 : 6);
}
void decode_mode(int mode, int& f16, int& terms) {
    f16 = (mode == 6 || mode == 16) ? 1 : 0;
    terms = mode == 6 ? 3 : mode == 16 ? 1 : mode == 66 ? 6 : mode;
}
int ilog2(int d) { int l = 0; while ((1 << l) < d) ++l; return l; }

// LDS-tiled 3x3 kernel usable for every conv_i of this ResNet?
bool resnet_tiled(const kws_handle* h, const ResnetShape& s) {
    return h->lw_mode == LW_TILED && conv3x3_tile_supported(s.C, s.C, s.W);
}

// cells per clip of the largest sub-map layout any layer uses (padding of the sub-maps included)
size_t resnet_cl_cells(const kws_handle* h, const ResnetShape& s) {
    size_t mx = (size_t)s.H * s.W;
    for (int i = 1; i <= h->d.n_layers; ++i) {
        const int d = resnet_dilation(h->d, i);
        mx = std::max(mx, (size_t)d * d * ((s.H + d - 1) / d) * ((s.W + d - 1) / d));
    }
    return mx;
}

// clips per launch of the tiled plan: tensors under 1 GiB and under 2^24 cells (conv3x3_tile.hip decodes positions with
// fp32 reciprocals)
int chunk_clips(size_t per_clip_elems, int B, size_t cap = 1024);
int tiled_chunk(const kws_handle* h, const ResnetShape& s, int B) {
    const size_t cells = resnet_cl_cells(h, s);
    // (r5) a chunk is sized by its CELLS, not its clips: ~5.5 M cells of the widest layout (1 024 clips of res15's 101 x 40 map, 4 096 of res26's pooled 50 x 20 one), every
    // tensor under 1 GiB.  The stream kernels give each CU one contiguous span of a launch's cells and pay ~11 steps of pipeline fill per span: at 1 024
    // res26 clips a span was 62 steps long (stream 6.35 ms against 5.80 for the pair kernels at B = 4 096), at 4 096 it is 250.
    // (Only where streams can run -- the 16-bit dtypes, 41 - 48 channels; every other plan keeps chunks of <= 1 024 clips, the unit the fp16 range guard
    // recomputes.)
    // (r5) as for the cnn plans: where three fp32 tensors of more than 1 024 clips fit the 256 MB cache (res8-sized pooled maps of the narrow models), a chunk takes
    // them -- up to 4 096 -- and the ~10 gated launches per chunk weigh less (res8-narrow: 8 of them are a fifth of a 1 024-clip chunk's 0.2 ms)
    const size_t live3 = 3 * cells * pad8(s.C) * 4;
    const size_t cap32 = std::min<size_t>(4096, std::max<size_t>(1024, ((size_t)256 << 20) / std::max<size_t>(live3, 1)));
    int cb = chunk_clips(std::max((size_t)s.C * s.T * s.F, cells * pad8(s.C)), B, cap32);
    if ((h->d.dtype == KWS_DTYPE_BF16 || h->d.dtype == KWS_DTYPE_F16) && h->t3_stream && pad8(s.C) == 48) {
        static const int budget_clips = std::max(1, experiment_int("KWS_T3_CHUNK_BUDGET", 2048));
        size_t cbs = std::max<size_t>(1, std::min<size_t>((size_t)budget_clips * 5376 / cells, 4096));   // (5 376 = res15's cells at dilation 16, padded sub-maps included: 2 048 res15 clips, 4 096 res26 clips)
        cbs = std::min(cbs, ((size_t)1 << 29) / std::max<size_t>(cells * pad8(s.C), 1));                 // (two-byte elements: 1 GiB per tensor)
        cb = (int)std::max<size_t>(1, std::min<size_t>(cbs, (size_t)std::max(B, 1)));
    }
    size_t cap = (((size_t)1 << 24) - 4096) / cells;
    static const int env_cap = experiment_int("KWS_TILED_CHUNK", 0);
    if (env_cap > 0) cap = std::min<size_t>(cap, env_cap);
    return (int)std::max<size_t>(1, std::min<size_t>(cb, cap));
}

// clips per layer-wise launch: keep every activation tensor under 2^28 elements (1 GiB) and 32-bit indexable
int chunk_clips(size_t per_clip_elems, int B, size_t cap) {
    size_t cb = ((size_t)1 << 28) / std::max<size_t>(per_clip_elems, 1);
    cb = std::max<size_t>(1, std::min<size_t>(cb, cap));
    return (int)std::min<size_t>(cb, (size_t)std::max(B, 1));
}

// clips per launch of a cnn-* plan (KWS_CNN_CHUNK: experiments.  r4, cnn-trad-pool2 fp16 at B = 8 192: 512 clips 2.19 ms, 768 1.91, 1 024 1.79, 1 536 1.78 -- fewer, larger launches win
// over whole rounds of workgroups; f32 likewise)
// largest per-clip tensor a call really writes: where every MaxPool is reduced in its conv's accumulators (the default kernels, windows of <= 16 members)
// the un-pooled maps in cnn_max_elems never exist
size_t cnn_live_elems(const kws_handle* h) {
    size_t mx = 1;
    for (size_t i = 0; i < h->cconv.size(); ++i) {
        const ConvGeom& g = h->cconv[i].g;
        const int members = h->d.pool_kh[i] * h->d.pool_kw[i];
        if (!(h->cconv[i].use_x && members <= 16)) return h->cnn_max_elems;
        mx = std::max(mx, (size_t)((g.Cout + 15) / 16 * 16) * (g.Ho / h->d.pool_kh[i]) * (g.Wo / h->d.pool_kw[i]));
    }
    for (const auto& L : h->clin) mx = std::max(mx, (size_t)L.g.Cout * L.g.Ho * L.g.Wo);
    return std::min(mx, std::max<size_t>(h->cnn_max_elems, 1));
}

// bytes per clip a chunk keeps alive between its kernels on the channels-last plans (conv_0's cells, conv_1's cells); 0: another plan
size_t cnn_live_bytes(const kws_handle* h) {
    if (!(h->cnn_band_R > 0 || h->cnn_cl1) || !h->cnn_in1) return 0;
    const size_t esz = h->d.dtype == KWS_DTYPE_F16 ? 2 : 4;
    size_t n = (size_t)h->cnn_cp[0] * h->cnn_shape[1][1] * h->cnn_shape[1][2];
    if (h->cnn_band_R > 0) n += (size_t)h->cnn_cp[1] * h->cconv[1].g.Ho * h->cconv[1].g.Wo;
    return n * esz;
}

// (r5) Clips per chunk of a cnn-* call.  Rounds 3 - 4 found 1 024 best (larger chunks lost the Infinity Cache between the layers, smaller ones paid the
// launches); with a chunk's small kernels on a second stream and conv_1 persistent the balance moved: the fixed cost of a chunk's ~10 launches no longer
// hides, and a chunk is as large as keeps its activations inside the 256 MB cache -- 1 024 clips at least, 4 096 at most (B = 12 288, `fp16`, against
// 1 024-clip chunks: cnn-one-fstride4 2.12 -> 1.23 ms, cnn-tstride8 1.85 -> 0.98, cnn-tstride4 2.05 -> 1.34, cnn-trad-pool2 2.13 -> 2.07; f32: cnn-tstride8
// 2.26 -> 1.71, cnn-one-fstride4 3.04 -> 2.55, the large two-conv models within 1 %).  Chunks of a call are equal (to 64 clips), so no short chunk trails.
int cnn_chunk(const kws_handle* h, int B) {
    static const int env = experiment_int("KWS_CNN_CHUNK", 0);
    const size_t live = cnn_live_bytes(h);
    size_t cap = 1024;
    if (live) cap = std::min<size_t>(4096, std::max<size_t>(1024, ((size_t)256 << 20) / live));
    if (env > 0) cap = (size_t)env;
    const size_t fit = ((size_t)1 << 28) / std::max<size_t>(cnn_live_elems(h), 1);     // every activation tensor under 1 GiB (chunk_clips)
    const size_t cb = std::max<size_t>(1, std::min(fit, cap)), b = (size_t)std::max(B, 1);
    if (b <= cb) return (int)b;
    const size_t n = (b + cb - 1) / cb;
    return (int)std::min(cb, ((b + n - 1) / n + 63) / 64 * 64);
}

// Linears (flat Cin == 1 "convs") over a chunk of clips launch only B/256 workgroups; split K so the chip is filled.
int plan_ksplit(const ConvGeom& g, int nb, int steps) {
    if (!(g.kx_inner && g.ph == 0 && g.pw == 0)) return 1;
    const int mt = g.x_mt > 0 ? g.x_mt : g.MT;
    const long long wgs = (((long long)nb * g.Ho * g.Wo + 255) / 256) * ((g.mtiles + mt - 1) / mt);
    if (wgs >= 256 || steps < 64) return 1;
    static const int min_steps = std::max(1, experiment_int("KWS_KSPLIT_MIN_STEPS", 16));   // k-steps per split at least
    int ks = (int)std::min<long long>((1024 + wgs - 1) / wgs, steps / min_steps);
    return std::max(1, std::min(ks, 256));
}

size_t cnn_partial_bytes(const kws_handle* h, int cb) {
    size_t mx = 0;
    auto one = [&](const ConvLayer& L) {
        const int ks = std::max(plan_ksplit(L.g, cb, L.g.ksteps), plan_ksplit(L.g, cb, std::max(L.g.x_ksteps, 1)));
        if (ks > 1) mx = std::max(mx, (size_t)ks * cb * L.g.Cout * L.g.Ho * L.g.Wo * 4);
    };
    for (const auto& L : h->clin) one(L);
    if (h->cnn_band_R > 0 || h->cnn_cl1) one(h->clin0_cl);
    return align256(mx);
}

size_t cnn_lin_elems(const kws_handle* h) {   // widest Linear output per clip
    size_t mx = 1;
    for (const auto& L : h->clin) mx = std::max(mx, (size_t)L.g.Cout * L.g.Ho * L.g.Wo);
    return mx;
}

size_t act_bytes(const kws_handle* h, int B, int T) {
    if (h->plan == PLAN_RESNET) {
        if (use_fused(h, T)) return 0;   // (kws_forward's per-clip feature shifts live in a buffer of the handle: r8_shift)
        const ResnetShape s = resnet_shape(h, T);
        const size_t full = (size_t)s.C * s.T * s.F, small = (size_t)s.C * s.H * s.W;
        if (resnet_tiled(h, s)) {   // three channels-last tensors (conv_0 writes the first one directly)
            const size_t cl = resnet_cl_cells(h, s) * pad8(s.C);
            const int cb = tiled_chunk(h, s, B);
            return 3 * align256(cl * cb * 4) + align256((size_t)8 * s.T * s.F * 4 + 4096);
        }
        const int cb = chunk_clips(full, B);
        // + slack: padded channel blocks of the last clip read (and discard) up to 7 planes past a tensor's end
        return (s.pooled ? align256(full * cb * 4) : 0) + 2 * align256(small * cb * 4) + align256((size_t)8 * s.T * s.F * 4 + 4096);
    }
    if (h->plan == PLAN_CNN) {
        const int cb = cnn_chunk(h, B);
        // two flip buffers + split-K partials; with the side stream: a second buffer for what the Linears read, the gated pass's own flip pair, two small
        // buffers between Linears
        const size_t big = align256(cnn_live_elems(h) * cb * 4);      // (the tensors a call really writes: run_cnn carves the same)
        return (h->side ? 6 : 2) * big + (h->side ? 2 * align256(cnn_lin_elems(h) * cb * 4) : 0) + cnn_partial_bytes(h, cb) +
               align256((size_t)8 * h->d.time * h->d.freq * 4 + 4096);
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------- profiling helpers
int prof_mark(kws_handle* h, std::vector<hipEvent_t>& v, hipStream_t s) {
    if (!h->prof) return KWS_OK;
    if (h->ev_next == h->ev_pool.size()) {   // more calls between two reads than kws_profile_enable provided for
        hipEvent_t ne;
        HIP_TRY(hipEventCreate(&ne));
        h->ev_pool.push_back(ne);
    }
    hipEvent_t e = h->ev_pool[h->ev_next++];
    v.push_back(e);
    HIP_TRY(hipEventRecord(e, s));
    return KWS_OK;
}

// ---------------------------------------------------------------------------------------------- model dispatch
// ResNet, LDS-tiled plan: conv_0 (fp32 MFMA, NCHW out) -> pool / transpose to channels-last -> conv3x3_tile_kernel per
// layer -> mean + Linear.  The residual stream alternates between two buffers because a layer reads prev_x in the
// layout it was written in and writes the layout its consumer wants; odd layers write Y.
// fp16 range guard (kws_internal.h, RangeGate): modes whose operands are fp16 parts run every chunk twice -- the second pass
// on bf16 parts, gated by the device word the first pass sets when it stores a value fp16 cannot hold.
bool guarded_mode(const kws_handle* h, int terms) { return (terms == 6 || terms == 16) && h->lw_mode != LW_FP32; }
constexpr int RANGE_FREE_MODE = 66;

template <class Pass>
int run_guarded(kws_handle* h, int terms, hipStream_t s, Pass&& pass) {
    if (!guarded_mode(h, terms)) return pass(terms, RangeGate{nullptr, 0});
    unsigned* flag = h->range_flag.as<unsigned>();
    HIP_TRY(hipMemsetAsync(flag, 0, sizeof(unsigned), s));
    int rc = pass(terms, RangeGate{flag, 0});
    if (rc) return rc;
    return pass(RANGE_FREE_MODE, RangeGate{flag, 1});
}

int run_resnet_tiled(kws_handle* h, const float* feat, int B, int T, float* logits, char* ws, hipStream_t s) {
    const kws_model_desc& d = h->d;
    const ResnetShape sh = resnet_shape(h, T);
    const int C = sh.C, cp = pad8(C);
    const size_t cl = resnet_cl_cells(h, sh) * cp;
    const int cb = tiled_chunk(h, sh, B);
    float* X = (float*)ws; ws += align256(cl * cb * 4);
    float* X2 = (float*)ws; ws += align256(cl * cb * 4);
    float* Y = (float*)ws;
    int rc;
    const bool note = h->plan_detail_T != T;     // (once per clip length: the launch sequence depends on nothing else)
    std::string detail = "conv0";
    auto note_layers = [&](const char* what, int i0, int n) {
        if (!note) return;
        detail += std::string(" ") + what + "(";
        for (int u = 0; u < n; ++u) detail += (u ? "," : "") + std::to_string(i0 + u);
        detail += ")";
    };
    for (int b0 = 0; b0 < B; b0 += cb) {
        const int nb = std::min(cb, B - b0);
        auto pass = [&](int terms, RangeGate rg) -> int {
            const bool first_pass = b0 == 0 && !rg.gated;
            // conv_0 + ReLU (+ AvgPool) in plain fp32, straight into the channels-last tensor; with single-term products (the
            // `bf16` / `fp16` dtypes) the tensors between the layers hold the 16-bit operand type itself
            int m_f16, m_terms;
            decode_mode(terms, m_f16, m_terms);
            const int clt = m_terms == 1 ? (m_f16 ? CL_F16 : CL_BF16) : CL_F32;
            HIP_TRY(launch_conv0_cl(feat + (size_t)b0 * sh.T * sh.F, h->rconv[0].w9cl.as<float>(), X, clt, nb, sh.T, sh.F,
                                    sh.pooled ? d.pool_h : 1, sh.pooled ? d.pool_w : 1, cp, s, rg));
            float* xc = X;
            float* xn = X2;
            int ld_x = 0;   // layout of xc
            for (int i = 1; i <= d.n_layers; ++i) {
                const int ld_in = ilog2(resnet_dilation(d, i));
                const int ld_out = i < d.n_layers ? ilog2(resnet_dilation(d, i + 1)) : 0;
                const int dd = 1 << ld_in;
                const bool even = (i % 2) == 0;
                // a run of exactly three layers of one dilation (res15: (4,5,6) (7,8,9) (10,11,12)), 16-bit tensors: one kernel for the run
                // (t3_triple == 2, A/B runs: any three consecutive layers of one dilation)
                const int Hs_i = (sh.H + dd - 1) / dd, Ws_i = (sh.W + dd - 1) / dd;
                // (r5) any three consecutive layers of one dilation as ONE persistent weight-stationary stream (conv3x3_stream.hip); bit-identical to the forms below
                // (measured, res15 `bf16`, B = 4 096, one machine: every run a stream 12.54 ms, odd-first runs only 13.23, tile kernels only 14.03; t3_stream == 3: odd-first runs only)
                // A stream gives every CU one span of the launch's cells and pays ~11 steps of 64 cells of pipeline fill per span: below ~7 700 cells per CU
                // the tile kernels win (res26, 1.02 M cells per 1 024 clips: 1.90 against 1.76 ms; 2 048 clips: 3.40 / 3.36; 4 096: 6.05 / 6.12); t3_stream == 2: any size (tests)
                const bool stream_size = h->t3_stream == 2 || (long long)nb * dd * dd * Hs_i * Ws_i >= 7680LL * h->n_cu;
                if (m_terms == 1 && h->t3_stream && stream_size && (!even || h->t3_stream != 3) && i + 2 <= d.n_layers && resnet_dilation(d, i + 1) == dd &&
                    resnet_dilation(d, i + 2) == dd && conv3x3_stream_supported(C, Ws_i)) {
                    const int ld_out3 = i + 2 < d.n_layers ? ilog2(resnet_dilation(d, i + 3)) : 0;
                    std::unique_ptr<ConvLayer::PosTab>& pt3 = h->rconv[i].postabs3[T];
                    if (!pt3) {
                        std::vector<int> tab;
                        std::unique_ptr<ConvLayer::PosTab> fresh(new ConvLayer::PosTab);
                        build_tile_conv_table(sh.H, sh.W, ld_in, ld_out3, ld_x, tab, fresh->cpc[0], fresh->cpc[1], fresh->cpc[2]);
                        if ((rc = fresh->mem.upload(tab.data(), tab.size() * sizeof(int)))) { h->rconv[i].postabs3.erase(T); return rc; }
                        pt3 = std::move(fresh);
                    }
                    StreamConvParams sp{};
                    sp.first_even = even ? 1 : 0;
                    sp.n_layers = 3;
                    sp.in = even ? Y : xc;
                    sp.res = even ? xc : nullptr;
                    sp.out = even ? xn : Y;
                    sp.out2 = even ? nullptr : xn;
                    sp.f16 = m_f16;
                    for (int u = 0; u < 3; ++u) {
                        const ConvLayer& L = h->rconv[i + u];
                        sp.apk[u] = m_f16 ? L.apk_t3h.as<unsigned short>() : L.apk16.as<unsigned short>();
                        sp.inv_scale[u] = m_f16 ? 1.0f / L.t3h_scale : 1.0f;
                        sp.border[u] = L.has_border ? L.border_pad.as<float>() : nullptr;
                    }
                    sp.B = nb; sp.Ws = Ws_i;
                    sp.total = nb * dd * dd * Hs_i * Ws_i;
                    sp.rg = rg;
                    sp.postab = pt3->mem.as<int>();
                    sp.cpc_in = pt3->cpc[0]; sp.cpc_out = pt3->cpc[1]; sp.cpc_res = pt3->cpc[2];
                    HIP_TRY(launch_conv3x3_stream(sp, C, h->n_cu, s));
                    if (first_pass) note_layers("stream", i, 3);
                    std::swap(xc, xn);
                    ld_x = even ? ld_out3 : ld_in;
                    i += 2;
                    continue;
                }
                if (m_terms == 1 && h->t3_pair && h->t3_triple && i + 2 <= d.n_layers && resnet_dilation(d, i + 1) == dd &&
                    resnet_dilation(d, i + 2) == dd &&
                    (h->t3_triple == 2 || ((i == 1 || resnet_dilation(d, i - 1) != dd) && (i + 3 > d.n_layers || resnet_dilation(d, i + 3) != dd))) &&
                    dd * dd * Hs_i * Ws_i >= 16 && conv3x3_triple_supported(C, Ws_i, even)) {
                    const int ld_out3 = i + 2 < d.n_layers ? ilog2(resnet_dilation(d, i + 3)) : 0;
                    std::unique_ptr<ConvLayer::PosTab>& pt3 = h->rconv[i].postabs3[T];
                    if (!pt3) {
                        std::vector<int> tab;
                        std::unique_ptr<ConvLayer::PosTab> fresh(new ConvLayer::PosTab);
                        build_tile_conv_table(sh.H, sh.W, ld_in, ld_out3, ld_x, tab, fresh->cpc[0], fresh->cpc[1], fresh->cpc[2]);
                        if ((rc = fresh->mem.upload(tab.data(), tab.size() * sizeof(int)))) { h->rconv[i].postabs3.erase(T); return rc; }
                        pt3 = std::move(fresh);
                    }
                    TripleConvParams tp3{};
                    tp3.first_even = even ? 1 : 0;
                    tp3.in = even ? Y : xc;
                    tp3.res = even ? xc : nullptr;
                    tp3.out = even ? xn : Y;
                    tp3.out2 = even ? nullptr : xn;
                    tp3.f16 = m_f16;
                    for (int u = 0; u < 3; ++u) {
                        const ConvLayer& L = h->rconv[i + u];
                        tp3.apk[u] = m_f16 ? L.apk_t3h.as<unsigned short>() : L.apk16.as<unsigned short>();
                        tp3.inv_scale[u] = m_f16 ? 1.0f / L.t3h_scale : 1.0f;
                        tp3.border[u] = L.has_border ? L.border_pad.as<float>() : nullptr;
                    }
                    tp3.B = nb; tp3.H = sh.H; tp3.W = sh.W; tp3.Cout = C;
                    tp3.ld = ld_in; tp3.ld_out = ld_out3;
                    tp3.Hs = Hs_i; tp3.Ws = Ws_i;
                    tp3.total = nb * dd * dd * Hs_i * Ws_i;
                    tp3.rg = rg;
                    tp3.postab = pt3->mem.as<int>();
                    tp3.cpc_in = pt3->cpc[0]; tp3.cpc_out = pt3->cpc[1]; tp3.cpc_res = pt3->cpc[2];
                    static const int triple_dbg = experiment_int("KWS_T3_DEBUG", 0);
                    tp3.debug = triple_dbg;
                    // KWS_T3_TIMING=<file> (timing builds only): phase stamps of the run that STARTS at layer KWS_T3_TIMING_LAYER, first chunk
                    static const char* t3x_file = experiment_str("KWS_T3_TIMING");
                    static const int t3x_layer = experiment_int("KWS_T3_TIMING_LAYER", 2);
                    static DevMem t3x_buf;
                    const bool t3x_this = t3x_file && i == t3x_layer && b0 == 0 && !rg.gated;
                    if (t3x_this) {
                        std::vector<unsigned long long> z((size_t)8192 * 4 * 12, 0ull);
                        if ((rc = t3x_buf.upload(z.data(), z.size() * 8))) return rc;
                        tp3.dbg_ts = t3x_buf.as<unsigned long long>();
                    }
                    HIP_TRY(launch_conv3x3_triple(tp3, C, s));
                    if (t3x_this) {
                        std::vector<unsigned long long> z((size_t)8192 * 4 * 12);
                        HIP_TRY(hipStreamSynchronize(s));
                        HIP_TRY(hipMemcpy(z.data(), t3x_buf.p, z.size() * 8, hipMemcpyDeviceToHost));
                        if (FILE* f = std::fopen(t3x_file, "wb")) {
                            std::fwrite(z.data(), 8, z.size(), f);
                            std::fclose(f);
                        }
                    }
                    if (first_pass) note_layers("triple", i, 3);
                    // a even: x_{a+2} went to xn in the next run's layout; a odd: x_{a+1} went to xn in THIS run's layout (y_{a+2} is in Y)
                    std::swap(xc, xn);
                    ld_x = even ? ld_out3 : ld_in;
                    i += 2;
                    continue;
                }
                // odd i and i + 1 with the same dilation, 16-bit tensors: one kernel for both (y_i never leaves the CU)
                if (!even && m_terms == 1 && h->t3_pair && i + 1 <= d.n_layers && resnet_dilation(d, i + 1) == dd &&
                    conv3x3_pair_tile(C, (sh.W + dd - 1) / dd) > 0) {
                    const int ld_out2 = i + 1 < d.n_layers ? ilog2(resnet_dilation(d, i + 2)) : 0;
                    std::unique_ptr<ConvLayer::PosTab>& pt2 = h->rconv[i + 1].postabs[T];
                    if (!pt2) {
                        std::vector<int> tab;
                        std::unique_ptr<ConvLayer::PosTab> fresh(new ConvLayer::PosTab);
                        build_tile_conv_table(sh.H, sh.W, ld_in, ld_out2, ld_x, tab, fresh->cpc[0], fresh->cpc[1], fresh->cpc[2]);
                        if ((rc = fresh->mem.upload(tab.data(), tab.size() * sizeof(int)))) { h->rconv[i + 1].postabs.erase(T); return rc; }
                        pt2 = std::move(fresh);
                    }
                    PairConvParams pp{};
                    pp.in = xc; pp.out = xn;
                    pp.f16 = m_f16;
                    pp.apk_a = m_f16 ? h->rconv[i].apk_t3h.as<unsigned short>() : h->rconv[i].apk16.as<unsigned short>();
                    pp.apk_b = m_f16 ? h->rconv[i + 1].apk_t3h.as<unsigned short>() : h->rconv[i + 1].apk16.as<unsigned short>();
                    pp.inv_scale_a = m_f16 ? 1.0f / h->rconv[i].t3h_scale : 1.0f;
                    pp.inv_scale_b = m_f16 ? 1.0f / h->rconv[i + 1].t3h_scale : 1.0f;
                    pp.border_a = h->rconv[i].has_border ? h->rconv[i].border_pad.as<float>() : nullptr;
                    pp.border_b = h->rconv[i + 1].has_border ? h->rconv[i + 1].border_pad.as<float>() : nullptr;
                    pp.B = nb; pp.H = sh.H; pp.W = sh.W; pp.Cout = C;
                    pp.ld = ld_in; pp.ld_out = ld_out2;
                    pp.Hs = (sh.H + dd - 1) / dd; pp.Ws = (sh.W + dd - 1) / dd;
                    pp.total = nb * dd * dd * pp.Hs * pp.Ws;
                    pp.rg = rg;
                    pp.postab = pt2->mem.as<int>();
                    pp.cpc_in = pt2->cpc[0]; pp.cpc_out = pt2->cpc[1];
                    static const int pair_dbg = experiment_int("KWS_T3_DEBUG", 0);
                    pp.debug = pair_dbg;
                    HIP_TRY(launch_conv3x3_pair(pp, C, s));
                    if (first_pass) note_layers("pair", i, 2);
                    std::swap(xc, xn);
                    ld_x = ld_out2;
                    ++i;
                    continue;
                }
                TileConvParams tp{};
                tp.in = even ? Y : xc;
                tp.out = even ? xn : Y;
                tp.res = even ? xc : nullptr;
                decode_mode(terms, tp.f16, tp.terms);
                tp.apk16 = tp.f16 ? h->rconv[i].apk_t3h.as<unsigned short>() : h->rconv[i].apk16.as<unsigned short>();
                tp.inv_scale = tp.f16 ? 1.0f / h->rconv[i].t3h_scale : 1.0f;
                tp.border = h->rconv[i].has_border ? h->rconv[i].border_pad.as<float>() : nullptr;
                tp.B = nb; tp.H = sh.H; tp.W = sh.W; tp.Cout = C;
                tp.ld_in = ld_in; tp.ld_out = ld_out; tp.ld_res = ld_x;
                tp.Hs = (sh.H + dd - 1) / dd; tp.Ws = (sh.W + dd - 1) / dd;
                tp.total = nb * dd * dd * tp.Hs * tp.Ws;
                tp.rg = rg;
                std::unique_ptr<ConvLayer::PosTab>& pt = h->rconv[i].postabs[T];
                if (!pt) {   // the per-cell table of this layer's geometry: built and uploaded (blocking) on the FIRST call with this clip length
                    std::vector<int> tab;
                    std::unique_ptr<ConvLayer::PosTab> fresh(new ConvLayer::PosTab);
                    build_tile_conv_table(sh.H, sh.W, ld_in, ld_out, ld_x, tab, fresh->cpc[0], fresh->cpc[1], fresh->cpc[2]);
                    if ((rc = fresh->mem.upload(tab.data(), tab.size() * sizeof(int)))) { h->rconv[i].postabs.erase(T); return rc; }
                    pt = std::move(fresh);
                }
                tp.postab = pt->mem.as<int>();
                tp.cpc_in = pt->cpc[0]; tp.cpc_out = pt->cpc[1]; tp.cpc_res = pt->cpc[2];
                static const int t3_dbg = experiment_int("KWS_T3_DEBUG", 0);
                tp.debug = t3_dbg;
                // KWS_T3_TIMING=<file> (timing builds only): phase stamps of layer KWS_T3_TIMING_LAYER (default 2) of the first chunk
                static const char* t3_file = experiment_str("KWS_T3_TIMING");
                static const int t3_layer = experiment_int("KWS_T3_TIMING_LAYER", 2);
                static DevMem t3_buf;
                const bool t3_this = t3_file && i == t3_layer && b0 == 0 && !rg.gated;
                if (t3_this) {
                    std::vector<unsigned long long> z((size_t)8192 * 4 * 8, 0ull);
                    if ((rc = t3_buf.upload(z.data(), z.size() * 8))) return rc;
                    tp.dbg_ts = t3_buf.as<unsigned long long>();
                }
                const bool stream1 = m_terms == 1 && h->t3_stream && (h->t3_stream == 2 || (long long)tp.total >= 7680LL * h->n_cu) && !even && !t3_this && conv3x3_stream_supported(C, tp.Ws);   // (an even single layer takes its residual from memory: the tile kernel)
                if (stream1) {   // (r5) a single layer as a persistent weight-stationary stream: the same tensors, table and bits
                    StreamConvParams sp{};
                    sp.first_even = even ? 1 : 0;
                    sp.n_layers = 1;
                    sp.in = tp.in; sp.res = tp.res; sp.out = tp.out; sp.out2 = nullptr;
                    sp.f16 = tp.f16;
                    sp.apk[0] = tp.apk16; sp.inv_scale[0] = tp.inv_scale; sp.border[0] = tp.border;
                    sp.B = nb; sp.Ws = tp.Ws; sp.total = tp.total; sp.rg = rg;
                    sp.postab = tp.postab;
                    sp.cpc_in = tp.cpc_in; sp.cpc_out = tp.cpc_out; sp.cpc_res = tp.cpc_res;
                    HIP_TRY(launch_conv3x3_stream(sp, C, h->n_cu, s));
                } else {
                    HIP_TRY(launch_conv3x3_tile(tp, C, s));
                }
                if (first_pass) note_layers(stream1 ? "stream" : "conv", i, 1);
                if (t3_this) {
                    std::vector<unsigned long long> z((size_t)8192 * 4 * 8);
                    HIP_TRY(hipStreamSynchronize(s));
                    HIP_TRY(hipMemcpy(z.data(), t3_buf.p, z.size() * 8, hipMemcpyDeviceToHost));
                    if (FILE* f = std::fopen(t3_file, "wb")) {
                        std::fwrite(z.data(), 8, z.size(), f);
                        std::fclose(f);
                    }
                }
                if (even) {
                    std::swap(xc, xn);
                    ld_x = ld_out;
                }
            }
            const float* fin = (d.n_layers % 2 == 0) ? xc : Y;
            HIP_TRY(launch_mean_linear_cl(fin, clt, logits + (size_t)b0 * d.n_labels, nb, C, cp, sh.H * sh.W,
                                          h->bn_mean.as<float>() + (size_t)(d.n_layers - 1) * C,
                                          h->bn_rstd.as<float>() + (size_t)(d.n_layers - 1) * C, h->out_w.as<float>(),
                                          h->out_b.as<float>(), d.n_labels, s, rg));
            return KWS_OK;
        };
        if ((rc = run_guarded(h, dtype_terms(d.dtype), s, pass))) return rc;
    }
    if (note && B > 0) {
        h->plan_detail = detail + " mean+linear";
        h->plan_detail_T = T;
    }
    return KWS_OK;
}

int run_resnet_layerwise(kws_handle* h, const float* feat, int B, int T, float* logits, char* ws, hipStream_t s) {
    const kws_model_desc& d = h->d;
    const ResnetShape sh = resnet_shape(h, T);
    if (sh.H < 1 || sh.W < 1) return fail(KWS_EINVAL, "feature map smaller than the pooling window");
    if (resnet_tiled(h, sh)) return run_resnet_tiled(h, feat, B, T, logits, ws, s);
    const size_t full = (size_t)sh.C * sh.T * sh.F, small = (size_t)sh.C * sh.H * sh.W;
    const int cb = chunk_clips(full, B);
    float* bufA = nullptr;
    if (sh.pooled) { bufA = (float*)ws; ws += align256(full * cb * 4); }
    float* X = (float*)ws; ws += align256(small * cb * 4);
    float* Y = (float*)ws;
    const int C = sh.C;
    int rc;
    for (int b0 = 0; b0 < B; b0 += cb) {
        const int nb = std::min(cb, B - b0);
        auto pass = [&](int terms, RangeGate rg) -> int {
            int rcp;
            // conv_0 + ReLU (+ AvgPool); the features are split into fp16 parts too: note what fp16 cannot hold
            if (rg.flag && !rg.gated) HIP_TRY(launch_range_check(feat + (size_t)b0 * sh.T * sh.F, (long long)nb * sh.T * sh.F, s, rg));
            ConvGeom g0 = h->rconv[0].g;
            set_spatial(g0, nb, sh.T, sh.F);
            ConvArgs a0{feat + (size_t)b0 * sh.T * sh.F, sh.pooled ? bufA : X, h->rconv[0].apk.as<float>(), nullptr, nullptr, nullptr, nullptr, rg};
            if ((rcp = launch_layer(h->rconv[0], g0, a0, s, terms))) return rcp;
            if (sh.pooled) HIP_TRY(launch_pool(bufA, X, nb * C, sh.T, sh.F, d.pool_h, d.pool_w, 0, s, rg));
            // conv_i: odd i writes Y from X, even i accumulates into X from Y (prev_x lives in X)
            for (int i = 1; i <= d.n_layers; ++i) {
                ConvGeom g = h->rconv[i].g;
                set_spatial(g, nb, sh.H, sh.W);
                const bool even = (i % 2) == 0;
                ConvArgs a{even ? Y : X, even ? X : Y, h->rconv[i].apk.as<float>(), nullptr, nullptr, nullptr,
                           h->rconv[i].has_border ? h->rconv[i].border.as<float>() : nullptr, rg};
                if ((rcp = launch_layer(h->rconv[i], g, a, s, terms))) return rcp;
            }
            const float* fin = (d.n_layers % 2 == 0) ? X : Y;
            HIP_TRY(launch_mean_linear(fin, logits + (size_t)b0 * d.n_labels, nb, C, sh.H * sh.W,
                                       h->bn_mean.as<float>() + (size_t)(d.n_layers - 1) * C,
                                       h->bn_rstd.as<float>() + (size_t)(d.n_layers - 1) * C, h->out_w.as<float>(),
                                       h->out_b.as<float>(), d.n_labels, s, rg));
            return KWS_OK;
        };
        if ((rc = run_guarded(h, dtype_terms(d.dtype), s, pass))) return rc;
    }
    return KWS_OK;
}

int launch_conv_auto(const ConvLayer& L, ConvGeom g, ConvArgs a, int nb, float* partial, size_t partial_bytes, hipStream_t s,
                     int terms) {
    g.B = nb;
    const int steps = L.use_x ? g.x_ksteps : g.ksteps;
    g.ksplit = plan_ksplit(g, nb, steps);
    if (g.ksplit > 1) {
        g.ksteps_split = (steps + g.ksplit - 1) / g.ksplit;
        g.ksplit = (steps + g.ksteps_split - 1) / g.ksteps_split;
        const long long total = (long long)nb * g.Cout * g.Ho * g.Wo;
        if (!partial || (size_t)g.ksplit * total * 4 > partial_bytes) g.ksplit = 1;   // no room: fall back to one pass
        else {
            a.partial = partial;
            int rc = launch_layer(L, g, a, s, terms);
            if (rc) return rc;
            HIP_TRY(launch_splitk_reduce(partial, a.out, a.bias, g.ksplit, total, g.Cout, g.Ho * g.Wo, g.relu, s, a.rg));
            return KWS_OK;
        }
    }
    g.ksteps_split = steps;
    return launch_layer(L, g, a, s, terms);
}

// The band plan needs fp16-part operands (modes 6 / 16); the other dtypes and the range-free second pass keep the generic kernels
bool cnn_band_plan(const kws_handle* h, int mode) {
    int f16, terms;
    decode_mode(mode, f16, terms);
    return h->cnn_band_R > 0 && f16 && (h->cnn_band_parts == 2 || terms == 1) && h->cconv[0].use_x && h->clin0_cl.use_x && h->cconv[1].apk_band.p != nullptr;
}
// single-conv models: conv_0 from the LDS image (conv_in1.hip), channels-last cells straight into the permuted Linear
bool cnn_in1_plan(const kws_handle* h, int mode) {
    int f16, terms;
    decode_mode(mode, f16, terms);
    return h->cnn_cl1 && f16 && h->clin0_cl.use_x && h->cconv[0].apk_in1[0].p != nullptr;
}

int run_cnn(kws_handle* h, const float* feat, int B, int T, float* logits, char* ws, hipStream_t s_main) {
    const kws_model_desc& d = h->d;
    if (T != d.time) return fail(KWS_EINVAL, "CNN was built for a different number of frames (config[\"time\"])");
    const int cb = cnn_chunk(h, B);
    const size_t big = align256(cnn_live_elems(h) * cb * 4), part_bytes = cnn_partial_bytes(h, cb);
    char* w = ws;
    auto carve = [&](size_t n) { float* q = (float*)w; w += n; return q; };
    float* P = carve(big);
    float* Q = carve(big);
    float* part = carve(part_bytes);
    // (r5) two streams.  Per chunk the convolutions fill the chip (conv_in1 + conv_band: 185 of 234 us on cnn-trad-pool2 `fp16`) and the rest does not: a Linear of
    // ~200 workgroups, its split-K reduce, the range guard's gated second pass (four launches that read a flag and return).  With more than one chunk in the
    // call that tail runs on the handle's own stream while the caller's stream goes on with the next chunk's convolutions: fork after conv_1 (ev_fork), join
    // three chunks later -- when the buffer the Linear reads and the chunk's flag word come round again (ev_join; with a ring of two the side stream, whose
    // gated launches find no free wave slot beside the persistent conv_cols_kernel and finish right behind it, held the caller's stream up by ~10 us per
    // chunk) -- and at the end of the call.  Same kernels on
    // the same operands: the logits are bit-identical to the one-stream form (KWS_CNN_STREAMS=0; tested).  Under stream capture the side stream joins the
    // capture at the first fork and has left it at the last join.
    const int terms0 = dtype_terms(d.dtype);
    const bool piped = h->side && B > cb && (cnn_band_plan(h, terms0) || cnn_in1_plan(h, terms0));
    constexpr int RING = kws_handle::CNN_RING;
    float *P1 = nullptr, *P3 = nullptr, *P2 = nullptr, *Q2 = nullptr, *T0 = nullptr, *T1 = nullptr;
    if (h->side) {
        P1 = carve(big); P3 = carve(big); P2 = carve(big); Q2 = carve(big);
        const size_t lin = align256(cnn_lin_elems(h) * cb * 4);
        T0 = carve(lin); T1 = carve(lin);
    }
    const bool guarded = guarded_mode(h, terms0);
    if (piped && guarded) HIP_TRY(hipMemsetAsync(h->range_flag.as<unsigned>(), 0, RING * sizeof(unsigned), s_main));
    int k = 0;
    for (int b0 = 0; b0 < B; b0 += cb, ++k) {
        const int nb = std::min(cb, B - b0);
        float* head_out = k % RING == 0 ? P : k % RING == 1 ? P1 : P3;   // (two-stream form) what this chunk's first Linear reads
        auto pass = [&](int terms, RangeGate rg) -> int {
            const bool on_side = piped && rg.gated;   // the gated pass of a two-stream chunk: the side stream, its own flip pair
            hipStream_t s = on_side ? h->side : s_main, st = piped ? h->side : s_main;   // convolutions / Linears
            float *FP = on_side ? P2 : P, *FQ = on_side ? Q2 : Q;
            auto other = [&](const float* c) -> float* { return c == FP ? FQ : FP; };
            auto lin_dst = [&](const float* c) -> float* { return piped && !rg.gated ? (c == T0 ? T1 : T0) : other(c); };
            const float* cur = feat + (size_t)b0 * d.time * d.freq;
            int m_f16, m_terms;
            decode_mode(terms, m_f16, m_terms);
            size_t first_lin = 0;
            const bool band = cnn_band_plan(h, terms), cl1 = cnn_in1_plan(h, terms);
            // the feature maps come from the caller: the first pass checks them like any stored activation -- conv_in1_kernel does that while it
            // stages the clip (NaN-aware); every other conv_0 gets range_check_kernel in front
            if (rg.flag && !rg.gated && !((band || cl1) && h->cnn_in1)) HIP_TRY(launch_range_check(cur, (long long)nb * d.time * d.freq, s, rg));
            // single-conv models in the `fp16` dtype: conv_0's cells go straight into the first Linear, which rounds them to fp16 itself (see lin_f16 below)
            const bool cl1_f16 = cl1 && !band && m_f16 && m_terms == 1 && h->lin_in_f16 &&
                                 ((long long)(h->cconv[0].g.Ho / std::max(d.pool_kh[0], 1)) * (h->cconv[0].g.Wo / std::max(d.pool_kw[0], 1)) * h->cnn_cp[0]) % 8 == 0;
            if (band || cl1) {
                float* c0_out = piped && !band ? head_out : Q;
                // conv_0 (+ fused MaxPool) writes channels-last cells -- fp32, or fp16 with single-term products: from an LDS image
                // of the clip (conv_in1.hip) where the layer fits it, else through the generic kernel's channels-last epilogue
                int rcb;
                if (h->cnn_in1) {
                    const ConvGeom& g0 = h->cconv[0].g;
                    In1ConvParams ip{};
                    ip.feat = cur; ip.out = c0_out;
                    ip.apk = h->cconv[0].apk_in1[m_terms == 1 ? 1 : 0].as<unsigned short>();
                    ip.bias = h->cconv[0].bias.as<float>();
                    ip.B = nb; ip.T = d.time; ip.F = d.freq; ip.Cout = g0.Cout; ip.Cp = h->cnn_cp[0]; ip.mtiles = g0.mtiles;
                    ip.kh = g0.kh; ip.sh = g0.sh; ip.sw = g0.sw; ip.ph = d.pool_kh[0]; ip.pw = d.pool_kw[0];
                    ip.Hq = g0.Ho / d.pool_kh[0]; ip.Wq = g0.Wo / d.pool_kw[0];
                    ip.terms = m_terms; ip.inv_scale = 1.0f / h->cconv[0].x_scale; ip.relu = 1; ip.out_f16 = (band && m_terms == 1) || cl1_f16; ip.rg = rg;
                    HIP_TRY(launch_conv_in1(ip, s));
                } else {
                    ConvGeom g0 = h->cconv[0].g;
                    g0.B = nb;
                    if (d.pool_kh[0] * d.pool_kw[0] >= 2) {
                        g0.pool_h = d.pool_kh[0];
                        g0.pool_w = d.pool_kw[0];
                    }
                    g0.out_cl = ((band && m_terms == 1) || cl1_f16) ? 2 : 1;   // fp16 cells for conv_band.hip, or for a Linear that would round them to fp16 anyway
                    g0.out_cp = h->cnn_cp[0];
                    ConvArgs a0{cur, c0_out, h->cconv[0].apk.as<float>(), nullptr, h->cconv[0].bias.as<float>(), nullptr, nullptr, rg};
                    if ((rcb = launch_layer(h->cconv[0], g0, a0, s, terms))) return rcb;
                }
                const float* lin_in = c0_out;
                bool lin_f16 = cl1_f16;
                if (band) {
                const ConvGeom& g1 = h->cconv[1].g;
                BandConvParams bp{};
                bp.in = Q; bp.out = piped ? head_out : P;
                bp.apk = h->cconv[1].apk_band.as<unsigned short>();
                bp.bias = h->cconv[1].bias.as<float>();
                bp.B = nb; bp.H = g1.H; bp.W = g1.W; bp.Cpi = h->cnn_cp[0];
                bp.Ho = g1.Ho; bp.Wo = g1.Wo; bp.Cout = g1.Cout; bp.Cpo = h->cnn_cp[1];
                bp.kh = g1.kh; bp.kw = g1.kw; bp.ksteps = (g1.kh * g1.kw * (bp.Cpi / 8) + 3) / 4;
                bp.R = h->cnn_band_R; bp.nbands = (g1.Ho + bp.R - 1) / bp.R;
                bp.Wl = h->cnn_band.Wl; bp.PS = h->cnn_band.PS; bp.ntiles = h->cnn_band.ntiles;
                bp.postab = h->cnn_band_tab.as<int>();
                bp.terms = m_terms; bp.inv_scale = 1.0f / h->cconv[1].band_scale; bp.relu = 1; bp.rg = rg;
                // (r4) single-term fp16 products: the Linear behind conv_1 rounds its input to fp16 anyway, so conv_1 stores that fp16 value (same bits out,
                // half the bytes both ways) -- where the Linear takes its eight k-slots as one 16-byte load (rows of a multiple of eight cells' channels)
                lin_f16 = m_f16 && m_terms == 1 && h->lin_in_f16 && ((long long)g1.Ho * g1.Wo * h->cnn_cp[1]) % 8 == 0;
                bp.out_f16 = lin_f16 ? 1 : 0;
                // KWS_BAND_TIMING=<file> (timing builds only): phase stamps of the first chunk's band kernel (tools/band_phases.py)
                static const char* band_file = experiment_str("KWS_BAND_TIMING");
                static DevMem band_buf;
                const bool band_this = band_file && b0 == 0 && !rg.gated;
                if (band_this) {
                    std::vector<unsigned long long> z((size_t)8192 * 4 * 8, 0ull);
                    int rcz = band_buf.upload(z.data(), z.size() * 8);
                    if (rcz) return rcz;
                    bp.dbg_ts = band_buf.as<unsigned long long>();
                }
                const bool cols_this = h->cnn_cols && m_terms == 1 && h->cconv[1].apk_cols.p;
                const int detail_code = -2 - ((h->cnn_in1 ? 1 : 0) | (cols_this ? 2 : 0) | (piped ? 4 : 0));      // (plan_detail_T of a cnn handle: which of the eight texts it holds)
                if (b0 == 0 && !rg.gated && h->plan_detail_T != detail_code) {
                    h->plan_detail = std::string(h->cnn_in1 ? "conv_in1" : "conv_0") + (cols_this ? " conv_cols" : " conv_band") + " linear" + (piped ? " | two streams" : "");
                    h->plan_detail_T = detail_code;
                }
                if (cols_this) {
                    ColsConvParams cp{};
                    cp.in = reinterpret_cast<const unsigned short*>(bp.in); cp.out = bp.out;
                    cp.apk = h->cconv[1].apk_cols.as<unsigned short>(); cp.bias = bp.bias;
                    cp.B = nb; cp.H = g1.H; cp.W = g1.W; cp.Cpi = bp.Cpi; cp.Ho = g1.Ho; cp.Wo = g1.Wo; cp.Cout = g1.Cout; cp.Cpo = bp.Cpo;
                    cp.kh = g1.kh; cp.nbands = (g1.Ho + 15) / 16;
                    cp.inv_scale = bp.inv_scale; cp.relu = 1; cp.out_f16 = bp.out_f16; cp.rg = rg; cp.dbg_ts = bp.dbg_ts;
                    cp.queue = h->range_flag.as<unsigned>() + 16;   // (the fused res8 kernel's words: no cnn handle runs that kernel)
                    HIP_TRY(launch_conv_cols(cp, h->n_cu, s));
                } else
                HIP_TRY(launch_conv_band(bp, s));
                if (band_this) {
                    std::vector<unsigned long long> z((size_t)8192 * 4 * 8);
                    HIP_TRY(hipStreamSynchronize(s));
                    HIP_TRY(hipMemcpy(z.data(), band_buf.p, z.size() * 8, hipMemcpyDeviceToHost));
                    if (FILE* f = std::fopen(band_file, "wb")) {
                        std::fwrite(z.data(), 8, z.size(), f);
                        std::fclose(f);
                    }
                }
                lin_in = bp.out;
                }
                if (piped) {
                    HIP_TRY(hipEventRecord(h->ev_fork[k % RING], s));
                    HIP_TRY(hipStreamWaitEvent(st, h->ev_fork[k % RING], 0));
                }
                ConvGeom gl = h->clin0_cl.g;
                gl.B = nb;
                gl.in_f16 = lin_f16 ? 1 : 0;
                const bool last = h->clin.size() == 1;
                float* dst = last ? logits + (size_t)b0 * d.n_labels : lin_dst(lin_in);
                ConvArgs al{lin_in, dst, h->clin0_cl.apk.as<float>(), nullptr, h->clin[0].bias.as<float>(), nullptr, nullptr, rg};
                rcb = launch_conv_auto(h->clin0_cl, gl, al, nb, part, part_bytes, st, terms);
                if (rcb) return rcb;
                cur = dst;
                first_lin = 1;
            }
            for (int i = 0; i < (first_lin ? 0 : d.n_conv); ++i) {
                ConvGeom g = h->cconv[i].g;
                g.B = nb;
                float* conv_out_buf = other(cur);
                ConvArgs a{cur, conv_out_buf, h->cconv[i].apk.as<float>(), nullptr, h->cconv[i].bias.as<float>(), nullptr, nullptr, rg};
                const int members = d.pool_kh[i] * d.pool_kw[i];
                // MaxPool windows are reduced in the conv's accumulators (bf16x6 kernel, up to four members per pass over K);
                // 1 x 1 pools (every pool_1 of the shipped configs) are the identity
                if (h->cconv[i].use_x && members >= 2 && members <= 16) {
                    g.pool_h = d.pool_kh[i];
                    g.pool_w = d.pool_kw[i];
                }
                int rcc = launch_layer(h->cconv[i], g, a, s, terms);
                if (rcc) return rcc;
                if (g.pool_h || members == 1) {
                    cur = conv_out_buf;
                    continue;
                }
                float* pooled = other(conv_out_buf);
                HIP_TRY(launch_pool(conv_out_buf, pooled, nb * g.Cout, g.Ho, g.Wo, d.pool_kh[i], d.pool_kw[i], 1, s, rg));
                cur = pooled;
            }
            for (size_t i = first_lin; i < h->clin.size(); ++i) {
                ConvGeom g = h->clin[i].g;
                g.B = nb;
                const bool last = i + 1 == h->clin.size();
                float* dst = last ? logits + (size_t)b0 * d.n_labels : lin_dst(cur);
                ConvArgs a{cur, dst, h->clin[i].apk.as<float>(), nullptr, h->clin[i].bias.as<float>(), nullptr, nullptr, rg};
                int rcl = launch_conv_auto(h->clin[i], g, a, nb, part, part_bytes, st, terms);
                if (rcl) return rcl;
                cur = dst;
            }
            return KWS_OK;
        };
        int rc;
        if (!piped) {
            if ((rc = run_guarded(h, terms0, s_main, pass))) return rc;
            continue;
        }
        unsigned* flag = h->range_flag.as<unsigned>() + k % RING;
        if (k >= RING) HIP_TRY(hipStreamWaitEvent(s_main, h->ev_join[k % RING], 0));   // chunk k - RING has let go of head_out and of the flag word
        if ((rc = pass(terms0, RangeGate{guarded ? flag : nullptr, 0}))) return rc;
        if (guarded) {
            if ((rc = pass(RANGE_FREE_MODE, RangeGate{flag, 1}))) return rc;
            HIP_TRY(hipMemsetAsync(flag, 0, sizeof(unsigned), h->side));
        }
        HIP_TRY(hipEventRecord(h->ev_join[k % RING], h->side));
    }
    if (piped) HIP_TRY(hipStreamWaitEvent(s_main, h->ev_join[(k - 1) % RING], 0));
    return KWS_OK;
}


// own_feat: the features come from this library's front end (log-mel values, far inside fp16's range)
int run_model(kws_handle* h, const float* feat, int B, int T, float* logits, char* ws_act, hipStream_t s, bool own_feat) {
    int rc;
    if ((rc = prof_mark(h, h->ev_model, s))) return rc;
    if (h->plan == PLAN_RESNET) {
        if (use_fused(h, T)) {
            static const int dbg = experiment_int("KWS_R8_DEBUG", 0);
            if (h->res8_impl == 0) {
                h->last_plan = "res8_fused";
                Res8hParams p{};
                p.feat = feat; p.logits = logits; p.w0h = h->r8h_w0.p; p.inv_scale0 = 1.0f / h->r8h_scale0; p.apk2 = h->r8h_apk.p;
                p.bn_tab = h->r8h_bn.as<float>(); p.out_w = h->out_w.as<float>(); p.out_b = h->out_b.as<float>();
                for (int i = 0; i < R8_LAYERS; ++i) {
                    p.inv_scale[i] = 1.0f / h->r8h_scale[i];
                    p.kappa[i] = h->r8h_kappa[i];
                }
                p.B = B; p.T = T; p.F = h->d.freq; p.n_labels = h->d.n_labels; p.debug = dbg;
                p.terms = h->d.dtype == KWS_DTYPE_F16 ? 1 : 3;
                p.queue = h->range_flag.as<unsigned>() + 16;   // (word 0 of that block is the layer-wise range flag)
                if (!own_feat) {   // caller-provided features: any finite fp32 value (reference model/resnet.py:39-41)
                    // B words of the handle's own (ABI version 2 promised that kws_forward needs no workspace on this plan): grown, with a
                    // blocking allocation, the first time a larger batch arrives (include/kws.h: the third warm-up exception); the outgrown
                    // block stays alive for graphs captured at the smaller size
                    if ((rc = h->r8_shift.reserve_parked(align256((size_t)B * sizeof(int)), h->parked, s, "kws_forward (fused res8 plan)"))) return rc;
                    int* fsh = h->r8_shift.as<int>();
                    HIP_TRY(launch_feat_shift(feat, B, T * h->d.freq, fsh, s));
                    p.feat_shift = fsh;
                }
                static const int r8_wgs = experiment_int("KWS_R8_WGS_PER_CU", 2);
                HIP_TRY(launch_res8h(p, std::min(B, r8_wgs * h->n_cu), s));
            } else if (h->res8_impl == 2) {
                h->last_plan = "res8_fused_fp32mfma";
                Res8Params p{feat, logits, h->r8_w0a.as<float>(), h->r8_apk.as<f32x4>(), h->r8_bn.as<float>(),
                             h->r8_zcells.as<int>(), h->out_w.as<float>(), h->out_b.as<float>(), B, T, h->d.freq,
                             h->d.n_labels, dbg};
                static const int grid_env = experiment_int("KWS_R8_GRID", 512);
                HIP_TRY(launch_res8(p, std::min(B, grid_env), s));
            } else {
                h->last_plan = "res8_fused_bf16x6";
                Res8xParams p{feat, logits, h->r8_w0a.as<float>(), h->r8x_apk.p, h->r8_bn.as<float>(),
                              h->out_w.as<float>(), h->out_b.as<float>(), B, T, h->d.freq, h->d.n_labels, dbg,
                              h->d.dtype == KWS_DTYPE_BF16X3 ? 3 : h->d.dtype == KWS_DTYPE_BF16 ? 1 : 6};
                HIP_TRY(launch_res8x(p, std::min(B, 256), s));
            }
        } else {
            h->last_plan = resnet_tiled(h, resnet_shape(h, T)) ? "resnet_tiled" : "layerwise";
            if ((rc = run_resnet_layerwise(h, feat, B, T, logits, ws_act, s))) return rc;
        }
    } else if (h->plan == PLAN_CNN) {
        h->last_plan = cnn_band_plan(h, dtype_terms(h->d.dtype)) ? "cnn_band" : cnn_in1_plan(h, dtype_terms(h->d.dtype)) ? "cnn_in1" : "layerwise";
        if ((rc = run_cnn(h, feat, B, T, logits, ws_act, s))) return rc;
    } else {
        return fail(KWS_EUNSUPPORTED, "handle was created with family KWS_MODEL_NONE (front end only)");
    }
    return prof_mark(h, h->ev_model, s);
}

int check_ws(const kws_handle* h, size_t need) {
    if (need == 0) return KWS_OK;
    if (!h->ws || h->ws_bytes < need)
        return fail(KWS_ENOWORKSPACE, "workspace too small: need " + std::to_string(need) + " bytes, have " +
                                          std::to_string(h->ws_bytes));
    return KWS_OK;
}

int strip_and_match(const char* name, std::string& out) {
    if (!name) return fail(KWS_EINVAL, "null tensor name");
    out = name;
    if (out.rfind("module.", 0) == 0) out = out.substr(7);   // DataParallel prefix (reference run/test.py:69-70)
    return KWS_OK;
}

}  // namespace

// Nothing may throw across the C ABI (include/kws.h): every entry point runs inside this guard.  std::bad_alloc (std::vector /
// std::string growth in kws_create, kws_load_weights and finalize) becomes KWS_ENOMEM, anything else KWS_EINVAL with the
// exception's message; size queries return 0.
template <typename R, typename F>
static R guarded(F&& body) noexcept {
    int code = KWS_EINVAL;
    const char* what = "unexpected C++ exception";
    char msg[256];
    try {
        return body();
    } catch (const std::bad_alloc&) {
        code = KWS_ENOMEM;
        what = "out of host memory";
    } catch (const std::exception& e) {
        std::snprintf(msg, sizeof(msg), "%s", e.what());
        what = msg;
    } catch (...) {
    }
    try {
        g_err = what;
    } catch (...) {
        g_err.clear();
    }
    return std::is_same<R, size_t>::value ? (R)0 : (R)code;
}

// ================================================================================================== C ABI
extern "C" {

int kws_abi_version(void) { return KWS_ABI_VERSION; }
const char* kws_last_error(void) { return g_err.c_str(); }

int kws_create(const kws_model_desc* desc, kws_handle** out) {
    return guarded<int>([&]() -> int {
    if (!desc || !out) return fail(KWS_EINVAL, "null argument");
#ifdef KWS_EXPERIMENTS
    if (const char* t = std::getenv("KWS_TEST_THROW")) {   // fault injection (experiments build only; tests/test_host.py loads that build for it): what the guard makes of an exception
        if (std::strcmp(t, "bad_alloc") == 0) throw std::bad_alloc();
        throw std::runtime_error(t);
    }
#endif
    if (desc->struct_size != (int)sizeof(kws_model_desc)) return fail(KWS_EINVAL, "kws_model_desc size mismatch (ABI)");
    if (desc->dtype != KWS_DTYPE_F32 && desc->dtype != KWS_DTYPE_BF16X3 && desc->dtype != KWS_DTYPE_BF16 &&
        desc->dtype != KWS_DTYPE_F16)
        return fail(KWS_EUNSUPPORTED, "dtype must be one of KWS_DTYPE_F32 / BF16X3 / BF16 / F16");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) return fail(KWS_EHIP, "no HIP device available: the HIP path is mandatory, there is no CPU fallback");
    std::unique_ptr<kws_handle> h(new (std::nothrow) kws_handle());
    if (!h) return fail(KWS_ENOMEM, "out of host memory");
    h->d = *desc;
    HIP_TRY(hipGetDevice(&h->device));
    HIP_TRY(hipDeviceGetAttribute(&h->n_cu, hipDeviceAttributeMultiprocessorCount, h->device));
    if (const int nc = experiment_int("KWS_N_CU", 0))   // experiments with CU-masked streams: size the persistent grids for fewer CUs
        if (nc > 0 && nc < h->n_cu) h->n_cu = nc;
    const char* fl = std::getenv("KWS_FORCE_LAYERWISE");
    h->force_layerwise = fl && fl[0] == '1';
    if (const char* lw = std::getenv("KWS_LAYERWISE_IMPL")) {
        if (std::strcmp(lw, "fp32") == 0) h->lw_mode = LW_FP32;
        else if (std::strcmp(lw, "nchw") == 0) h->lw_mode = LW_NCHW;
    }
    if (const char* tp = std::getenv("KWS_T3_TRIPLE")) h->t3_triple = std::atoi(tp);
    if (const char* tp = std::getenv("KWS_CNN_LIN_F16")) h->lin_in_f16 = std::atoi(tp) != 0;
    if (const char* tp = std::getenv("KWS_T3_PAIR")) h->t3_pair = std::atoi(tp) != 0;   // A/B and tests: 0 = one kernel per layer
    if (const char* tp = std::getenv("KWS_T3_STREAM")) h->t3_stream = std::atoi(tp);   // A/B and tests: 0 = tile / pair / triple kernels only, 2 = streams whatever the launch size, 3 = stream odd-first runs only
    int rc = setup_frontend(h.get());
    if (rc) return rc;
    const unsigned zero_word[64] = {0};
    if ((rc = h->range_flag.upload(zero_word, sizeof(zero_word)))) return rc;
    switch (desc->family) {
        case KWS_MODEL_NONE: h->plan = PLAN_FRONTEND_ONLY; break;
        case KWS_MODEL_RESNET: h->plan = PLAN_RESNET; rc = build_resnet(h.get()); break;
        case KWS_MODEL_CNN: h->plan = PLAN_CNN; rc = build_cnn(h.get()); break;
        default: return fail(KWS_EINVAL, "unknown model family");
    }
    if (rc) return rc;
    if (h->plan == PLAN_CNN) {
        if (const char* tp = std::getenv("KWS_CNN_STREAMS")) h->cnn_streams = std::atoi(tp) != 0;   // A/B and tests: 0 = one stream
        if (h->cnn_streams) {
            // (the side stream's kernels are small and find the chip full of the next chunk's convolution workgroups: highest priority, so that they are
            // dispatched as slots come free instead of behind the convolution's queue)
            int prio_least = 0, prio_greatest = 0;
            HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
            HIP_TRY(hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, prio_greatest));
            // the events order two streams of ONE device: no system-scope fence (an L2 write-back per record, ~8 us between conv_band and the next conv_in1)
            for (int i = 0; i < kws_handle::CNN_RING; ++i) {
                HIP_TRY(hipEventCreateWithFlags(&h->ev_fork[i], hipEventDisableTiming | hipEventDisableSystemFence));
                HIP_TRY(hipEventCreateWithFlags(&h->ev_join[i], hipEventDisableTiming | hipEventDisableSystemFence));
            }
        }
    }
    *out = h.release();
    return KWS_OK;
    });
}

void kws_destroy(kws_handle* h) {
    try {
        delete h;
    } catch (...) {
    }
}

int kws_load_weights(kws_handle* h, const char* name_in, const void* host_ptr, size_t bytes) {
    return guarded<int>([&]() -> int {
    DeviceGuard dg(h);
    if (!h || !host_ptr) return fail(KWS_EINVAL, "null argument");
    std::string name;
    int rc = strip_and_match(name_in, name);
    if (rc) return rc;
    if (name.size() > 19 && name.compare(name.size() - 19, 19, "num_batches_tracked") == 0) return KWS_OK;
    if (!h->required.count(name)) return fail(KWS_EINVAL, "unexpected tensor name: " + name);
    const float* src = static_cast<const float*>(host_ptr);
    auto need = [&](size_t n) -> int {
        if (bytes != n * sizeof(float))
            return fail(KWS_EINVAL, name + ": expected " + std::to_string(n * 4) + " bytes, got " + std::to_string(bytes));
        return KWS_OK;
    };
    char kind[32];
    int idx = -1;
    char field[32];
    if (h->plan == PLAN_RESNET) {
        const int C = h->d.n_feature_maps;
        if (name == "layers.output.weight") {
            if ((rc = need((size_t)h->d.n_labels * C))) return rc;
            if ((rc = h->out_w.upload(src, bytes))) return rc;
        } else if (name == "layers.output.bias") {
            if ((rc = need(h->d.n_labels))) return rc;
            if ((rc = h->out_b.upload(src, bytes))) return rc;
        } else if (std::sscanf(name.c_str(), "layers.conv_%d.%31s", &idx, field) == 2) {
            const size_t n = idx == 0 ? (size_t)C * 9 : (size_t)C * C * 9;
            if ((rc = need(n))) return rc;
            h->rconv[idx].w_host.assign(src, src + n);
            if (h->res8_eligible) {
                if (idx == 0) {
                    std::vector<float> frag(3 * 3 * 64);
                    pack_res8_conv0(src, frag.data());
                    if ((rc = h->r8_w0a.upload(frag.data(), frag.size() * 4))) return rc;
                    std::vector<unsigned short> w0h((size_t)(3 * 2 + 3) * 64 * 8);
                    h->r8h_scale0 = weight_scale_pow2(src, n);
                    pack_res8h_conv0(src, h->r8h_scale0, w0h.data());
                    if ((rc = h->r8h_w0.upload(w0h.data(), w0h.size() * 2))) return rc;
                } else {
                    pack_res8_layer(src, h->r8_apk_host.data() + (size_t)(idx - 1) * R8_GROUPS * 3 * 64 * 4);
                    pack_res8x_layer(src, h->r8x_apk_host.data() + (size_t)(idx - 1) * R8X_KSTEPS * 3 * 3 * 64 * 8);
                    // (res8h_kernel's fragments are packed in finalize(): the previous layer's BatchNorm is folded into them)
                }
            }
        } else if (std::sscanf(name.c_str(), "layers.bn_%d.%31s", &idx, field) == 2) {
            if ((rc = need(C))) return rc;
            std::vector<float>& dst = std::strcmp(field, "running_mean") == 0 ? h->bn[idx - 1].mean : h->bn[idx - 1].var;
            dst.assign(src, src + C);
        }
    } else if (h->plan == PLAN_CNN) {
        if (std::sscanf(name.c_str(), "layers.conv_%d.%31s", &idx, field) == 2) {
            ConvLayer& L = h->cconv[idx];
            if (std::strcmp(field, "weight") == 0) {
                if ((rc = need((size_t)L.g.Cout * L.g.Cin * L.g.kh * L.g.kw))) return rc;
                if ((rc = upload_packed(L, src, h->lw_mode))) return rc;
                if (idx == 0 && h->cnn_in1) {   // the same weights in conv_in1.hip's fragment order, for both of its tile groupings
                    const int mhs[2] = {IN1_MH3, IN1_MH1};
                    for (int v = 0; v < 2; ++v) {
                        std::vector<unsigned short> pki;
                        pack_conv_in1_weights(L.g.Cout, L.g.kh, mhs[v], src, L.x_scale, pki);
                        if ((rc = L.apk_in1[v].upload(pki.data(), pki.size() * sizeof(unsigned short)))) return rc;
                    }
                }
                if (idx == 1 && h->cnn_band_R > 0) {   // the same weights in conv_band.hip's fragment order
                    std::vector<unsigned short> pkb;
                    L.band_scale = weight_scale_pow2(src, (size_t)L.g.Cout * L.g.Cin * L.g.kh * L.g.kw);
                    pack_conv_band_weights(L.g.Cin, L.g.Cout, L.g.kh, L.g.kw, src, L.band_scale, pkb);
                    if ((rc = L.apk_band.upload(pkb.data(), pkb.size() * sizeof(unsigned short)))) return rc;
                    if (h->cnn_cols) {
                        pack_conv_cols_weights(L.g.Cin, L.g.Cout, L.g.kh, src, L.band_scale, pkb);
                        if ((rc = L.apk_cols.upload(pkb.data(), pkb.size() * sizeof(unsigned short)))) return rc;
                    }
                }
            } else {
                if ((rc = need(L.g.Cout))) return rc;
                if ((rc = L.bias.upload(src, bytes))) return rc;
            }
        } else if (std::sscanf(name.c_str(), "layers.%31[^.].%31s", kind, field) == 2) {
            size_t li = 0;
            for (; li < h->clin_names.size(); ++li)
                if (h->clin_names[li] == kind) break;
            if (li == h->clin_names.size()) return fail(KWS_EINVAL, "unexpected tensor name: " + name);
            ConvLayer& L = h->clin[li];
            if (std::strcmp(field, "weight") == 0) {
                if ((rc = need((size_t)L.g.Cout * L.g.kw))) return rc;
                if ((rc = upload_packed(L, src, h->lw_mode))) return rc;
                if (li == 0 && (h->cnn_band_R > 0 || h->cnn_cl1)) {
                    // the channels-last plans hand this Linear the last conv's output as (position, channel padded to 16) instead
                    // of the reference's flatten order (channel, position): same weights, columns permuted, zeros for the padding
                    ConvLayer& Lc = h->clin0_cl;
                    const int C1 = h->cl_last[0], npos = h->cl_last[1], cp1 = h->cl_last[2];
                    std::vector<float> wcl((size_t)L.g.Cout * Lc.g.kw, 0.f);
                    for (int o = 0; o < L.g.Cout; ++o)
                        for (int c = 0; c < C1; ++c)
                            for (int ps = 0; ps < npos; ++ps)
                                wcl[(size_t)o * Lc.g.kw + (size_t)ps * cp1 + c] = src[(size_t)o * L.g.kw + (size_t)c * npos + ps];
                    if ((rc = upload_packed(Lc, wcl.data(), h->lw_mode))) return rc;
                }
            } else {
                if ((rc = need(L.g.Cout))) return rc;
                if ((rc = L.bias.upload(src, bytes))) return rc;
            }
        }
    }
    h->loaded.insert(name);
    h->dirty = true;
    return KWS_OK;
    });
}

int kws_num_frames(const kws_handle* h, int n_samples) {
    return guarded<int>([&]() -> int {
    if (!h || n_samples < 0) return fail(KWS_EINVAL, "bad argument");
    return 1 + n_samples / h->d.hop_length;
    });
}

int kws_chunk_clips(const kws_handle* h, int B, int T) {
    return guarded<int>([&]() -> int {
    if (!h || B < 1 || T < 1) return 0;
    if (h->plan == PLAN_CNN) return cnn_chunk(h, B);
    if (h->plan == PLAN_RESNET) {
        if (use_fused(h, T)) return B;
        const ResnetShape s = resnet_shape(h, T);
        if (resnet_tiled(h, s)) return tiled_chunk(h, s, B);
        return chunk_clips((size_t)s.C * s.T * s.F, B);
    }
    return B;
    });
}

size_t kws_workspace_bytes(const kws_handle* h, int B, int T) {
    return guarded<size_t>([&]() -> size_t {
    if (!h || B < 0 || T < 1) return 0;
    return feat_bytes(h, B, T) + act_bytes(h, B, T);
    });
}

int kws_set_workspace(kws_handle* h, void* d_ptr, size_t bytes) {
    return guarded<int>([&]() -> int {
    DeviceGuard dg(h);
    if (!h) return fail(KWS_EINVAL, "null handle");
    if (d_ptr && (reinterpret_cast<uintptr_t>(d_ptr) & 255)) return fail(KWS_EINVAL, "workspace must be 256-byte aligned");
    h->ws = d_ptr;
    h->ws_bytes = d_ptr ? bytes : 0;
    // The generic layer-wise kernels read (and multiply by zero weights) up to 7 planes past the end of a tensor whose
    // channel count is not a multiple of 8: make sure what they find there is finite from the first call on.
    if (d_ptr && bytes) {
        HIP_TRY(hipMemsetAsync(d_ptr, 0, bytes, nullptr));
        HIP_TRY(hipStreamSynchronize(nullptr));
    }
    return KWS_OK;
    });
}

static int mfcc_any(kws_handle* h, const float* d_wav, const int16_t* d_pcm, const float* d_noise, float noise_pct,
                    int B, int n_samples, float* d_feat, void* stream, long long clip_stride = -1) {
    DeviceGuard dg(h);
    if (!h || (!d_wav && !d_pcm) || !d_feat || B < 0) return fail(KWS_EINVAL, "bad argument");
    if (n_samples <= FE_NFFT / 2) return fail(KWS_EINVAL, "clip shorter than the reflect padding (n_fft/2 + 1 samples needed)");
    if (h->d.n_mels != h->d.freq && h->plan != PLAN_FRONTEND_ONLY) return fail(KWS_EINVAL, "n_mels != model frequency bins");
    const int T = 1 + n_samples / FE_HOP;
    hipStream_t s = static_cast<hipStream_t>(stream);
    int rc;
    if ((rc = prof_mark(h, h->ev_front, s))) return rc;
    FrontendParams p{d_wav, reinterpret_cast<const short*>(d_pcm), d_noise, noise_pct, d_feat, h->dft.as<f32x4>(),
                     h->hann.as<float>(), h->melw.as<float>(), h->mel_lo.as<int>(), h->mel_hi.as<int>(), B, n_samples, T,
                     h->d.n_mels, (T + FE_FRAMES - 1) / FE_FRAMES, h->mel_maxw, clip_stride < 0 ? n_samples : clip_stride,
                     h->dft16.as<void>(), h->consts16.as<float>(), h->mel_a.as<float>(),
                     {h->mel_fb[0], h->mel_fb[1], h->mel_fb[2]}, {h->mel_ns[0], h->mel_ns[1], h->mel_ns[2]},
                     h->range_flag.as<unsigned>() + 32};
    // the fp16 kernel reads whole 16-byte groups: rows (and base pointers) that are not 16-byte aligned -- odd clip
    // lengths, sliced buffers -- take the fp32-input kernel, which stages element-wise
    const bool aligned = (p.clip_stride & 3) == 0 && ((reinterpret_cast<uintptr_t>(d_wav) | reinterpret_cast<uintptr_t>(d_noise)) & 15) == 0 &&
                         (reinterpret_cast<uintptr_t>(d_pcm) & 7) == 0;
    if (h->fe_fp32 || !aligned) {
        HIP_TRY(launch_frontend(p, s));
    } else {
        p.chunks = (T + frontend_f16_frames() - 1) / frontend_f16_frames();     // its units are shorter
        HIP_TRY(launch_frontend_f16(p, h->n_cu, s));
    }
    return prof_mark(h, h->ev_front, s);
}

int kws_mfcc(kws_handle* h, const float* d_wav, int B, int n_samples, float* d_feat, void* stream) {
    return guarded<int>([&]() -> int {
    return mfcc_any(h, d_wav, nullptr, nullptr, 0.f, B, n_samples, d_feat, stream);
    });
}

int kws_mfcc_pcm16(kws_handle* h, const int16_t* d_pcm, const float* d_noise, float noise_pct, int B, int n_samples,
                   float* d_feat, void* stream) {
    return guarded<int>([&]() -> int {
    return mfcc_any(h, nullptr, d_pcm, d_noise, noise_pct, B, n_samples, d_feat, stream);
    });
}

int kws_forward(kws_handle* h, const void* d_feat, int B, int T, void* d_logits, void* stream) {
    return guarded<int>([&]() -> int {
    DeviceGuard dg(h);
    if (!h || !d_feat || !d_logits || B < 0 || T < 1) return fail(KWS_EINVAL, "bad argument");
    int rc = finalize(h);
    if (rc) return rc;
    const size_t need = act_bytes(h, B, T);
    if ((rc = check_ws(h, need))) return rc;
    // activations use the TAIL of the workspace so that kws_forward_wav's feature block (the head) stays intact
    char* ws_act = need ? static_cast<char*>(h->ws) + (h->ws_bytes - need) : nullptr;
    if (ws_act) ws_act = reinterpret_cast<char*>(reinterpret_cast<uintptr_t>(ws_act) & ~(uintptr_t)255);
    return run_model(h, static_cast<const float*>(d_feat), B, T, static_cast<float*>(d_logits), ws_act,
                     static_cast<hipStream_t>(stream), false);
    });
}

static int forward_any(kws_handle* h, const float* d_wav, const int16_t* d_pcm, const float* d_noise, float noise_pct,
                       int B, int n_samples, float* d_logits, void* stream, long long clip_stride = -1) {
    DeviceGuard dg(h);
    if (!h || (!d_wav && !d_pcm) || !d_logits || B < 0) return fail(KWS_EINVAL, "bad argument");
    int rc = finalize(h);
    if (rc) return rc;
    const int T = 1 + n_samples / FE_HOP;
    const size_t fb = feat_bytes(h, B, T), ab = act_bytes(h, B, T);
    if ((rc = check_ws(h, fb + ab))) return rc;
    float* feat = static_cast<float*>(h->ws);
    if ((rc = mfcc_any(h, d_wav, d_pcm, d_noise, noise_pct, B, n_samples, feat, stream, clip_stride))) return rc;
    char* ws_act = ab ? static_cast<char*>(h->ws) + fb : nullptr;
    return run_model(h, feat, B, T, d_logits, ws_act, static_cast<hipStream_t>(stream), true);
}

int kws_forward_wav(kws_handle* h, const float* d_wav, int B, int n_samples, float* d_logits, void* stream) {
    return guarded<int>([&]() -> int {
    return forward_any(h, d_wav, nullptr, nullptr, 0.f, B, n_samples, d_logits, stream);
    });
}

int kws_forward_pcm16(kws_handle* h, const int16_t* d_pcm, const float* d_noise, float noise_pct, int B, int n_samples,
                      float* d_logits, void* stream) {
    return guarded<int>([&]() -> int {
    return forward_any(h, nullptr, d_pcm, d_noise, noise_pct, B, n_samples, d_logits, stream);
    });
}

static int check_windows(size_t n_stream, int window, int shift, int n_windows) {
    if (window < 1 || shift < 1 || n_windows < 0) return fail(KWS_EINVAL, "bad window description");
    if (n_windows > 0 && (size_t)(n_windows - 1) * (size_t)shift + (size_t)window > n_stream)
        return fail(KWS_EINVAL, "windows run past the end of the stream");
    return KWS_OK;
}

// Windows whose shift and length are multiples of the hop share all but their four edge frames with the stream's own
// frames; worth it when the shared rows outnumber the stream's rows (i.e. the windows overlap).
static bool windows_share_frames(int window, int shift, int n_windows) {
    if (shift % FE_HOP || window % FE_HOP || window <= FE_NFFT) return false;
    const long long T = 1 + window / FE_HOP, rows_g = (long long)(n_windows - 1) * (shift / FE_HOP) + T;
    return n_windows >= 4 && T >= 8 && 2 * rows_g <= (long long)n_windows * (T - 4);
}
static size_t global_feat_bytes(const kws_handle* h, int window, int shift, int n_windows) {
    const size_t rows = (size_t)std::max(n_windows - 1, 0) * (size_t)(shift / FE_HOP) + 1 + window / FE_HOP;
    return align256(rows * h->d.n_mels * sizeof(float));
}

// features of all windows; gbuf: device scratch of global_feat_bytes() for the shared-frame path, or nullptr
static int mfcc_windows_impl(kws_handle* h, const float* d_stream, int window, int shift, int n_windows, float* d_feat,
                             void* stream, float* gbuf) {
    DeviceGuard dg(h);
    const bool no_share = std::getenv("KWS_WINDOWS_NO_SHARE") != nullptr;   // A/B and tests (read per call on purpose)
    if (!gbuf || no_share || !windows_share_frames(window, shift, n_windows))
        return mfcc_any(h, d_stream, nullptr, nullptr, 0.f, n_windows, window, d_feat, stream, shift);
    if (!d_stream || !d_feat) return fail(KWS_EINVAL, "bad argument");
    const long long covered = (long long)(n_windows - 1) * shift + window;
    if (covered > 0x7fffffffLL) return fail(KWS_EINVAL, "stream too long for one call");
    int rc = mfcc_any(h, d_stream, nullptr, nullptr, 0.f, 1, (int)covered, gbuf, stream);   // the stream as one clip
    if (rc) return rc;
    WindowEdgeParams p{d_stream, gbuf, d_feat, h->edge_hann.as<float>(), h->edge_trig.as<f32x2>(), h->melw.as<float>(),
                       h->mel_lo.as<int>(), h->mel_hi.as<int>(), window, shift, n_windows, 1 + window / FE_HOP, h->d.n_mels};
    HIP_TRY(launch_window_edges(p, static_cast<hipStream_t>(stream)));
    return KWS_OK;
}

size_t kws_workspace_bytes_windows(const kws_handle* h, int window, int shift, int n_windows) {
    return guarded<size_t>([&]() -> size_t {
    if (!h || window < 1 || shift < 1 || n_windows < 0) return 0;
    const size_t base = kws_workspace_bytes(h, n_windows, 1 + window / FE_HOP);
    return base + (windows_share_frames(window, shift, n_windows) ? global_feat_bytes(h, window, shift, n_windows) : 0);
    });
}

int kws_mfcc_windows(kws_handle* h, const float* d_stream, size_t n_stream, int window, int shift, int n_windows,
                     float* d_feat, void* stream) {
    return guarded<int>([&]() -> int {
    DeviceGuard dg(h);
    if (!h) return fail(KWS_EINVAL, "null handle");
    int rc = check_windows(n_stream, window, shift, n_windows);
    if (rc) return rc;
    // the scratch for the stream's own frames comes from the head of the workspace when there is one that is big enough
    float* gbuf = (h->ws && h->ws_bytes >= global_feat_bytes(h, window, shift, n_windows)) ? static_cast<float*>(h->ws) : nullptr;
    return mfcc_windows_impl(h, d_stream, window, shift, n_windows, d_feat, stream, gbuf);
    });
}

int kws_forward_windows(kws_handle* h, const float* d_stream, size_t n_stream, int window, int shift, int n_windows,
                        float* d_logits, void* stream) {
    return guarded<int>([&]() -> int {
    DeviceGuard dg(h);
    if (!h || !d_stream || !d_logits) return fail(KWS_EINVAL, "bad argument");
    int rc = check_windows(n_stream, window, shift, n_windows);
    if (rc) return rc;
    if ((rc = finalize(h))) return rc;
    const int T = 1 + window / FE_HOP;
    const size_t fb = feat_bytes(h, n_windows, T), ab = act_bytes(h, n_windows, T);
    if ((rc = check_ws(h, fb + ab))) return rc;
    const size_t gb = global_feat_bytes(h, window, shift, n_windows);
    float* feat = static_cast<float*>(h->ws);
    float* gbuf = h->ws_bytes >= fb + ab + gb ? reinterpret_cast<float*>(static_cast<char*>(h->ws) + fb + ab) : nullptr;
    if ((rc = mfcc_windows_impl(h, d_stream, window, shift, n_windows, feat, stream, gbuf))) return rc;
    char* ws_act = ab ? static_cast<char*>(h->ws) + fb : nullptr;
    return run_model(h, feat, n_windows, T, d_logits, ws_act, static_cast<hipStream_t>(stream), true);
    });
}

int kws_eval_batch(kws_handle* h, const float* d_logits, const int64_t* d_target, int B, int64_t* d_stats,
                   double* d_loss_sum, void* stream) {
    return guarded<int>([&]() -> int {
    DeviceGuard dg(h);
    if (!h || !d_logits || !d_target || !d_stats || !d_loss_sum || B < 0) return fail(KWS_EINVAL, "bad argument");
    HIP_TRY(launch_eval_tail(d_logits, d_target, B, h->d.n_labels, d_stats, d_loss_sum, static_cast<hipStream_t>(stream)));
    return KWS_OK;
    });
}

const char* kws_plan_name(const kws_handle* h) { return h ? h->last_plan : "none"; }
const char* kws_plan_detail(const kws_handle* h) {
    if (!h) return "none";
    const bool detailed = h->last_plan && (std::strcmp(h->last_plan, "resnet_tiled") == 0 || std::strcmp(h->last_plan, "cnn_band") == 0 || std::strcmp(h->last_plan, "cnn_in1") == 0);
    return (detailed && !h->plan_detail.empty()) ? h->plan_detail.c_str() : h->last_plan;
}

int kws_profile_enable(kws_handle* h, int enable) {
    return guarded<int>([&]() -> int {
    DeviceGuard dg(h);
    if (!h) return fail(KWS_EINVAL, "null handle");
    h->prof = enable != 0;
    while (h->prof && h->ev_pool.size() < 4 * 128) {   // four events per wav -> logits call: 128 calls between two reads
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        h->ev_pool.push_back(e);
    }
    return KWS_OK;
    });
}

int kws_profile_read(kws_handle* h, double* model_ms, double* frontend_ms, int* calls) {
    return guarded<int>([&]() -> int {
    DeviceGuard dg(h);
    if (!h) return fail(KWS_EINVAL, "null handle");
    auto drain = [&](std::vector<hipEvent_t>& v, double& acc) -> int {
        for (size_t i = 0; i + 1 < v.size(); i += 2) {
            HIP_TRY(hipEventSynchronize(v[i + 1]));
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, v[i], v[i + 1]));
            acc += ms;
        }
        v.clear();
        return KWS_OK;
    };
    const int ncalls = (int)(h->ev_model.size() / 2);
    int rc;
    if ((rc = drain(h->ev_model, h->acc_model_ms))) return rc;
    if ((rc = drain(h->ev_front, h->acc_front_ms))) return rc;
    if (model_ms) *model_ms = h->acc_model_ms;
    if (frontend_ms) *frontend_ms = h->acc_front_ms;
    if (calls) *calls = ncalls;
    h->ev_next = 0;
    h->acc_model_ms = h->acc_front_ms = 0;
    return KWS_OK;
    });
}

}  // extern "C"
