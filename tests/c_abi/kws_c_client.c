/*
 * A plain-C consumer of the C ABI (include/kws.h): no Python, no PyTorch -- only libkws_hip.so and the HIP runtime.
 * It is what the cgo / JNI / ctypes stub of INTEGRATION.md section 2 boils down to, and tests/test_gpu_parity.py builds and runs it
 * to show that the library needs nothing from the process it is loaded into (its own device memory, its own stream).
 *
 *   kws_c_client <bundle> <logits.out> [pcm]
 *
 * bundle (little endian): kws_model_desc | int32 n_tensors | n x { int32 name_len, name, int64 bytes, fp32 data } |
 *                         int32 B, int32 n_samples | B x n_samples fp32 waveforms
 * Writes B x n_labels fp32 logits of kws_forward_wav on a stream of its own; with a third argument the clips are first rounded to
 * 16-bit PCM on the host and take kws_forward_pcm16.
 *
 * build: gcc -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tests/c_abi/kws_c_client.c -o kws_c_client \
 *            -Lhonk2_amd -lkws_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/honk2_amd -Wl,-rpath,/opt/rocm/lib
 */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "kws.h"

#define CHECK_HIP(x)                                                                  \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                   \
            return 2;                                                                 \
        }                                                                             \
    } while (0)
#define CHECK_KWS(x)                                                                  \
    do {                                                                              \
        int r_ = (x);                                                                 \
        if (r_ != KWS_OK) {                                                           \
            fprintf(stderr, "%s: %d (%s)\n", #x, r_, kws_last_error());               \
            return 3;                                                                 \
        }                                                                             \
    } while (0)

static int read_exact(FILE* f, void* p, size_t n) { return fread(p, 1, n, f) == n; }

int main(int argc, char** argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: %s <bundle> <logits.out> [pcm]\n", argv[0]);
        return 1;
    }
    const int pcm = argc > 3;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 1;
    kws_model_desc desc;
    int32_t n_tensors = 0;
    if (!read_exact(f, &desc, sizeof desc) || desc.struct_size != (int32_t)sizeof desc || !read_exact(f, &n_tensors, 4)) return 1;
    if (kws_abi_version() != KWS_ABI_VERSION) return 1;

    kws_handle* h = NULL;
    CHECK_KWS(kws_create(&desc, &h));
    for (int i = 0; i < n_tensors; ++i) {
        int32_t len;
        int64_t bytes;
        char name[256];
        if (!read_exact(f, &len, 4) || len <= 0 || len >= (int32_t)sizeof name || !read_exact(f, name, (size_t)len) || !read_exact(f, &bytes, 8)) return 1;
        name[len] = 0;
        void* data = malloc((size_t)bytes);
        if (!data || !read_exact(f, data, (size_t)bytes)) return 1;
        CHECK_KWS(kws_load_weights(h, name, data, (size_t)bytes));   /* copies: the host buffer is ours again */
        free(data);
    }
    int32_t B, n_samples;
    if (!read_exact(f, &B, 4) || !read_exact(f, &n_samples, 4)) return 1;
    const size_t n = (size_t)B * (size_t)n_samples;
    float* wav = (float*)malloc(n * sizeof(float));
    if (!wav || !read_exact(f, wav, n * sizeof(float))) return 1;
    fclose(f);

    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));
    const size_t ws_bytes = kws_workspace_bytes(h, B, kws_num_frames(h, n_samples));
    void *d_ws = NULL, *d_in = NULL;
    float* d_logits = NULL;
    CHECK_HIP(hipMalloc(&d_ws, ws_bytes ? ws_bytes : 16));
    CHECK_KWS(kws_set_workspace(h, d_ws, ws_bytes));
    CHECK_HIP(hipMalloc((void**)&d_logits, (size_t)B * desc.n_labels * sizeof(float)));
    if (pcm) {
        int16_t* q = (int16_t*)malloc(n * sizeof(int16_t));
        if (!q) return 1;
        for (size_t i = 0; i < n; ++i) {
            float v = wav[i] * 32768.0f;
            v = v > 32767.0f ? 32767.0f : (v < -32768.0f ? -32768.0f : v);
            q[i] = (int16_t)(v < 0 ? v - 0.5f : v + 0.5f);
        }
        CHECK_HIP(hipMalloc(&d_in, n * sizeof(int16_t)));
        CHECK_HIP(hipMemcpyAsync(d_in, q, n * sizeof(int16_t), hipMemcpyHostToDevice, stream));
        CHECK_HIP(hipStreamSynchronize(stream));
        free(q);
        CHECK_KWS(kws_forward_pcm16(h, (const int16_t*)d_in, NULL, 0.0f, B, n_samples, d_logits, stream));
    } else {
        CHECK_HIP(hipMalloc(&d_in, n * sizeof(float)));
        CHECK_HIP(hipMemcpyAsync(d_in, wav, n * sizeof(float), hipMemcpyHostToDevice, stream));
        CHECK_KWS(kws_forward_wav(h, (const float*)d_in, B, n_samples, d_logits, stream));
    }
    float* logits = (float*)malloc((size_t)B * desc.n_labels * sizeof(float));
    if (!logits) return 1;
    CHECK_HIP(hipMemcpyAsync(logits, d_logits, (size_t)B * desc.n_labels * sizeof(float), hipMemcpyDeviceToHost, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    FILE* o = fopen(argv[2], "wb");
    if (!o || fwrite(logits, sizeof(float), (size_t)B * desc.n_labels, o) != (size_t)B * desc.n_labels) return 1;
    fclose(o);
    printf("plan %s, %d clips, logits[0][0] = %g\n", kws_plan_name(h), B, logits[0]);
    kws_destroy(h);
    CHECK_HIP(hipFree(d_in));
    CHECK_HIP(hipFree(d_logits));
    CHECK_HIP(hipFree(d_ws));
    CHECK_HIP(hipStreamDestroy(stream));
    free(wav);
    free(logits);
    return 0;
}
