/* A host in plain C on the C ABI (include/css_mi355.h): no Python, no torch.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/c_host.c -o c_host -Lnotsofar1-challenge_amd -lcss_mi355 \
 *       -Wl,-rpath,$PWD/notsofar1-challenge_amd -Wl,-rpath,/opt/rocm/lib -Wl,-rpath-link,/opt/rocm/lib -lm
 *   python examples/export_for_c_host.py /tmp/css_c        # model.bin (CssModelDesc + weight blob), pcm.f32 ([n][7] float32)
 *   ./c_host /tmp/css_c/model.bin /tmp/css_c/pcm.f32 7 /tmp/css_c/wav.f32
 *
 * What separate_and_stitch does for one recording (css/css.py:110-338), then the same recording eight times through the queue
 * (css_run_enqueue / css_wait: sessions share mask-estimator batches, PCIe legs hide under the neighbours' kernels).
 * tests/test_hip_session.py runs this program on the GPU box and compares wav.f32 with the Python shim's result, bit for bit. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "css_mi355.h"

static double now_ms(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return 1e3 * (double)ts.tv_sec + 1e-6 * (double)ts.tv_nsec;
}

#define CHECK(call)                                                                                   \
    do {                                                                                              \
        int rc_ = (call);                                                                             \
        if (rc_ != CSS_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, css_last_error(h)); return 1; } \
    } while (0)

int main(int argc, char** argv) {
    css_handle_t h = NULL;
    if (argc < 5) { fprintf(stderr, "usage: %s model.bin pcm.f32 n_channels wav_out.f32\n", argv[0]); return 2; }
    /* ---- the model: CssModelDesc, int64 count, float32 blob (weights.py::pack_blob's layout, css_mi355.h) */
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    CssModelDesc desc;
    int64_t n_floats = 0;
    if (fread(&desc, sizeof desc, 1, f) != 1 || fread(&n_floats, sizeof n_floats, 1, f) != 1 || n_floats != css_blob_num_floats(&desc)) {
        fprintf(stderr, "%s: not a model file for this library\n", argv[1]);
        return 2;
    }
    float* blob = (float*)malloc((size_t)n_floats * sizeof(float));
    if (!blob || fread(blob, sizeof(float), (size_t)n_floats, f) != (size_t)n_floats) { fprintf(stderr, "short model file\n"); return 2; }
    fclose(f);
    /* ---- the recording: [n][C] float32, exactly load_audio's layout (css/helpers.py:40-65); read into page-locked memory below */
    const int n_ch = atoi(argv[3]);
    f = fopen(argv[2], "rb");
    if (!f) { perror(argv[2]); return 2; }
    fseek(f, 0, SEEK_END);
    const int64_t n = (int64_t)(ftell(f) / (long)(sizeof(float) * (size_t)n_ch));
    fseek(f, 0, SEEK_SET);
    /* ---- CssCfg in seconds -> frames and the three stitching windows (css.py:144-152, 341-390) */
    CssCfgSeconds sec = {3.0, 1.5, 0.15, 0.3, 0.4, 0.2, /* activity_th, inference_v1.yaml:17 */ 0.3,
                         /* mc_mask_floor_db */ 0.0, /* mc_mvdr */ 1, /* l1 */ 0, /* mask */ 0, 0};
    if (n_ch == 1) sec.mask_floor_db = -1.0 / 0.0;   /* sc_mask_floor_db = -inf (css.py:223) */
    static float windows[3 * 1024];
    CssRunCfg cfg;
    CHECK(css_make_run_cfg(&desc, &sec, 16000, &cfg, windows, 3 * 1024));
    CssPlan plan;
    CHECK(css_plan(&desc, &cfg, n, &plan));
    printf("%s: %lld samples x %d channels = %lld segments of %d frames, %lld output samples per stream\n", css_version(), (long long)n, n_ch,
           (long long)plan.num_segments, cfg.segment_frames, (long long)plan.n_out);
    if (css_device_count() < 1) { fprintf(stderr, "no HIP device\n"); return 3; }
    float* pcm = NULL;
    CHECK(css_host_alloc((size_t)n * (size_t)n_ch * sizeof(float), (void**)&pcm));
    if (fread(pcm, sizeof(float) * (size_t)n_ch, (size_t)n, f) != (size_t)n) { fprintf(stderr, "short pcm file\n"); return 2; }
    fclose(f);
    float* wav = NULL;
    CHECK(css_host_alloc((size_t)desc.num_spks * (size_t)plan.n_out * sizeof(float), (void**)&wav));
    CHECK(css_create(&desc, blob, n_floats, /* device */ 0, /* stream: the library's own */ NULL, /* max_batch_segments */ 256, &h));
    free(blob);
    /* ---- one synchronous pass (float32 operands: the library's default arithmetic) */
    CHECK(css_run(h, pcm, n, n_ch, &cfg, wav, plan.n_out));   /* the first call also sizes the workspace */
    double t0 = now_ms();
    CHECK(css_run(h, pcm, n, n_ch, &cfg, wav, plan.n_out));
    double dt = now_ms() - t0;
    printf("css_run: %.2f ms = %.0f x real time\n", dt, (double)n / 16.0 / dt);
    f = fopen(argv[4], "wb");
    if (!f || fwrite(wav, sizeof(float), (size_t)desc.num_spks * (size_t)plan.n_out, f) != (size_t)desc.num_spks * (size_t)plan.n_out) { perror(argv[4]); return 2; }
    fclose(f);
    /* ---- the same recording as eight queued sessions with their own output buffers */
    enum { K = 8 };
    float* out[K];
    for (int k = 0; k < K; ++k) CHECK(css_host_alloc((size_t)desc.num_spks * (size_t)plan.n_out * sizeof(float), (void**)&out[k]));
    for (int rep = 0; rep < 2; ++rep) {
        t0 = now_ms();
        for (int k = 0; k < K; ++k) CHECK(css_run_enqueue(h, pcm, n, n_ch, &cfg, out[k], plan.n_out));
        CHECK(css_wait(h));
        dt = now_ms() - t0;
    }
    int same = 1;
    for (int k = 0; k < K; ++k) same = same && memcmp(out[k], wav, (size_t)desc.num_spks * (size_t)plan.n_out * sizeof(float)) == 0;
    printf("css_run_enqueue x %d + css_wait: %.2f ms per session = %.0f x real time; every session equals css_run bit for bit: %s\n", K, dt / K,
           (double)n / 16.0 / (dt / K), same ? "yes" : "NO");
    for (int k = 0; k < K; ++k) css_host_free(out[k]);
    css_host_free(wav);
    css_host_free(pcm);
    CHECK(css_destroy(h));
    return same ? 0 : 4;
}
