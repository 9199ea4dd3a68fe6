// A host WITHOUT Python or torch: the SOT step of unicorn_sot.py:39-55,78-108 through nothing but include/unicorn_hip.h and the HIP runtime.
// This is what "the drop-in boundary is a C-ABI shared library" means in practice: plain pointers and sizes, a flat weights
// file (uni_weights_file_cfg / uni_ctx_load_file, written by tools/export_weights.py) instead of a pickled checkpoint, caller-owned device buffers.
//     tools/build/capi_host_demo weights.uniw H W out.bin
// Frames are a deterministic hash pattern (tests/test_model_gpu.py::test_c_host_without_torch_matches_the_python_path regenerates them in numpy and
// holds the rows written to out.bin to the Python path).  Built by csrc/build.sh (non-fatally) against unicorn_amd/lib/libunicorn_hip.so.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../include/unicorn_hip.h"

#define HIPCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); return 2; } } while (0)
#define UNI(x) do { int rc_ = (x); if (rc_ < 0) { fprintf(stderr, "%s:%d %s -> %d: %s\n", __FILE__, __LINE__, #x, rc_, uni_last_error()); return 3; } } while (0)

static float* dalloc(size_t n) {
    float* p = nullptr;
    if (hipMalloc(&p, n * sizeof(float)) != hipSuccess) { fprintf(stderr, "hipMalloc of %zu floats failed\n", n); exit(2); }
    return p;
}
// frame t, element i of the (3, H, W) image: a multiplicative hash of the index -> 0..255 (numpy: ((i + 977 t) * 2654435761 mod 2^32) >> 24)
static void make_frame(std::vector<float>& img, int t) {
    for (size_t i = 0; i < img.size(); ++i) img[i] = (float)((uint32_t)((uint32_t)(i + 977u * (uint32_t)t) * 2654435761u) >> 24);
}

int main(int argc, char** argv) {
    if (argc < 5) { fprintf(stderr, "usage: %s weights.uniw H W out.bin\n", argv[0]); return 1; }
    const char* wpath = argv[1];
    const int H = atoi(argv[2]), W = atoi(argv[3]);
    uni_model_cfg cfg;
    UNI(uni_weights_file_cfg(wpath, &cfg));
    uni_ctx* ctx = uni_ctx_create(0, &cfg);
    if (!ctx) { fprintf(stderr, "uni_ctx_create: %s\n", uni_last_error()); return 3; }
    int n_loaded = 0, n_missing = 0;
    UNI(uni_ctx_load_file(ctx, wpath, &n_loaded));
    UNI(uni_ctx_finalize(ctx, &n_missing));
    if (n_missing) { fprintf(stderr, "%d parameters missing, first: %s\n", n_missing, uni_ctx_missing_name(ctx, 0)); return 3; }
    UNI(uni_ctx_reserve(ctx, 1, H, W));
    hipStream_t s;
    HIPCK(hipStreamCreate(&s));
    const int c1 = cfg.dims[1], c2 = cfg.dims[2], c3 = cfg.dims[3];
    const int H8 = H / 8, W8 = W / 8, h = H / 16, w = W / 16, H32 = H / 32, W32 = W / 32;
    const size_t npx = (size_t)3 * H * W, R = (size_t)H8 * W8, A = R + (size_t)h * w + (size_t)H32 * W32;
    const int nch = 6;                                          // SOT head: [cx, cy, w, h, obj, cls]
    std::vector<float> host(npx);
    float *img_ref = dalloc(npx), *img_cur = dalloc(npx);
    float* fpn[2][3];
    for (int k = 0; k < 2; ++k) { fpn[k][0] = dalloc(R * c1); fpn[k][1] = dalloc((size_t)h * w * c2); fpn[k][2] = dalloc((size_t)H32 * W32 * c3); }
    float *f16_ref = dalloc((size_t)h * w * c2), *f16_cur = dalloc((size_t)h * w * c2), *pos = dalloc((size_t)h * w * 256);
    float *o_ref = dalloc((size_t)h * w * 256), *o_cur = dalloc((size_t)h * w * 256), *e_ref = dalloc(R * 128), *e_cur = dalloc(R * 128);
    float *box = dalloc(4), *lbs = dalloc(R), *coarse = dalloc(R), *p16 = dalloc((size_t)h * w), *p32 = dalloc((size_t)H32 * W32), *out = dalloc(A * nch);
    const size_t wsb = uni_corr_workspace_bytes((int)R, (int)R, 1);
    void* ws = nullptr;
    HIPCK(hipMalloc(&ws, wsb ? wsb : 16));
    make_frame(host, 0);
    HIPCK(hipMemcpyAsync(img_ref, host.data(), npx * sizeof(float), hipMemcpyHostToDevice, s));
    HIPCK(hipStreamSynchronize(s));
    make_frame(host, 1);
    HIPCK(hipMemcpyAsync(img_cur, host.data(), npx * sizeof(float), hipMemcpyHostToDevice, s));
    const float hbox[4] = {W * 0.25f, H * 0.25f, W * 0.5f, H * 0.5f};      // init box xyxy (oracle/synth.py's)
    HIPCK(hipMemcpyAsync(box, hbox, sizeof(hbox), hipMemcpyHostToDevice, s));
    // initialize (unicorn_sot.py:39-55): reference-frame backbone + label map
    UNI(uni_backbone_fpn(ctx, img_ref, 1, H, W, fpn[0][0], fpn[0][1], fpn[0][2], f16_ref, s));
    UNI(uni_label_map_s8(box, lbs, H, W, s));
    UNI(uni_pos_embed(ctx, h, w, pos, s));
    // track (unicorn_sot.py:78-108)
    UNI(uni_backbone_fpn(ctx, img_cur, 1, H, W, fpn[1][0], fpn[1][1], fpn[1][2], f16_cur, s));
    UNI(uni_interaction(ctx, f16_ref, pos, f16_cur, pos, 1, h, w, o_ref, o_cur, s));
    UNI(uni_upsample(ctx, o_ref, 1, h, w, e_ref, s));
    UNI(uni_upsample(ctx, o_cur, 1, h, w, e_cur, s));
    UNI(uni_corr_softmax_pv(e_ref, e_cur, lbs, coarse, (int)R, (int)R, 128, 1, 2, ws, wsb, s));
    UNI(uni_prior_pyramid(coarse, p16, p32, 1, H8, W8, s));
    UNI(uni_head(ctx, fpn[1][0], fpn[1][1], fpn[1][2], coarse, p16, p32, 1, H, W, /*sot*/ 0, out, nullptr, nullptr, nullptr, s));
    std::vector<float> rows(A * nch);
    HIPCK(hipMemcpyAsync(rows.data(), out, rows.size() * sizeof(float), hipMemcpyDeviceToHost, s));
    HIPCK(hipStreamSynchronize(s));
    size_t best = 0;
    double sum = 0.0;
    for (size_t a = 0; a < A; ++a) {
        if (rows[a * nch + 4] * rows[a * nch + 5] > rows[best * nch + 4] * rows[best * nch + 5]) best = a;
        for (int c = 0; c < nch; ++c) sum += rows[a * nch + c];
    }
    FILE* f = fopen(argv[4], "wb");
    if (!f || fwrite(rows.data(), sizeof(float), rows.size(), f) != rows.size()) { fprintf(stderr, "cannot write %s\n", argv[4]); return 4; }
    fclose(f);
    printf("{\"tensors_loaded\": %d, \"anchors\": %zu, \"checksum\": %.9g, \"best_anchor\": %zu, \"best_row\": [%.6f, %.6f, %.6f, %.6f, %.6g, %.6g]}\n", n_loaded, A, sum, best,
           rows[best * nch], rows[best * nch + 1], rows[best * nch + 2], rows[best * nch + 3], rows[best * nch + 4], rows[best * nch + 5]);
    uni_ctx_destroy(ctx);
    return 0;
}
