/*
 * A host program in plain C that drives the denoiser through the C ABI alone -- no Python, no torch in the process:
 * what a cgo / JNI / FFI binding of another host language would do.
 *
 *   client <blob> <out>
 * blob (little endian): int32 n_cfg_fields(19) | 19 x int32 mdt_config fields | float sigma_data |
 *   int32 n_params | per parameter: int32 name_len, name bytes, int64 numel, numel x float |
 *   int32 B, int32 n_steps, (n_steps+1) x float sigmas | tokens | goal | x_T          (MDT-V: tokens (B,n_tok,obs))
 * out: B*Ta*A floats (sampled actions) followed by B*Te*D floats (latent_encoder_emb)
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mdt_hip.h"

#define CHECK(st)                                                                        \
    do {                                                                                 \
        if ((st) != MDT_OK) { fprintf(stderr, "mdt error: %s\n", mdt_last_error()); return 2; } \
    } while (0)
#define HIPCHECK(e)                                                                      \
    do {                                                                                 \
        if ((e) != hipSuccess) { fprintf(stderr, "hip error %d at line %d\n", (int)(e), __LINE__); return 3; } \
    } while (0)

static int rd(FILE* f, void* p, size_t n) { return fread(p, 1, n, f) == n ? 0 : -1; }

static float* to_device(FILE* f, size_t n) {
    float* h = (float*)malloc(n * sizeof(float));
    float* d = NULL;
    if (!h || rd(f, h, n * sizeof(float)) || hipMalloc((void**)&d, n * sizeof(float)) != hipSuccess ||
        hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { free(h); return NULL; }
    free(h);
    return d;
}

int main(int argc, char** argv) {
    if (argc != 3) { fprintf(stderr, "usage: client <blob> <out>\n"); return 1; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror("blob"); return 1; }
    int32_t nf = 0, fields[32];
    mdt_config cfg;
    memset(&cfg, 0, sizeof cfg); /* fields behind sigma_data (no_goal_conditioning) keep their defaults */
    if (rd(f, &nf, 4) || nf != 19 || rd(f, fields, 4 * nf) || rd(f, &cfg.sigma_data, 4)) return 1;
    memcpy(&cfg, fields, 4 * nf); /* the 19 int32 fields lead the struct in declaration order */
    mdt_model* m = NULL;
    CHECK(mdt_create(&cfg, &m));
    hipStream_t s;
    HIPCHECK(hipStreamCreate(&s));
    int32_t np = 0;
    if (rd(f, &np, 4)) return 1;
    for (int i = 0; i < np; ++i) {
        int32_t nl = 0;
        char name[512];
        int64_t numel = 0;
        if (rd(f, &nl, 4) || nl <= 0 || nl >= (int)sizeof name || rd(f, name, nl) || rd(f, &numel, 8)) return 1;
        name[nl] = 0;
        float* h = (float*)malloc((size_t)numel * sizeof(float));
        if (!h || rd(f, h, (size_t)numel * sizeof(float))) return 1;
        CHECK(mdt_load_param(m, name, h, numel, s)); /* host pointer: the library stages it */
        HIPCHECK(hipStreamSynchronize(s));
        free(h);
    }
    int32_t B = 0, n_steps = 0;
    float sigmas[128];
    if (rd(f, &B, 4) || rd(f, &n_steps, 4) || n_steps < 1 || n_steps > 64 || rd(f, sigmas, 4 * (n_steps + 1))) return 1;
    const int n_tok = cfg.arch == MDT_ARCH_MDTV ? cfg.n_obs_token : 1;
    const size_t ntok = (size_t)B * n_tok * cfg.obs_dim, ngoal = (size_t)B * cfg.goal_dim;
    const size_t nact = (size_t)B * cfg.action_seq_len * cfg.action_dim;
    const size_t nctx = (size_t)B * (1 + (cfg.arch == MDT_ARCH_MDTV ? cfg.n_obs_token : 2)) * cfg.embed_dim;
    float* tok = to_device(f, ntok);
    float* tok2 = cfg.arch == MDT_ARCH_MDT ? to_device(f, ntok) : NULL;
    float* goal = to_device(f, ngoal);
    float* xT = to_device(f, nact);
    fclose(f);
    if (!tok || !goal || !xT) return 1;
    float *out = NULL, *ctx = NULL;
    HIPCHECK(hipMalloc((void**)&out, nact * sizeof(float)));
    HIPCHECK(hipMalloc((void**)&ctx, nctx * sizeof(float)));
    CHECK(mdt_sample_ddim(m, tok, tok2, goal, MDT_MODALITY_LANG, xT, sigmas, n_steps, B, out, ctx, s));
    HIPCHECK(hipStreamSynchronize(s));
    float* h = (float*)malloc((nact + nctx) * sizeof(float));
    HIPCHECK(hipMemcpy(h, out, nact * sizeof(float), hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(h + nact, ctx, nctx * sizeof(float), hipMemcpyDeviceToHost));
    FILE* o = fopen(argv[2], "wb");
    if (!o || fwrite(h, sizeof(float), nact + nctx, o) != nact + nctx) return 1;
    fclose(o);
    printf("sampled %d chunks (%d steps), %s, %.3f GFLOP per chunk\n", B, n_steps, mdt_version(), mdt_flops_per_chunk(m, n_steps) / 1e9);
    CHECK(mdt_destroy(m));
    return 0;
}
