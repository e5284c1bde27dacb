/* TEST INFRASTRUCTURE ONLY -- plain-C restatement of the CCDM per-voxel reverse step.
 *
 * Restates, for one voxel at a time and in a fixed left-to-right fp32 evaluation order:
 *   DiffusionModel.theta_post_prob     ccdm/ddpm/models/diffusion_denoising.py:105-139
 *   torch.clamp(probs, min=1e-12)      ccdm/ddpm/models/diffusion_denoising.py:216
 *   OneHotCategoricalBCHW.sample()     ccdm/ddpm/models/one_hot_categorical.py:30-32
 *     == torch.multinomial(p,1,True) == argmax_k (p_k/sum p)/E_k, E~Exp(1), first max wins
 *   max_prob_sample / prob_sample      ccdm/ddpm/models/one_hot_categorical.py:46-55 (E == NULL)
 *
 * Layout: channels-last, p0[m*K+k], E[m*K+k], one int32 label per voxel.
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (see oracle/Makefile).
 * The HIP kernel gg_ccdm_posterior_sample evaluates the same expressions in the same order
 * (also with contraction off), so labels must agree bit-for-bit.
 */
#include <stdint.h>
#include <stddef.h>

#define GG_MAXK 64

void gg_oracle_ccdm_posterior_sample(const float *p0, const int32_t *xt, const float *E, float a, float abar,
                                     int K, int64_t M, int32_t *labels_out, float *probs_out)
{
    const float Kf = (float)K;
    const float u = (1.0f - a) / Kf;       /* (1 - alphas_t) / num_classes      */
    const float v = (1.0f - abar) / Kf;    /* (1 - cumalphas_tm1) / num_classes */
    const float bd = abar * 1.0f + v;      /* B[c,c] */
    const float bo = abar * 0.0f + v;      /* B[c,d], c != d */
    for (int64_t m = 0; m < M; ++m) {
        const int x = xt[m];
        float A[GG_MAXK], out[GG_MAXK];
        for (int c = 0; c < K; ++c) {
            A[c] = a * (c == x ? 1.0f : 0.0f) + u;
            out[c] = 0.0f;
        }
        for (int d = 0; d < K; ++d) {
            float den = 0.0f;
            for (int c = 0; c < K; ++c) den = den + A[c] * (c == d ? bd : bo);
            const float pd = p0[m * K + d];
            for (int c = 0; c < K; ++c) {
                const float post = (A[c] * (c == d ? bd : bo)) / den;
                out[c] = out[c] + post * pd;
            }
        }
        float s = 0.0f;
        for (int c = 0; c < K; ++c) {
            if (out[c] < 1e-12f) out[c] = 1e-12f;
            s = s + out[c];
        }
        int best = 0;
        float bestv = -1.0f;
        for (int c = 0; c < K; ++c) {
            const float pn = out[c] / s;
            const float r = E ? pn / E[m * K + c] : pn;
            if (probs_out) probs_out[m * K + c] = pn;
            if (r > bestv) { bestv = r; best = c; }
        }
        labels_out[m] = best;
    }
}

/* softmax over K logits in fp32 (nn.Softmax(dim=1), ccdm/ddpm/models/unet_openai/unet.py:715-721) */
#include <math.h>
void gg_oracle_softmax_lastdim(const float *logits, int K, int64_t M, float *probs)
{
    for (int64_t m = 0; m < M; ++m) {
        float mx = logits[m * K];
        for (int c = 1; c < K; ++c) if (logits[m * K + c] > mx) mx = logits[m * K + c];
        float s = 0.0f;
        for (int c = 0; c < K; ++c) { float e = expf(logits[m * K + c] - mx); probs[m * K + c] = e; s = s + e; }
        for (int c = 0; c < K; ++c) probs[m * K + c] = probs[m * K + c] / s;
    }
}
