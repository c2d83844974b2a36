/* TEST INFRASTRUCTURE ONLY -- plain-C restatement of the reference's on-the-fly
 * correlation kernels.  Never linked into the product library.
 *
 * Follows /root/reference/alt_cuda_corr/correlation_kernel.cu:
 *   corr_forward_kernel   :19-119  -> alt_corr_forward_ref
 *   corr_backward_kernel  :123-256 -> alt_corr_backward_ref
 * including its order of arithmetic: channels are processed in chunks of
 * CHANNEL_STRIDE = 32 (:10,:43), each chunk's partial dot product s (:88-90) is
 * multiplied by the four bilinear weights (:97-100) and ADDED to the output cell
 * (:104-114), so a cell is a float32 sum of (C/32) * (up to 4) products.
 * The reference CUDA source itself cannot be built in this image (no nvcc, no
 * CUDA device) and is not hipified; this file is the CPU stand-in for it.
 *
 * Layouts (all float32, contiguous):
 *   fmap1 (B,H1,W1,C)  fmap2 (B,H2,W2,C)  coords (B,N,H1,W1,2) = (x,y) pixels
 *   corr / corr_grad (B,N,rd*rd,H1,W1), channel = iy + rd*ix  (:92-95)
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC (see oracle/Makefile).
 */
#include <math.h>
#include <string.h>
#include <stddef.h>

#define CHANNEL_STRIDE 32

static int within(int h, int w, int H, int W) { return h >= 0 && h < H && w >= 0 && w < W; }

void alt_corr_forward_ref(const float* fmap1, const float* fmap2, const float* coords, float* corr,
                          int B, int N, int H1, int W1, int H2, int W2, int C, int r)
{
    const int rd = 2 * r + 1;
    const size_t plane = (size_t)H1 * W1;
    memset(corr, 0, sizeof(float) * (size_t)B * N * rd * rd * plane);   /* torch::zeros, :273 */
    for (int b = 0; b < B; b++)
    for (int h1 = 0; h1 < H1; h1++)
    for (int w1 = 0; w1 < W1; w1++) {
        const float* f1 = fmap1 + (((size_t)b * H1 + h1) * W1 + w1) * C;
        for (int c = 0; c < C; c += CHANNEL_STRIDE) {                    /* :43 */
            const int cn = (C - c < CHANNEL_STRIDE) ? C - c : CHANNEL_STRIDE;
            for (int n = 0; n < N; n++) {                                /* :59 */
                const float* cp = coords + ((((size_t)b * N + n) * H1 + h1) * W1 + w1) * 2;
                const float x = cp[0], y = cp[1];
                const float dx = x - floorf(x);                          /* :67-68 */
                const float dy = y - floorf(y);
                float* out = corr + ((size_t)b * N + n) * rd * rd * plane + (size_t)h1 * W1 + w1;
                for (int iy = 0; iy < rd + 1; iy++)
                for (int ix = 0; ix < rd + 1; ix++) {
                    const int h2 = (int)floorf(y) - r + iy;              /* :75-76 */
                    const int w2 = (int)floorf(x) - r + ix;
                    float s = 0.0f;
                    if (within(h2, w2, H2, W2)) {                        /* :80-83 */
                        const float* f2 = fmap2 + (((size_t)b * H2 + h2) * W2 + w2) * C;
                        for (int k = 0; k < cn; k++) s += f1[c + k] * f2[c + k];   /* :88-90 */
                    }
                    const float nw = s * dy * dx;                        /* :97-100 */
                    const float ne = s * dy * (1 - dx);
                    const float sw = s * (1 - dy) * dx;
                    const float se = s * (1 - dy) * (1 - dx);
                    if (iy > 0 && ix > 0)   out[plane * ((iy - 1) + rd * (ix - 1))] += nw;   /* :104-114 */
                    if (iy > 0 && ix < rd)  out[plane * ((iy - 1) + rd * ix)] += ne;
                    if (iy < rd && ix > 0)  out[plane * (iy + rd * (ix - 1))] += sw;
                    if (iy < rd && ix < rd) out[plane * (iy + rd * ix)] += se;
                }
            }
        }
    }
}

void alt_corr_backward_ref(const float* fmap1, const float* fmap2, const float* coords,
                           const float* corr_grad, float* fmap1_grad, float* fmap2_grad,
                           float* coords_grad,
                           int B, int N, int H1, int W1, int H2, int W2, int C, int r)
{
    const int rd = 2 * r + 1;
    const size_t plane = (size_t)H1 * W1;
    memset(fmap1_grad, 0, sizeof(float) * (size_t)B * H1 * W1 * C);      /* :305-307 */
    memset(fmap2_grad, 0, sizeof(float) * (size_t)B * H2 * W2 * C);
    memset(coords_grad, 0, sizeof(float) * (size_t)B * N * plane * 2);   /* never written, :307 */
    for (int b = 0; b < B; b++)
    for (int h1 = 0; h1 < H1; h1++)
    for (int w1 = 0; w1 < W1; w1++) {
        const float* f1 = fmap1 + (((size_t)b * H1 + h1) * W1 + w1) * C;
        float* g1 = fmap1_grad + (((size_t)b * H1 + h1) * W1 + w1) * C;
        for (int n = 0; n < N; n++) {
            const float* cp = coords + ((((size_t)b * N + n) * H1 + h1) * W1 + w1) * 2;
            const float x = cp[0], y = cp[1];
            const float dx = x - floorf(x);
            const float dy = y - floorf(y);
            const float* gp = corr_grad + ((size_t)b * N + n) * rd * rd * plane + (size_t)h1 * W1 + w1;
            for (int iy = 0; iy < rd + 1; iy++)
            for (int ix = 0; ix < rd + 1; ix++) {
                float g = 0.0f;                                          /* :205-222 */
                if (iy > 0 && ix > 0)   g += gp[plane * ((iy - 1) + rd * (ix - 1))] * dy * dx;
                if (iy > 0 && ix < rd)  g += gp[plane * ((iy - 1) + rd * ix)] * dy * (1 - dx);
                if (iy < rd && ix > 0)  g += gp[plane * (iy + rd * (ix - 1))] * (1 - dy) * dx;
                if (iy < rd && ix < rd) g += gp[plane * (iy + rd * ix)] * (1 - dy) * (1 - dx);
                const int h2 = (int)floorf(y) - r + iy;
                const int w2 = (int)floorf(x) - r + ix;
                if (!within(h2, w2, H2, W2)) continue;                   /* f2 == 0, no scatter (:194-197,:236) */
                const float* f2 = fmap2 + (((size_t)b * H2 + h2) * W2 + w2) * C;
                float* g2 = fmap2_grad + (((size_t)b * H2 + h2) * W2 + w2) * C;
                for (int k = 0; k < C; k++) {                            /* :224-238 */
                    g1[k] += g * f2[k];
                    g2[k] += g * f1[k];
                }
            }
        }
    }
}
