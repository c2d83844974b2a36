/* TEST INFRASTRUCTURE: runs the C oracle (oracle/alt_corr_oracle.c) under AddressSanitizer + UBSan on buffers of
 * exactly the documented sizes, at the edge cases the parity tests use: coordinates far outside the map on every side,
 * exactly integer coordinates, channel counts that are not multiples of the reference's chunk of 32, a target map of a
 * different size than the query map, radius 0, one-pixel maps.  Built and run by tests/test_sanitizers.py
 * (gcc -fsanitize=address,undefined -fno-sanitize-recover); any report aborts with a non-zero exit status. */
#include <stdio.h>
#include <stdlib.h>
#include <math.h>

void alt_corr_forward_ref(const float*, const float*, const float*, float*, int, int, int, int, int, int, int, int);
void alt_corr_backward_ref(const float*, const float*, const float*, const float*, float*, float*, float*,
                           int, int, int, int, int, int, int, int);

static unsigned long long s = 88172645463325252ull;
static float rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (float)((s >> 11) % 200001) / 100000.0f - 1.0f; }
static float* buf(size_t n, int fill) {
    float* p = (float*)malloc((n ? n : 1) * sizeof(float));       /* exact size: the red zone starts right behind it */
    if (!p) { fprintf(stderr, "out of memory\n"); exit(2); }
    for (size_t i = 0; i < n; i++) p[i] = fill ? rnd() : NAN;
    return p;
}

static int run(int B, int N, int H1, int W1, int H2, int W2, int C, int r, int mode)
{
    const int rd = 2 * r + 1;
    const size_t q = (size_t)B * H1 * W1, t = (size_t)B * H2 * W2;
    float *f1 = buf(q * C, 1), *f2 = buf(t * C, 1), *co = buf((size_t)B * N * H1 * W1 * 2, 1);
    float *out = buf((size_t)B * N * rd * rd * H1 * W1, 0), *gr = buf((size_t)B * N * rd * rd * H1 * W1, 1);
    float *g1 = buf(q * C, 0), *g2 = buf(t * C, 0), *gc = buf((size_t)B * N * H1 * W1 * 2, 0);
    for (size_t i = 0; i < (size_t)B * N * H1 * W1; i++) {
        float x = rnd(), y = rnd();
        switch (mode) {
        case 0: x = (x + 1) * 0.5f * (W2 - 1); y = (y + 1) * 0.5f * (H2 - 1); break;                 /* inside */
        case 1: x = floorf((x + 1) * 0.5f * W2); y = floorf((y + 1) * 0.5f * H2); break;             /* integers */
        case 2: x = x * 3.0f * (W2 + r + 2); y = y * 3.0f * (H2 + r + 2); break;                     /* all sides, far out */
        default: x = x * 1.0e6f; y = y * 1.0e6f; break;                                              /* very far out */
        }
        co[2 * i] = x; co[2 * i + 1] = y;
    }
    alt_corr_forward_ref(f1, f2, co, out, B, N, H1, W1, H2, W2, C, r);
    alt_corr_backward_ref(f1, f2, co, gr, g1, g2, gc, B, N, H1, W1, H2, W2, C, r);
    int bad = 0;
    for (size_t i = 0; i < (size_t)B * N * rd * rd * H1 * W1; i++) bad |= !isfinite(out[i]);      /* every cell written */
    for (size_t i = 0; i < q * C; i++) bad |= !isfinite(g1[i]);
    for (size_t i = 0; i < t * C; i++) bad |= !isfinite(g2[i]);
    for (size_t i = 0; i < (size_t)B * N * H1 * W1 * 2; i++) bad |= (gc[i] != 0.0f);               /* zeros, :307 */
    free(f1); free(f2); free(co); free(out); free(gr); free(g1); free(g2); free(gc);
    return bad;
}

int main(void)
{
    static const int shapes[][8] = {   /* B N H1 W1 H2 W2 C r */
        {1, 1, 5, 7, 5, 7, 32, 4}, {2, 1, 6, 9, 3, 4, 40, 3}, {1, 2, 4, 4, 2, 2, 7, 1}, {1, 1, 1, 1, 1, 1, 1, 0},
        {1, 1, 3, 5, 1, 2, 65, 4}, {2, 3, 2, 3, 7, 5, 16, 2}, {1, 1, 9, 2, 9, 2, 33, 0},
    };
    int cases = 0, bad = 0;
    for (unsigned k = 0; k < sizeof(shapes) / sizeof(shapes[0]); k++)
        for (int mode = 0; mode < 4; mode++, cases++) {
            const int* v = shapes[k];
            const int b = run(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], mode);
            if (b) fprintf(stderr, "shape %u mode %d: unwritten or non-finite output\n", k, mode);
            bad += b;
        }
    printf("oracle under asan+ubsan: %d cases, %d bad\n", cases, bad);
    return bad ? 1 : 0;
}
