/*
 * Oracle (C restatement) of PointPillarNet.forward for ONE cloud - TEST INFRASTRUCTURE, see oracle/__init__.py.
 * Follows /root/reference/lav/models/point_pillar.py:
 *   grid_locations :70-79, coords.unique(dim=0, return_inverse) :82, decorate :55-68 (+ torch_scatter.scatter_mean),
 *   DynamicPointNet.forward :28-35 (Linear, eval BatchNorm1d, ReLU, twice; torch_scatter.scatter_max), scatter_points :87-90.
 * Scalar float32 code, one thread.  Used as the CPU baseline of bench.py and to check larger clouds than the
 * numpy oracle handles comfortably.  Built by oracle/Makefile into oracle/_build/liboracle.so.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    float min_x, max_x, min_y, max_y, ppm;
    int nx, ny;
} oracle_grid;

/* PointNet parameters exactly as in the state_dict (no folding): layer l = 0,1 */
typedef struct {
    const float *w[2], *b[2], *bn_mean[2], *bn_var[2], *bn_gamma[2], *bn_beta[2];
    int num_input, channels;
    float eps;
} oracle_pointnet;

/* returns 0; outputs: canvas [C][ny][nx] (zeroed here), unique_coords [P][3] = (0, xi, yi), inverse [kept],
 * counts[0] = P, counts[1] = kept.  unique_coords / inverse must hold n rows. */
int oracle_pillar_forward(const float *points, int n, int D, const oracle_grid *g, const oracle_pointnet *net,
                          float *canvas, int *unique_coords, int *inverse, int *counts) {
    const int C = net->channels, K = net->num_input, KY = g->ny + 1, KX = g->nx + 1;
    const long ncell = (long)KX * KY;
    int *cell_rank = (int *)malloc(sizeof(int) * ncell);
    int *kept_idx = (int *)malloc(sizeof(int) * (n > 0 ? n : 1));
    int *kept_key = (int *)malloc(sizeof(int) * (n > 0 ? n : 1));
    memset(cell_rank, 0, sizeof(int) * ncell);
    int kept = 0;
    for (int i = 0; i < n; ++i) { /* :70-79 */
        const float x = points[(long)i * D], y = points[(long)i * D + 1];
        if (!(x >= g->min_x && x < g->max_x && y >= g->min_y && y < g->max_y)) continue;
        volatile float fx = x - g->min_x, fy = y - g->min_y; /* volatile: keep the two roundings separate */
        fx = fx * g->ppm;
        fy = fy * g->ppm;
        const int xi = (int)fx, yi = (int)fy;
        kept_idx[kept] = i;
        kept_key[kept] = xi * KY + yi;
        cell_rank[kept_key[kept]] = 1;
        ++kept;
    }
    int P = 0; /* :82 sorted unique rows (lexicographic == ascending key) */
    for (long k = 0; k < ncell; ++k)
        if (cell_rank[k]) {
            unique_coords[P * 3 + 0] = 0;
            unique_coords[P * 3 + 1] = (int)(k / KY);
            unique_coords[P * 3 + 2] = (int)(k % KY);
            cell_rank[k] = P++;
        }
    float *sum = (float *)calloc((size_t)(P > 0 ? P : 1) * 3, sizeof(float));
    float *cnt = (float *)calloc((size_t)(P > 0 ? P : 1), sizeof(float));
    float *feat = (float *)calloc((size_t)(P > 0 ? P : 1) * C, sizeof(float));
    for (int j = 0; j < kept; ++j) { /* scatter_mean: running float32 sums in point order */
        const int p = cell_rank[kept_key[j]];
        inverse[j] = p;
        const float *pt = points + (long)kept_idx[j] * D;
        for (int d = 0; d < 3; ++d) {
            volatile float s = sum[p * 3 + d] + pt[d];
            sum[p * 3 + d] = s;
        }
        cnt[p] += 1.f;
    }
    float *f = (float *)malloc(sizeof(float) * K), *h1 = (float *)malloc(sizeof(float) * C), *h2 = (float *)malloc(sizeof(float) * C);
    for (int j = 0; j < kept; ++j) {
        const int p = inverse[j];
        const float *pt = points + (long)kept_idx[j] * D;
        const int xi = unique_coords[p * 3 + 1], yi = unique_coords[p * 3 + 2];
        for (int d = 0; d < D; ++d) f[d] = pt[d];
        for (int d = 0; d < 3; ++d) f[D + d] = pt[d] - sum[p * 3 + d] / cnt[p]; /* :62 */
        f[D + 3] = pt[0] - ((float)yi / g->ppm + g->min_x);                        /* :57, sic */
        f[D + 4] = pt[1] - ((float)xi / g->ppm + g->min_y);                        /* :58, sic */
        const float *in = f;
        float *out = h1;
        int kin = K;
        for (int l = 0; l < 2; ++l) {
            for (int c = 0; c < C; ++c) {
                float acc = 0.f;
                for (int k = 0; k < kin; ++k) acc += in[k] * net->w[l][(long)c * kin + k];
                acc += net->b[l][c];
                acc = (acc - net->bn_mean[l][c]) / sqrtf(net->bn_var[l][c] + net->eps) * net->bn_gamma[l][c] + net->bn_beta[l][c];
                out[c] = acc > 0.f ? acc : 0.f;
            }
            in = out;
            out = h2;
            kin = C;
        }
        for (int c = 0; c < C; ++c) /* scatter_max; every value >= 0 */
            if (h2[c] > feat[(long)p * C + c]) feat[(long)p * C + c] = h2[c];
    }
    memset(canvas, 0, sizeof(float) * (size_t)C * g->ny * g->nx);
    for (int p = 0; p < P; ++p) { /* :87-90, later rows overwrite earlier ones on clamped collisions */
        int row = g->ny - 1 - unique_coords[p * 3 + 1], col = unique_coords[p * 3 + 2];
        row = row < 0 ? 0 : (row > g->ny - 1 ? g->ny - 1 : row);
        col = col < 0 ? 0 : (col > g->nx - 1 ? g->nx - 1 : col);
        for (int c = 0; c < C; ++c) canvas[((long)c * g->ny + row) * g->nx + col] = feat[(long)p * C + c];
    }
    counts[0] = P;
    counts[1] = kept;
    free(cell_rank); free(kept_idx); free(kept_key); free(sum); free(cnt); free(feat); free(f); free(h1); free(h2);
    return 0;
}
