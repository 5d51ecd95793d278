/* box_giou_oracle.c -- CPU restatement of the reference's generalized 3D IoU in C. TEST INFRASTRUCTURE ONLY
 * (tests/, smoke(), bench.py's cpu_baseline leg).
 *
 * Same algorithm, rounding order and corner conventions as oracle/box_giou_oracle.py (the numpy restatement
 * of utils/box_util.py:655-745 and its helpers :509-653, pinned on tests/golden/giou.npz which the reference's
 * own function produced); this file exists so that the CPU port of the training step does not pay for
 * pure-Python loops.  Pinned against the numpy restatement and the same fixture in tests/test_giou.py.
 * Compiled with -ffp-contract=off: one rounding per source operation.
 */
#include <math.h>
#include <stdint.h>

#define ORACLE_API __attribute__((visibility("default")))

typedef struct { float x, y; } p2;

static int inside(p2 a, p2 b, p2 p) { /* :520-523 */
  return (b.x - a.x) * (p.y - a.y) > (b.y - a.y) * (p.x - a.x);
}

static p2 intersection(p2 cp1, p2 cp2, p2 s, p2 e) { /* :509-517 */
  const float dcx = cp1.x - cp2.x, dcy = cp1.y - cp2.y;
  const float dpx = s.x - e.x, dpy = s.y - e.y;
  const float n1 = cp1.x * cp2.y - cp1.y * cp2.x;
  const float n2 = s.x * e.y - s.y * e.x;
  const float n3 = 1.0f / (dcx * dpy - dcy * dpx);
  p2 r = {(n1 * dpx - n2 * dcx) * n3, (n1 * dpy - n2 * dcy) * n3};
  return r;
}

/* Sutherland-Hodgman, :526-577; returns the vertex count (<= 8 for two quadrilaterals) */
static int polygon_clip(const p2 *subject, const p2 *clipper, p2 *out) {
  p2 buf[2][16];
  int n = 4, cur = 0;
  for (int i = 0; i < 4; ++i) buf[0][i] = subject[i];
  p2 cp1 = clipper[3];
  for (int c = 0; c < 4; ++c) {
    const p2 cp2 = clipper[c];
    const p2 *in = buf[cur];
    p2 *o = buf[cur ^ 1];
    int m = 0;
    p2 s = in[n - 1];
    for (int k = 0; k < n; ++k) {
      const p2 e = in[k];
      if (inside(cp1, cp2, e)) {
        if (!inside(cp1, cp2, s)) o[m++] = intersection(cp1, cp2, s, e);
        o[m++] = e;
      } else if (inside(cp1, cp2, s)) {
        o[m++] = intersection(cp1, cp2, s, e);
      }
      s = e;
    }
    cp1 = cp2;
    cur ^= 1;
    n = m;
    if (n == 0) break;
  }
  for (int i = 0; i < n; ++i) out[i] = buf[cur][i];
  return n;
}

static float edge(const float *c, int i, int j) { /* :580-600 */
  const float dx = c[i * 3] - c[j * 3], dy = c[i * 3 + 1] - c[j * 3 + 1], dz = c[i * 3 + 2] - c[j * 3 + 2];
  return sqrtf(fmaxf(dx * dx + dy * dy + dz * dz, 1e-6f));
}

ORACLE_API void oracle_generalized_box3d_iou(const float *corners1, const float *corners2, const int32_t *nums_k2,
                                             float *out, int nb, int k1n, int k2n, int rotated, int vols_only,
                                             int k2_limit) {
  const float eps = 1e-8f;
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < nb; ++b) {
    for (int i = 0; i < k1n; ++i) {
      const float *c1 = corners1 + ((long)b * k1n + i) * 24;
      for (int j = 0; j < k2n; ++j) {
        const float *c2 = corners2 + ((long)b * k2n + j) * 24;
        float *o = out + ((long)b * k1n + i) * k2n + j;
        const int real = !nums_k2 || j < nums_k2[b];
        const float height = fmaxf(fminf(c1[1], c2[1]) - fmaxf(c1[13], c2[13]), 0.0f); /* :676-678 */
        p2 r1[4], r2[4];                                                                /* :681-686 */
        for (int q = 0; q < 4; ++q) {
          r1[q].x = c1[(3 - q) * 3]; r1[q].y = c1[(3 - q) * 3 + 2];
          r2[q].x = c2[(3 - q) * 3]; r2[q].y = c2[(3 - q) * 3 + 2];
        }
        const float w = fmaxf(fminf(r1[3].x, r2[3].x) - fmaxf(r1[1].x, r2[1].x), 0.0f);
        const float h = fmaxf(fminf(r1[3].y, r2[3].y) - fmaxf(r1[1].y, r2[1].y), 0.0f);
        float area = real ? w * h : 0.0f;
        if (rotated) {
          const int visit = real && area != 0.0f && (k2_limit < 0 || j < k2_limit);
          area = 0.0f;
          if (visit) {
            p2 poly[16];
            const int n = polygon_clip(r1, r2, poly);
            if (n > 0) { /* shoelace: |x . roll(y,1) - y . roll(x,1)| / 2, sums in index order */
              float a = 0.0f, c = 0.0f;
              for (int q = 0; q < n; ++q) {
                const int prev = (q + n - 1) % n;
                a += poly[q].x * poly[prev].y;
                c += poly[q].y * poly[prev].x;
              }
              area = fabsf(a - c) * 0.5f;
            }
          }
        }
        const float inter_vol = area * height;
        if (vols_only) { *o = inter_vol; continue; }
        float xmin = INFINITY, xmax = -INFINITY, zmin = INFINITY, zmax = -INFINITY;
        float y1max = -INFINITY, y1min = INFINITY, y2max = -INFINITY, y2min = INFINITY;
        for (int q = 0; q < 8; ++q) { /* :603-653, Y flipped */
          xmin = fminf(xmin, fminf(c1[q * 3], c2[q * 3]));
          xmax = fmaxf(xmax, fmaxf(c1[q * 3], c2[q * 3]));
          zmin = fminf(zmin, fminf(c1[q * 3 + 2], c2[q * 3 + 2]));
          zmax = fmaxf(zmax, fmaxf(c1[q * 3 + 2], c2[q * 3 + 2]));
          y1max = fmaxf(y1max, -c1[q * 3 + 1]); y1min = fminf(y1min, -c1[q * 3 + 1]);
          y2max = fmaxf(y2max, -c2[q * 3 + 1]); y2min = fminf(y2min, -c2[q * 3 + 1]);
        }
        const float dx = fabsf(xmax - xmin), dz = fabsf(zmax - zmin);
        const float dy = fabsf(fminf(y1min, y2min) - fmaxf(y1max, y2max));
        const float enclosing = dx * dy * dz;
        const float v1 = fmaxf(edge(c1, 0, 1) * edge(c1, 1, 2) * edge(c1, 0, 4), eps);
        const float v2 = fmaxf(edge(c2, 0, 1) * edge(c2, 1, 2) * edge(c2, 0, 4), eps);
        const float sum_vols = v1 + v2;
        const int good = enclosing > 2 * eps && sum_vols > 4 * eps;
        const float uni = fmaxf(sum_vols - inter_vol, eps);
        const float giou = inter_vol / uni - (1.0f - uni / enclosing);
        *o = (good && real) ? giou : 0.0f;
      }
    }
  }
}
