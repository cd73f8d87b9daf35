/*
 * ORACLE — test infrastructure only.  Nothing under r-yolov4_amd/ may import, link or call this.
 *
 * CPU restatement (plain C, single thread, fp32) of the third-party rotated-box ops the reference
 * calls but does not vendor:
 *     detectron2.layers.nms.nms_rotated                     (call site: lib/general.py:177)
 *     detectron2.layers.rotated_boxes.pairwise_iou_rotated  (call site: test.py:135)
 * Dependency: facebookresearch/detectron2, UN-PINNED git HEAD (docker/Dockerfile:33, Readme.md:51);
 * absent from /root/reference and not installable here.  The algorithm below restates the published
 * detectron2/layers/csrc/box_iou_rotated/box_iou_rotated_utils.h and nms_rotated/nms_rotated_{cpu,cuda}
 * (SURVEY.md Appendix A).  PARITY UNPINNED by the reference itself (it has no tests / golden vectors
 * for this boundary); pinned here instead by the analytic known-answer cases of SURVEY.md §8(c) and an
 * independent float64 Sutherland-Hodgman clip in tests/test_oracle_iou.py.
 *
 * Numerical contract shared with the HIP kernels (so keep sets are bit-exact):
 *   - all pair arithmetic in IEEE fp32, NO fused multiply-add (build with -ffp-contract=off);
 *   - deg->rad and cos/sin in double, rounded to float, times 0.5f  (get_rotated_vertices);
 *   - centre shift computed in double, rounded to float             (single_box_iou_rotated);
 *   - EPS comparisons done in double exactly as written upstream;
 *   - hull sort = the deterministic O(n^2) exchange sort of the CUDA path — the path lib/general.py:177 runs on (the reference calls
 *     nms_rotated on GPU tensors).  detectron2's CPU path orders the same points with std::sort and a comparator that is not a strict
 *     weak order for near-collinear points (so its result is implementation-defined); that variant is built beside this one from the
 *     same file with -DORA_HULL_STDSORT (symbols ora_cpusort_*, the ordering itself in hull_stdsort.cpp = the real std::sort of this
 *     toolchain's libstdc++) ONLY to measure how the two orderings differ on the failure families (tests/test_iou_fuzz.py,
 *     profiles/r05_iou_sort_variants.json) — it is not the parity contract.
 * Suppression predicate: gt_only=1 -> iou > thr (CUDA semantics, what the reference runs on GPU),
 *                        gt_only=0 -> iou >= thr (detectron2 CPU kernel semantics).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x, y; } pt_t;

#ifdef ORA_HULL_STDSORT
/* second build of this file: detectron2's CPU-path hull ordering; every exported symbol gets the ora_cpusort_ prefix */
void ora_hull_stdsort(pt_t *q_from_1, int n_minus_1);                      /* hull_stdsort.cpp: std::sort(q + 1, q + n, comparator) */
#define ora_single_box_iou_rotated ora_cpusort_single_box_iou_rotated
#define ora_pairwise_iou_rotated   ora_cpusort_pairwise_iou_rotated
#define ora_diag_iou_rotated       ora_cpusort_diag_iou_rotated
#define ora_nms_rotated            ora_cpusort_nms_rotated
#define ora_nms_mask               ora_cpusort_nms_mask
#endif

static inline float dot2(pt_t a, pt_t b) { return a.x * b.x + a.y * b.y; }
static inline float cross2(pt_t a, pt_t b) { return a.x * b.y - b.x * a.y; }
static inline pt_t sub(pt_t a, pt_t b) { pt_t r = {a.x - b.x, a.y - b.y}; return r; }

static void rotated_vertices(const float *b, pt_t *p)
{
    /* box = (x_ctr, y_ctr, w, h, angle_deg) */
    double theta = (double)b[4] * 0.01745329251;
    float c2 = (float)cos(theta) * 0.5f;
    float s2 = (float)sin(theta) * 0.5f;
    p[0].x = b[0] + s2 * b[3] + c2 * b[2];
    p[0].y = b[1] + c2 * b[3] - s2 * b[2];
    p[1].x = b[0] - s2 * b[3] + c2 * b[2];
    p[1].y = b[1] - c2 * b[3] - s2 * b[2];
    p[2].x = 2 * b[0] - p[0].x;
    p[2].y = 2 * b[1] - p[0].y;
    p[3].x = 2 * b[0] - p[1].x;
    p[3].y = 2 * b[1] - p[1].y;
}

static int intersection_points(const pt_t *p1, const pt_t *p2, pt_t *out)
{
    pt_t v1[4], v2[4];
    const double EPS = 1e-5;
    int num = 0;
    for (int i = 0; i < 4; i++) {
        v1[i] = sub(p1[(i + 1) % 4], p1[i]);
        v2[i] = sub(p2[(i + 1) % 4], p2[i]);
    }
    for (int i = 0; i < 4; i++) {
        for (int j = 0; j < 4; j++) {
            float det = cross2(v2[j], v1[i]);
            if (fabs((double)det) <= 1e-14) continue;
            pt_t v12 = sub(p2[j], p1[i]);
            float t1 = cross2(v2[j], v12) / det;
            float t2 = cross2(v1[i], v12) / det;
            if ((double)t1 > -EPS && (double)t1 < (double)1.0f + EPS &&
                (double)t2 > -EPS && (double)t2 < (double)1.0f + EPS) {
                out[num].x = p1[i].x + v1[i].x * t1;
                out[num].y = p1[i].y + v1[i].y * t1;
                num++;
            }
        }
    }
    /* vertices of rect1 inside rect2, then the reverse */
    for (int pass = 0; pass < 2; pass++) {
        const pt_t *pa = pass ? p2 : p1;       /* points tested     */
        const pt_t *pb = pass ? p1 : p2;       /* containing rect   */
        const pt_t *vb = pass ? v1 : v2;
        pt_t AB = vb[0], DA = vb[3];
        float ABdotAB = dot2(AB, AB);
        float ADdotAD = dot2(DA, DA);
        for (int i = 0; i < 4; i++) {
            pt_t AP = sub(pa[i], pb[0]);
            float APdotAB = dot2(AP, AB);
            float APdotAD = -dot2(AP, DA);
            if (((double)APdotAB > -EPS) && ((double)APdotAD > -EPS) &&
                ((double)APdotAB < (double)ABdotAB + EPS) && ((double)APdotAD < (double)ADdotAD + EPS)) {
                out[num++] = pa[i];
            }
        }
    }
    return num;
}

static int convex_hull_graham(const pt_t *p, int n, pt_t *q)
{
    int t = 0;
    for (int i = 1; i < n; i++)
        if (p[i].y < p[t].y || (p[i].y == p[t].y && p[i].x < p[t].x)) t = i;
    pt_t start = p[t];
    for (int i = 0; i < n; i++) q[i] = sub(p[i], start);
    pt_t tmp = q[0]; q[0] = q[t]; q[t] = tmp;

    float dist[24];
#ifdef ORA_HULL_STDSORT
    /* CPU path of box_iou_rotated_utils.h: std::sort(q + 1, q + num_in, cmp), cmp(A, B) = |cross(A, B)| < 1e-6 ? |A|^2 < |B|^2 : cross(A, B) > 0;
     * the squared distances are computed AFTER the sort there */
    ora_hull_stdsort(q + 1, n - 1);
    for (int i = 0; i < n; i++) dist[i] = dot2(q[i], q[i]);
    for (int i = n; i < n - 1; i++) {                                      /* (the exchange sort below is the CUDA path: skipped) */
#else
    for (int i = 0; i < n; i++) dist[i] = dot2(q[i], q[i]);
    for (int i = 1; i < n - 1; i++) {
#endif
        for (int j = i + 1; j < n; j++) {
            float cp = cross2(q[i], q[j]);
            if ((cp < -1e-6) || (fabs((double)cp) < 1e-6 && dist[i] > dist[j])) {
                pt_t qt = q[i]; q[i] = q[j]; q[j] = qt;
                float dt = dist[i]; dist[i] = dist[j]; dist[j] = dt;
            }
        }
    }
    int k;
    for (k = 1; k < n; k++)
        if (dist[k] > 1e-8) break;
    if (k == n) { q[0] = p[t]; return 1; }
    q[1] = q[k];
    int m = 2;
    for (int i = k + 1; i < n; i++) {
        while (m > 1) {
            pt_t q1 = sub(q[i], q[m - 2]), q2 = sub(q[m - 1], q[m - 2]);
            float a = q1.x * q2.y, b = q2.x * q1.y;   /* two roundings, no FMA, on purpose */
            if (a >= b) m--; else break;
        }
        q[m++] = q[i];
    }
    return m;   /* shift_to_zero = true: area only */
}

static float polygon_area(const pt_t *q, int m)
{
    if (m <= 2) return 0.f;
    float area = 0.f;
    for (int i = 1; i < m - 1; i++)
        area += fabsf(cross2(sub(q[i], q[0]), sub(q[i + 1], q[0])));
    return (float)((double)area / 2.0);
}

float ora_single_box_iou_rotated(const float *b1_raw, const float *b2_raw)
{
    float b1[5], b2[5];
    double sx = ((double)(b1_raw[0] + b2_raw[0])) / 2.0;
    double sy = ((double)(b1_raw[1] + b2_raw[1])) / 2.0;
    b1[0] = (float)((double)b1_raw[0] - sx); b1[1] = (float)((double)b1_raw[1] - sy);
    b2[0] = (float)((double)b2_raw[0] - sx); b2[1] = (float)((double)b2_raw[1] - sy);
    for (int k = 2; k < 5; k++) { b1[k] = b1_raw[k]; b2[k] = b2_raw[k]; }

    float area1 = b1[2] * b1[3], area2 = b2[2] * b2[3];
    if ((double)area1 < 1e-14 || (double)area2 < 1e-14) return 0.f;

    pt_t p1[4], p2[4], ipts[24], hull[24];
    rotated_vertices(b1, p1);
    rotated_vertices(b2, p2);
    int num = intersection_points(p1, p2, ipts);
    if (num <= 2) return 0.f;
    int m = convex_hull_graham(ipts, num, hull);
    float inter = polygon_area(hull, m);
    return inter / (area1 + area2 - inter);
}

/* pairwise_iou_rotated: [N,5] x [M,5] -> [N,M] row-major (test.py:135) */
void ora_pairwise_iou_rotated(const float *b1, int n, const float *b2, int m, float *out)
{
    for (int i = 0; i < n; i++)
        for (int j = 0; j < m; j++)
            out[(size_t)i * m + j] = ora_single_box_iou_rotated(b1 + 5 * i, b2 + 5 * j);
}

/* element-wise form (pair k = b1[k] vs b2[k]): the oracle side of the large-scale fuzz (tests/test_iou_fuzz.py) and of L8's diagonal SkewIoU */
void ora_diag_iou_rotated(const float *b1, const float *b2, int64_t n, float *out)
{
    for (int64_t i = 0; i < n; i++) out[i] = ora_single_box_iou_rotated(b1 + 5 * i, b2 + 5 * i);
}

/* nms_rotated(boxes[N,5] deg, scores[N], thr) -> keep indices into the input, in score-desc order.
 * Sort: score descending, ties by ascending original index (the build's fixed tie-break).
 * Lazy greedy exactly as detectron2's CPU kernel: IoU only against boxes that survive.             */
typedef struct { float s; int64_t i; } si_t;
static int cmp_si(const void *a, const void *b)
{
    const si_t *x = a, *y = b;
    if (x->s > y->s) return -1;
    if (x->s < y->s) return 1;
    return (x->i > y->i) - (x->i < y->i);
}

int64_t ora_nms_rotated(const float *boxes, const float *scores, int64_t n, float thr, int gt_only, int64_t *keep)
{
    if (n <= 0) return 0;
    si_t *ord = malloc(sizeof(si_t) * n);
    uint8_t *sup = calloc(n, 1);
    for (int64_t i = 0; i < n; i++) { ord[i].s = scores[i]; ord[i].i = i; }
    qsort(ord, n, sizeof(si_t), cmp_si);
    int64_t nk = 0;
    for (int64_t a = 0; a < n; a++) {
        int64_t i = ord[a].i;
        if (sup[i]) continue;
        keep[nk++] = i;
        for (int64_t b = a + 1; b < n; b++) {
            int64_t j = ord[b].i;
            if (sup[j]) continue;
            float ov = ora_single_box_iou_rotated(boxes + 5 * i, boxes + 5 * j);
            if (gt_only ? (ov > thr) : (ov >= thr)) sup[j] = 1;
        }
    }
    free(ord); free(sup);
    return nk;
}

/* Full-mask formulation (what the CUDA kernel computes): used by tests to check that the lazy greedy
 * and the mask+reduce formulation agree, and as the reference for the HIP mask kernel's bit patterns. */
void ora_nms_mask(const float *sorted_boxes, int64_t n, float thr, int gt_only, uint64_t *mask /* n*ceil(n/64) */)
{
    int64_t nw = (n + 63) / 64;
    memset(mask, 0, sizeof(uint64_t) * n * nw);
    for (int64_t i = 0; i < n; i++)
        for (int64_t j = i + 1; j < n; j++) {
            float ov = ora_single_box_iou_rotated(sorted_boxes + 5 * i, sorted_boxes + 5 * j);
            if (gt_only ? (ov > thr) : (ov >= thr)) mask[i * nw + (j >> 6)] |= 1ull << (j & 63);
        }
}
