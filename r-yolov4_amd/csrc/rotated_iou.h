// Rotated-box IoU for gfx950 — the pair function shared by the NMS mask kernel, pairwise IoU and the
// diagonal SkewIoU score.
//
// Replaces detectron2's single_box_iou_rotated<float> (box_iou_rotated_utils.h; third-party, un-vendored,
// call sites lib/general.py:177 and test.py:135 of the reference).  Semantics per SURVEY.md Appendix A.
//
// MI355X-first restructuring (results stay bit-identical to the scalar restatement in oracle/rotated_iou.c):
//   * everything that depends on ONE box only (deg->rad, double cos/sin, the four half-extent products, area,
//     circumscribed radius) is hoisted into a per-box prologue (BoxPrep, 48 B = 3 x dwordx4 loads), so the
//     O(N^2) pair loop never touches double trig;
//   * the pair function keeps the upstream association order of every float add (no FMA contraction: this file
//     is compiled with -ffp-contract=off) so keep-sets match the oracle bit for bit.
// RY_HD lets tests compile this header for the host (g++) to check the restructured arithmetic on CPU.
#pragma once
#include <math.h>

#ifdef __HIPCC__
#define RY_HD __host__ __device__ __forceinline__
#else
#define RY_HD static inline
#endif

struct BoxPrep {
    float x, y, w, h;        // centre (with class offset), size
    float sh, cw, ch, sw;    // (sin/2)*h, (cos/2)*w, (cos/2)*h, (sin/2)*w
    float area, rad, pad0, pad1;
};

struct Pt { float x, y; };

RY_HD void box_prep(const float* b, BoxPrep& o)
{
    const double theta = (double)b[4] * 0.01745329251;
    const float c2 = (float)cos(theta) * 0.5f;
    const float s2 = (float)sin(theta) * 0.5f;
    o.x = b[0]; o.y = b[1]; o.w = b[2]; o.h = b[3];
    o.sh = s2 * b[3]; o.cw = c2 * b[2]; o.ch = c2 * b[3]; o.sw = s2 * b[2];
    o.area = b[2] * b[3];
    // half diagonal, inflated: used only for the conservative disjointness test
    o.rad = 0.5f * sqrtf(b[2] * b[2] + b[3] * b[3]) * 1.001f + 0.02f;
    o.pad0 = 0.f; o.pad1 = 0.f;
}

// Conservative reject: circumscribed circles (inflated) do not meet => the boxes are disjoint and the exact
// algorithm returns 0 (or a ~1e-10 artefact of its EPS slack).  Only used when iou_threshold > 1e-6.
RY_HD bool boxes_far_apart(const BoxPrep& A, const BoxPrep& B)
{
    const float dx = A.x - B.x, dy = A.y - B.y;
    const float r = A.rad + B.rad;
    return dx * dx + dy * dy > r * r;
}

RY_HD float ry_cross(Pt a, Pt b) { return a.x * b.y - b.x * a.y; }
RY_HD float ry_dot(Pt a, Pt b) { return a.x * b.x + a.y * b.y; }
RY_HD Pt ry_sub(Pt a, Pt b) { Pt r; r.x = a.x - b.x; r.y = a.y - b.y; return r; }

RY_HD void ry_vertices(float cx, float cy, const BoxPrep& P, Pt* p)
{
    p[0].x = cx + P.sh + P.cw;
    p[0].y = cy + P.ch - P.sw;
    p[1].x = cx - P.sh + P.cw;
    p[1].y = cy - P.ch - P.sw;
    p[2].x = 2 * cx - p[0].x;
    p[2].y = 2 * cy - p[0].y;
    p[3].x = 2 * cx - p[1].x;
    p[3].y = 2 * cy - p[1].y;
}

RY_HD float rotated_iou_pair(const BoxPrep& A, const BoxPrep& B)
{
    const double EPS = 1e-5;
    if ((double)A.area < 1e-14 || (double)B.area < 1e-14) return 0.f;

    const double sx = ((double)(A.x + B.x)) / 2.0;
    const double sy = ((double)(A.y + B.y)) / 2.0;
    Pt p1[4], p2[4], v1[4], v2[4];
    ry_vertices((float)((double)A.x - sx), (float)((double)A.y - sy), A, p1);
    ry_vertices((float)((double)B.x - sx), (float)((double)B.y - sy), B, p2);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        v1[i] = ry_sub(p1[(i + 1) & 3], p1[i]);
        v2[i] = ry_sub(p2[(i + 1) & 3], p2[i]);
    }

    Pt ip[24];
    int num = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float det = ry_cross(v2[j], v1[i]);
            if (fabs((double)det) <= 1e-14) continue;
            const Pt v12 = ry_sub(p2[j], p1[i]);
            const float t1 = ry_cross(v2[j], v12) / det;
            const float t2 = ry_cross(v1[i], v12) / det;
            if ((double)t1 > -EPS && (double)t1 < (double)1.0f + EPS &&
                (double)t2 > -EPS && (double)t2 < (double)1.0f + EPS) {
                ip[num].x = p1[i].x + v1[i].x * t1;
                ip[num].y = p1[i].y + v1[i].y * t1;
                num++;
            }
        }
    }
    // vertices of rect1 inside rect2
    {
        const Pt AB = v2[0], DA = v2[3];
        const float ABdotAB = ry_dot(AB, AB), ADdotAD = ry_dot(DA, DA);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const Pt AP = ry_sub(p1[i], p2[0]);
            const float a = ry_dot(AP, AB), d = -ry_dot(AP, DA);
            if (((double)a > -EPS) && ((double)d > -EPS) && ((double)a < (double)ABdotAB + EPS) &&
                ((double)d < (double)ADdotAD + EPS)) { ip[num] = p1[i]; num++; }
        }
    }
    // vertices of rect2 inside rect1
    {
        const Pt AB = v1[0], DA = v1[3];
        const float ABdotAB = ry_dot(AB, AB), ADdotAD = ry_dot(DA, DA);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const Pt AP = ry_sub(p2[i], p1[0]);
            const float a = ry_dot(AP, AB), d = -ry_dot(AP, DA);
            if (((double)a > -EPS) && ((double)d > -EPS) && ((double)a < (double)ABdotAB + EPS) &&
                ((double)d < (double)ADdotAD + EPS)) { ip[num] = p2[i]; num++; }
        }
    }
    if (num <= 2) return 0.f;

    // Graham hull (area only, points kept relative to the pivot)
    int t = 0;
    for (int i = 1; i < num; i++)
        if (ip[i].y < ip[t].y || (ip[i].y == ip[t].y && ip[i].x < ip[t].x)) t = i;
    const Pt start = ip[t];
    Pt q[24];
    float dist[24];
    for (int i = 0; i < num; i++) q[i] = ry_sub(ip[i], start);
    { const Pt tmp = q[0]; q[0] = q[t]; q[t] = tmp; }
    for (int i = 0; i < num; i++) dist[i] = ry_dot(q[i], q[i]);
    for (int i = 1; i < num - 1; i++) {
        for (int j = i + 1; j < num; j++) {
            const float cp = ry_cross(q[i], q[j]);
            if (((double)cp < -1e-6) || (fabs((double)cp) < 1e-6 && dist[i] > dist[j])) {
                const Pt qt = q[i]; q[i] = q[j]; q[j] = qt;
                const float dt = dist[i]; dist[i] = dist[j]; dist[j] = dt;
            }
        }
    }
    int k = 1;
    for (; k < num; k++)
        if ((double)dist[k] > 1e-8) break;
    if (k == num) return 0.f;            // hull is a single point -> area 0
    q[1] = q[k];
    int m = 2;
    for (int i = k + 1; i < num; i++) {
        while (m > 1) {
            const Pt q1 = ry_sub(q[i], q[m - 2]), q2 = ry_sub(q[m - 1], q[m - 2]);
            const float a = q1.x * q2.y, b = q2.x * q1.y;
            if (a >= b) m--; else break;
        }
        q[m++] = q[i];
    }
    if (m <= 2) return 0.f;
    float area = 0.f;
    for (int i = 1; i < m - 1; i++)
        area += fabsf(ry_cross(ry_sub(q[i], q[0]), ry_sub(q[i + 1], q[0])));
    const float inter = (float)((double)area / 2.0);
    return inter / (A.area + B.area - inter);
}
