// Fused loss (forward + gradient) for gfx950 — SURVEY.md §8a rows L1-L6.
// Replaces ComputeCSLLoss (lib/loss.py:153-331) and ComputeKFIoULoss (:334-492) of the reference, including
// build_targets (:270-331 / :427-492), bbox_ciou (:36-78), KFLoss (:81-150) with xywhr2xywhrsigma (lib/general.py:107-133)
// and norm_angle (lib/general.py:7-20).
//
// The reference spends ~100 tiny launches, 4-5 device->host syncs (.cpu().item(), boolean-mask indexing) and an
// O(n^2) broadcast + batched 2x2 LU in KFLoss per step.  Here the whole loss is 7 kernels (15 launches: K2, K2b, K3, K4 once per scale) with no host round trip:
//   K1 loss_targets_*        32 workgroups per scale, two launches (count, then place): candidate (offset, anchor, target) triples
//                            are tested and compacted IN THE REFERENCE'S ORDER with ballot/popcount prefix sums (bit-exact indices);
//   K2a loss_match_box_kernel  one THREAD per match (r05; lane 0 of a wave per match before): differentiates the box term with
//                            forward-mode dual numbers (CIoU with constant alpha / KFIoU closed form); the box-term gradient of the
//                            match is parked for K2b; duplicate cells are resolved last-writer-wins via an atomicMax owner grid
//                             and linked into a per-cell chain;
//   K2 loss_match_kernel     one wavefront per match: the class / 180-bin CSL BCE with wave64 shuffle reductions, partial sums;
//   K2b loss_match_grad_kernel  the owner of every matched cell sums the gradient terms of the cell's matches in ascending match
//                            order (what autograd's index_put_(accumulate=True) does, in a FIXED order) and stores them;
//   K4 loss_obj_kernel       objectness BCE over every cell against the owner's IoU score (the only HBM-heavy pass: one logit per cell — from the
//                            engine's compact copy, LossParams.headobj, else strided out of the map — and the definition of the gradient map);
//   K5 loss_finalize_kernel  fixed-order sums -> the five loss items, already scaled (lib/loss.py:251-255, :410-413).
// Compiled with -ffp-contract=off so the float comparisons of target assignment match torch-CPU bit for bit.
#include "common.h"
#include "params.h"
#include "rotated_iou.h"

#define PI_F 3.14159265358979323846f


// ------------------------------------------------------------------------------------------------ workspace carving
struct ScaleWs {
    int* count;               // [1]
    int* rec;                 // [cap][8]: b, a, gj, gi, cls, tidx, cell, pad
    float* frec;              // [cap][8]: tbox[0..4], score, pad, pad
    int* owner;               // [cells]: largest match index of the cell (last writer wins), -1: no match
    float* part_match;        // [nblk_match][4]: reg_a, reg_b, cls, theta
    float* part_obj;          // [nblk_obj]
    int* head;                // [cells]: most recently linked match of the cell (-1: none) — chains of duplicate matches
    int* next;                // [cap]:   next match of the same cell
    float* gbox;              // [cap][8]: d(loss)/d(x, y, w, h[, a] logits) of the match
    int cap, cells, nblk_match, nblk_obj;
};
static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

// Layout (r05): [count x 3 | owner x 3, head x 3 | per scale: rec, frec, part_match, part_obj, next, gbox].  Only the first region is cleared (to
// zero) and the second filled with -1 per call — two fills instead of the seven of r02-r04, which also cleared the whole workspace (190 MB at the
// benchmark size) although every other array is written before it is read.  (The objectness target grid `tconf` is gone: the objectness pass
// takes the score from the owner's match record.)
__host__ __device__ static inline void carve(const LossParams& p, ScaleWs* s, size_t* total, size_t* ff_off = nullptr, size_t* ff_bytes = nullptr)
{
    size_t off = 0;
    char* base = reinterpret_cast<char*>(p.ws);
    for (int i = 0; i < 3; i++) {
        const int cap = 5 * p.na * p.nt;
        const int cells = p.batch * p.na * p.gs[i] * p.gs[i];
        s[i].cap = cap; s[i].cells = cells;
        s[i].nblk_match = (cap + 3) / 4;
        s[i].nblk_obj = (cells + 1023) / 1024 > 2048 ? 2048 : (cells + 1023) / 1024;
        if (s[i].nblk_obj < 1) s[i].nblk_obj = 1;
        s[i].count = reinterpret_cast<int*>(base + off); off += 256;
    }
    if (ff_off) *ff_off = off;
    for (int i = 0; i < 3; i++) { s[i].owner = reinterpret_cast<int*>(base + off); off += al256((size_t)s[i].cells * 4); }
    for (int i = 0; i < 3; i++) { s[i].head = reinterpret_cast<int*>(base + off); off += al256((size_t)s[i].cells * 4); }
    if (ff_bytes) *ff_bytes = off - 768;
    for (int i = 0; i < 3; i++) {
        const int cap = s[i].cap;
        s[i].rec = reinterpret_cast<int*>(base + off); off += al256((size_t)cap * 8 * 4);
        s[i].frec = reinterpret_cast<float*>(base + off); off += al256((size_t)cap * 8 * 4);
        s[i].part_match = reinterpret_cast<float*>(base + off); off += al256((size_t)(s[i].nblk_match > 0 ? s[i].nblk_match : 1) * 4 * 4);
        s[i].part_obj = reinterpret_cast<float*>(base + off); off += al256((size_t)s[i].nblk_obj * 4);
        s[i].next = reinterpret_cast<int*>(base + off); off += al256((size_t)cap * 4);
        s[i].gbox = reinterpret_cast<float*>(base + off); off += al256((size_t)cap * 8 * 4);
    }
    *total = off;
}

// ------------------------------------------------------------------------------------------------ K1 targets
// Candidate (offset o, anchor a, target t) of linear index cnd = (o*na + a)*nt + t in the reference's enumeration order
// (lib/loss.py:275-310 / :432-471): does it produce a match on scale i?
struct Cand { bool ok; int o, a, t; float gx, gy, gw, gh; };
__device__ __forceinline__ Cand cand_test(const LossParams& p, int i, int cnd, int total)
{
    Cand c;
    c.ok = false; c.o = 0; c.a = 0; c.t = 0; c.gx = c.gy = c.gw = c.gh = 0.f;
    if (cnd >= total) return c;
    const int na = p.na, nt = p.nt;
    const float fg = (float)p.gs[i];
    c.o = cnd / (na * nt);
    const int r = cnd - c.o * na * nt;
    c.a = r / nt;
    c.t = r - c.a * nt;
    const float* tg = p.targets + (int64_t)c.t * p.tcols;
    // image index outside the batch (a stale collate index, a data-parallel shard whose targets were not re-indexed): the
    // reference raises IndexError at pi[b, ...]; here the row is dropped instead of indexing the head maps out of bounds
    const int tb = (int)tg[0];
    if (tb < 0 || tb >= p.batch) return c;
    c.gx = tg[2] * fg; c.gy = tg[3] * fg; c.gw = tg[4] * fg; c.gh = tg[5] * fg;
    const float aw = p.anchors[i][c.a][0], ah = p.anchors[i][c.a][1];
    const float rw = c.gw / aw, rh = c.gh / ah;
    const float mw = fmaxf(rw, 1.0f / rw), mh = fmaxf(rh, 1.0f / rh);
    bool ok = fmaxf(mw, mh) < 4.0f;                                                    // lib/loss.py:297-298 / :454-455
    if (p.mode != 0) ok = ok && (fabsf(cosf(tg[6] - p.anchors[i][c.a][2])) > 0.866f);     // lib/loss.py:458-461
    if (ok && c.o > 0) {
        const float ix = fg - c.gx, iy = fg - c.gy;                                    // gxi = gain - gxy
        if (c.o == 1) ok = (c.gx - floorf(c.gx) < 0.5f) && (c.gx > 1.0f);
        else if (c.o == 2) ok = (c.gy - floorf(c.gy) < 0.5f) && (c.gy > 1.0f);
        else if (c.o == 3) ok = (ix - floorf(ix) < 0.5f) && (ix > 1.0f);
        else ok = (iy - floorf(iy) < 0.5f) && (iy > 1.0f);
    }
    c.ok = ok;
    return c;
}

// Target assignment in two launches over LT_BLOCKS workgroups per scale (one workgroup per scale walked 360 barrier-separated
// iterations at batch 64: 0.76 ms): pass 1 counts the matches of each workgroup's contiguous candidate range into count[1 + j];
// pass 2 starts each range at the sum of the preceding counts and compacts IN THE REFERENCE'S ORDER (ballot / popcount prefix inside
// the workgroup), so the records — and the last-writer-wins owner resolution downstream — are bit-identical to the serial walk.
#define LT_BLOCKS 32
__device__ __forceinline__ void lt_range(int total, int j, int& lo, int& hi)
{
    const int per = ((total + LT_BLOCKS - 1) / LT_BLOCKS + 1023) / 1024 * 1024;      // whole 1024-candidate iterations
    lo = min(total, j * per);
    hi = min(total, lo + per);
}

__global__ __launch_bounds__(1024) void loss_targets_count_kernel(const LossParams p, ScaleWs s0, ScaleWs s1, ScaleWs s2)
{
    const int i = blockIdx.x, j = blockIdx.y;
    const ScaleWs s = i == 0 ? s0 : (i == 1 ? s1 : s2);
    const int total = 5 * p.na * p.nt;
    int lo, hi;
    lt_range(total, j, lo, hi);
    __shared__ int wave_cnt[16];
    int mine = 0;
    for (int base = lo; base < hi; base += 1024) {
        const Cand c = cand_test(p, i, base + (int)threadIdx.x, hi);
        mine += __popcll(__ballot(c.ok));                                              // identical in every lane of the wave
    }
    if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int w = 0; w < 16; w++) tot += wave_cnt[w];
        s.count[1 + j] = tot;
    }
}

__global__ __launch_bounds__(1024) void loss_targets_kernel(const LossParams p, ScaleWs s0, ScaleWs s1, ScaleWs s2)
{
    const int i = blockIdx.x, j = blockIdx.y;
    const ScaleWs s = i == 0 ? s0 : (i == 1 ? s1 : s2);
    const int gs = p.gs[i];
    const int na = p.na;
    const int total = 5 * na * p.nt;
    int lo, hi;
    lt_range(total, j, lo, hi);
    __shared__ int wave_cnt[16];
    __shared__ int running;
    if (threadIdx.x == 0) {
        int before = 0, all = 0;
        for (int k = 0; k < LT_BLOCKS; k++) {
            const int ck = s.count[1 + k];
            if (k < j) before += ck;
            all += ck;
        }
        running = before;
        if (j == 0) s.count[0] = all;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = lo; base < hi; base += 1024) {
        const Cand c = cand_test(p, i, base + (int)threadIdx.x, hi);
        const bool ok = c.ok;
        const int o = c.o, a = c.a, t = c.t;
        const float gx = c.gx, gy = c.gy, gw = c.gw, gh = c.gh;
        const unsigned long long m = __ballot(ok);
        if (lane == 0) wave_cnt[wave] = __popcll(m);
        __syncthreads();
        int before = running;
        for (int w = 0; w < wave; w++) before += wave_cnt[w];
        if (ok) {
            const int e = before + __popcll(m & ((1ull << lane) - 1ull));
            const float offx = o == 1 ? 0.5f : (o == 3 ? -0.5f : 0.f);
            const float offy = o == 2 ? 0.5f : (o == 4 ? -0.5f : 0.f);
            int gi = (int)(gx - offx), gj = (int)(gy - offy);                              // .long(): trunc toward zero
            gi = min(max(gi, 0), gs - 1);                                                  // clamp_ in place (lib/loss.py:324)
            gj = min(max(gj, 0), gs - 1);
            const float* tg = p.targets + (int64_t)t * p.tcols;
            const int b = (int)tg[0];
            int* r = s.rec + (int64_t)e * 8;
            r[0] = b; r[1] = a; r[2] = gj; r[3] = gi; r[4] = (int)tg[1]; r[5] = t;
            r[6] = ((b * na + a) * gs + gj) * gs + gi;
            r[7] = 0;
            float* f = s.frec + (int64_t)e * 8;
            f[0] = gx - (float)gi; f[1] = gy - (float)gj; f[2] = gw; f[3] = gh;              // tbox (lib/loss.py:325 / :488)
            f[4] = p.mode != 0 ? tg[6] : 0.f;
            f[5] = 0.f; f[6] = 0.f; f[7] = 0.f;
        }
        __syncthreads();
        if (threadIdx.x == 0) { int tot = 0; for (int w = 0; w < 16; w++) tot += wave_cnt[w]; running += tot; }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ dual numbers
template <int N>
struct Dual {
    float v;
    float d[N];
};
template <int N> __device__ __forceinline__ Dual<N> dconst(float c) { Dual<N> r; r.v = c; for (int k = 0; k < N; k++) r.d[k] = 0.f; return r; }
template <int N> __device__ __forceinline__ Dual<N> dvar(float c, int idx) { Dual<N> r = dconst<N>(c); r.d[idx] = 1.f; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator+(Dual<N> a, Dual<N> b) { for (int k = 0; k < N; k++) a.d[k] += b.d[k]; a.v += b.v; return a; }
template <int N> __device__ __forceinline__ Dual<N> operator-(Dual<N> a, Dual<N> b) { for (int k = 0; k < N; k++) a.d[k] -= b.d[k]; a.v -= b.v; return a; }
template <int N> __device__ __forceinline__ Dual<N> operator*(Dual<N> a, Dual<N> b) { Dual<N> r; r.v = a.v * b.v; for (int k = 0; k < N; k++) r.d[k] = a.d[k] * b.v + a.v * b.d[k]; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator/(Dual<N> a, Dual<N> b) { Dual<N> r; r.v = a.v / b.v; for (int k = 0; k < N; k++) r.d[k] = (a.d[k] - r.v * b.d[k]) / b.v; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator*(Dual<N> a, float c) { a.v *= c; for (int k = 0; k < N; k++) a.d[k] *= c; return a; }
template <int N> __device__ __forceinline__ Dual<N> operator+(Dual<N> a, float c) { a.v += c; return a; }
template <int N> __device__ __forceinline__ Dual<N> operator-(Dual<N> a, float c) { a.v -= c; return a; }
template <int N> __device__ __forceinline__ Dual<N> dscale(Dual<N> a, float dv, float v) { for (int k = 0; k < N; k++) a.d[k] *= dv; a.v = v; return a; }   // f(a): value v, f'(a)=dv
// torch.max / torch.min (binary): ties split the gradient evenly
template <int N> __device__ __forceinline__ Dual<N> dmax(Dual<N> a, Dual<N> b)
{
    if (a.v > b.v) return a;
    if (a.v < b.v) return b;
    Dual<N> r; r.v = a.v; for (int k = 0; k < N; k++) r.d[k] = 0.5f * (a.d[k] + b.d[k]); return r;
}
template <int N> __device__ __forceinline__ Dual<N> dmin(Dual<N> a, Dual<N> b)
{
    if (a.v < b.v) return a;
    if (a.v > b.v) return b;
    Dual<N> r; r.v = a.v; for (int k = 0; k < N; k++) r.d[k] = 0.5f * (a.d[k] + b.d[k]); return r;
}
// clamp: gradient passes where min <= x <= max (torch clamp backward)
template <int N> __device__ __forceinline__ Dual<N> dclamp(Dual<N> a, float lo, float hi)
{
    if (a.v < lo) return dconst<N>(lo);
    if (a.v > hi) return dconst<N>(hi);
    return a;
}

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float softplus_neg_abs(float x) { return log1pf(expf(-fabsf(x))); }
// BCEWithLogits element, pos_weight pw: value and d/dx
__device__ __forceinline__ float bce_val(float x, float t, float pw)
{
    const float w = 1.f + (pw - 1.f) * t;
    return (1.f - t) * x + w * (softplus_neg_abs(x) + fmaxf(-x, 0.f));
}
__device__ __forceinline__ float bce_grad(float x, float t, float pw)
{
    const float w = 1.f + (pw - 1.f) * t;
    return (1.f - t) - w * (1.f - sigm(x));
}
// FocalLoss wrapper of lib/loss.py:10-33 (active when hyp['fl_gamma'] > 0, lib/loss.py:167-171 / :345-348; alpha = 0.25 is the
// constructor default the reference uses): element = BCE * (t a + (1 - t)(1 - a)) * (1 - p_t)^gamma, p_t = t p + (1 - t)(1 - p).
// Targets are soft (objectness = IoU score, CSL labels), so the general-t formulas are kept.
__device__ __forceinline__ float fl_val(float x, float t, float pw, float gamma, float alpha)
{
    const float l = bce_val(x, t, pw);
    if (!(gamma > 0.f)) return l;
    const float pr = sigm(x), q = 1.f - (t * pr + (1.f - t) * (1.f - pr));
    return l * (t * alpha + (1.f - t) * (1.f - alpha)) * powf(q, gamma);
}
__device__ __forceinline__ float fl_grad(float x, float t, float pw, float gamma, float alpha)
{
    const float dl = bce_grad(x, t, pw);
    if (!(gamma > 0.f)) return dl;
    const float pr = sigm(x), q = 1.f - (t * pr + (1.f - t) * (1.f - pr));
    const float af = t * alpha + (1.f - t) * (1.f - alpha);
    const float m = powf(q, gamma);
    const float dm = q > 0.f ? -gamma * powf(q, gamma - 1.f) * (2.f * t - 1.f) * pr * (1.f - pr) : 0.f;
    return af * (dl * m + bce_val(x, t, pw) * dm);
}

// CIoU (lib/loss.py:36-78) with alpha constant; inputs 4 duals (x,y,w,h), target floats
__device__ Dual<4> ciou_dual(Dual<4> x1, Dual<4> y1, Dual<4> w1, Dual<4> h1, float x2, float y2, float w2, float h2)
{
    typedef Dual<4> D;
    const D pl = x1 - w1 * 0.5f, pr = x1 + w1 * 0.5f, pt = y1 - h1 * 0.5f, pb = y1 + h1 * 0.5f;
    const D tl = dconst<4>(x2 - w2 / 2), tr = dconst<4>(x2 + w2 / 2), tt = dconst<4>(y2 - h2 / 2), tb = dconst<4>(y2 + h2 / 2);
    const D iw = dclamp(dmin(pr, tr) - dmax(pl, tl), 0.f, INFINITY);
    const D ih = dclamp(dmin(pb, tb) - dmax(pt, tt), 0.f, INFINITY);
    const D inter = iw * ih;
    const D dx = dconst<4>(x2) - x1, dy = dconst<4>(y2) - y1;
    const D d2 = dx * dx + dy * dy;
    const D ow = dclamp(dmax(pr, tr) - dmin(pl, tl), 0.f, INFINITY);
    const D oh = dclamp(dmax(pb, tb) - dmin(pt, tt), 0.f, INFINITY);
    const D c2 = ow * ow + oh * oh;
    const D uni = w1 * h1 + (w2 * h2) - inter;
    const D u = d2 / (c2 + 1e-15f);
    const D iou = inter / (uni + 1e-15f);
    const D ratio = w1 / h1;
    const float at2 = atanf(w2 / h2);
    const D at1 = dscale(ratio, 1.f / (1.f + ratio.v * ratio.v), atanf(ratio.v));
    const D diff = dconst<4>(at2) - at1;
    const D v = diff * diff * 0.4052847345693511f;            // (float)(4 / pi^2)
    const float alpha = v.v / ((1.f - iou.v) + v.v);
    D c = iou - (u + v * alpha);
    if (c.v < -1.f) c = dconst<4>(-1.f);
    if (c.v > 1.f) c = dconst<4>(1.f);
    return c;
}

// KFLoss pieces (lib/loss.py:100-150, fun='exp', alpha=3): returns xy_loss and kf_loss as duals of (x,y,w,h,r), plus KFIoU
__device__ void kf_dual(Dual<5> x, Dual<5> y, Dual<5> w, Dual<5> h, Dual<5> r, const float* t, Dual<5>& xy_loss, Dual<5>& kf_loss,
                        float& kfiou)
{
    typedef Dual<5> D;
    const D wp = dclamp(w, 1e-4f, 1e4f), hp = dclamp(h, 1e-4f, 1e4f);
    const float wt = fminf(fmaxf(t[2], 1e-4f), 1e4f), ht = fminf(fmaxf(t[3], 1e-4f), 1e4f), rt = t[4];
    // Sigma_t = R diag((wt/2)^2, (ht/2)^2) R^T, R = [[c,-s],[s,c]]; inverse in closed form
    const float c = cosf(rt), s = sinf(rt);
    const float a2 = (0.5f * wt) * (0.5f * wt), b2 = (0.5f * ht) * (0.5f * ht);
    const float s00 = c * c * a2 + s * s * b2, s01 = c * s * (a2 - b2), s11 = s * s * a2 + c * c * b2;
    const float det = s00 * s11 - s01 * s01;
    const D dx = x - t[0], dy = y - t[1];
    const D maha = (dx * dx * s11 - dx * dy * (2.f * s01) + dy * dy * s00) * (1.f / det);
    const D m1 = maha + 1.f;
    xy_loss = dscale(m1, 1.f / m1.v, logf(m1.v));
    const D wp2 = wp * wp, hp2 = hp * hp;
    const float wt2 = wt * wt, ht2 = ht * ht;
    const D dr = r - rt;
    const float cd = cosf(dr.v), sd = sinf(dr.v);
    const D cos2 = dscale(dr, -2.f * cd * sd, cd * cd);
    const D sin2 = dscale(dr, 2.f * sd * cd, sd * sd);
    const D A2 = (wp2 * hp2) * (1.f / (wt2 * ht2)) + (wp2 * (1.f / wt2) + hp2 * (1.f / ht2)) * cos2 + (wp2 * (1.f / ht2) + hp2 * (1.f / wt2)) * sin2 + 1.f;
    const D inv_wh = dconst<5>(wt2 * ht2) / (wp2 * hp2);
    const D B2 = inv_wh + (dconst<5>(wt2) / wp2 + dconst<5>(ht2) / hp2) * cos2 + (dconst<5>(wt2) / hp2 + dconst<5>(ht2) / wp2) * sin2 + 1.f;
    const D A = dscale(A2, 0.5f / sqrtf(A2.v), sqrtf(A2.v));
    const D B = dscale(B2, 0.5f / sqrtf(B2.v), sqrtf(B2.v));
    const D den = A + B - 3.f;
    const D k = dconst<5>(1.f) / den;                      // (4 - alpha) / (A + B - alpha), alpha = 3
    kfiou = k.v;
    const float ex = expf(1.f - k.v);
    kf_loss = dscale(k, -ex, ex - 1.f);
}

// ------------------------------------------------------------------------------------------------ K2 per-match
// K2a, box terms: ONE THREAD per match (r05).  The regression loss and its five derivatives are a few thousand dependent scalar operations per
// match (dual numbers through the CIoU / KFIoU formulas); with one WAVE per match and this block under `lane == 0` — the r02-r04 form — the 30 k
// matches of a benchmark-size scale were 30 k single-lane waves and the kernel was bound by instruction issue (121 us per scale).  Same arithmetic
// per match, so every record is bit-identical; reg_a / reg_b travel to K2b's partial sums through the two spare slots of the match record.
__global__ __launch_bounds__(256) void loss_match_box_kernel(const LossParams p, ScaleWs s0, ScaleWs s1, ScaleWs s2)
{
    const int scale = blockIdx.y;
    const ScaleWs s = scale == 0 ? s0 : (scale == 1 ? s1 : s2);
    const int e = blockIdx.x * 256 + threadIdx.x;
    const int n = *s.count;
    if (e >= n) return;
    const int attrs = p.nc + (p.mode == 0 ? 185 : 6);
    float reg_a = 0.f, reg_b = 0.f;
    const int* r = s.rec + (int64_t)e * 8;
    float* f = s.frec + (int64_t)e * 8;
    const int a = r[1], cell = r[6];
    const float* ps = p.head[scale] + (int64_t)cell * attrs;
    const float inv_n = 1.0f / (float)n;
    {
        const float aw = p.anchors[scale][a][0], ah = p.anchors[scale][a][1];
        const float sx = sigm(ps[0]), sy = sigm(ps[1]), sw = sigm(ps[2]), sh = sigm(ps[3]);
        float score, g[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        if (p.mode == 0) {
            const Dual<4> c = ciou_dual(dvar<4>(sx * 2.f - 0.5f, 0), dvar<4>(sy * 2.f - 0.5f, 1),
                                        dvar<4>((sw * 2.f) * (sw * 2.f) * aw, 2), dvar<4>((sh * 2.f) * (sh * 2.f) * ah, 3),
                                        f[0], f[1], f[2], f[3]);
            reg_a = 1.0f - c.v;                                              // (1 - ciou).mean()  lib/loss.py:218
            score = fmaxf(c.v, 0.f);
            const float k = -p.box * inv_n;
            g[0] = k * c.d[0] * 2.f * sx * (1.f - sx);
            g[1] = k * c.d[1] * 2.f * sy * (1.f - sy);
            g[2] = k * c.d[2] * 8.f * sw * sw * (1.f - sw) * aw;
            g[3] = k * c.d[3] * 8.f * sh * sh * (1.f - sh) * ah;
        } else if (p.mode == 2) {
            // smooth-L1-IoU regression (EXTRA mode: the reference names it, Readme.md:4,12-13, but ships no code for it; the
            // definition is this build's): per match  (L_sl1 / |L_sl1|) * |-log(SkewIoU)|  (R3Det, arXiv 1908.05612
            // eq. 5): the smooth-L1 of (x, y, w, h, theta) gives the DIRECTION, the exact rotated IoU of the decoded box against
            // its target (detached) the magnitude; the objectness target is that IoU.
            const float sa = sigm(ps[4]);
            float pa = (sa - 0.5f) * 1.1f + p.anchors[scale][a][2];
            const float hpi = (float)(3.14159265358979323846 / 2);
            if (pa >= hpi) pa = pa - PI_F;
            if (pa < -hpi) pa = pa + PI_F;
            const float pv[5] = {sx * 2.f - 0.5f, sy * 2.f - 0.5f, (sw * 2.f) * (sw * 2.f) * aw, (sh * 2.f) * (sh * 2.f) * ah, pa};
            float S = 0.f, dS[5];
            for (int j = 0; j < 5; j++) {
                const float d = pv[j] - f[j], ad = fabsf(d);
                S += ad < 1.f ? 0.5f * d * d : ad - 0.5f;                    // smooth L1, beta = 1
                dS[j] = ad < 1.f ? d : (d > 0.f ? 1.f : -1.f);
            }
            const float bp[5] = {pv[0], pv[1], pv[2], pv[3], pv[4] * 57.29577951308232f};
            const float bt[5] = {f[0], f[1], f[2], f[3], f[4] * 57.29577951308232f};
            BoxPrep A, Bx;
            box_prep(bp, A);
            box_prep(bt, Bx);
            const float iou = rotated_iou_pair(A, Bx);
            const float w = -logf(fmaxf(iou, 1e-6f));
            reg_a = S > 0.f ? w : 0.f;
            score = fmaxf(iou, 0.f);
            const float k = S > 0.f ? p.box * inv_n * w / S : 0.f;
            g[0] = k * dS[0] * 2.f * sx * (1.f - sx);
            g[1] = k * dS[1] * 2.f * sy * (1.f - sy);
            g[2] = k * dS[2] * 8.f * sw * sw * (1.f - sw) * aw;
            g[3] = k * dS[3] * 8.f * sh * sh * (1.f - sh) * ah;
            g[4] = k * dS[4] * 1.1f * sa * (1.f - sa);
        } else {
            const float sa = sigm(ps[4]);
            float pa = (sa - 0.5f) * 1.1f + p.anchors[scale][a][2];          // lib/loss.py:390
            const float hp = (float)(3.14159265358979323846 / 2);
            if (pa >= hp) pa = pa - PI_F;                                    // norm_angle, lib/general.py:14-15
            if (pa < -hp) pa = pa + PI_F;
            Dual<5> xy, kf;
            float kfiou;
            kf_dual(dvar<5>(sx * 2.f - 0.5f, 0), dvar<5>(sy * 2.f - 0.5f, 1), dvar<5>((sw * 2.f) * (sw * 2.f) * aw, 2),
                    dvar<5>((sh * 2.f) * (sh * 2.f) * ah, 3), dvar<5>(pa, 4), f, xy, kf, kfiou);
            reg_a = fmaxf(xy.v, 0.f);
            reg_b = fmaxf(kf.v, 0.f);
            score = fmaxf(kfiou, 0.f);
            const float k = p.box * inv_n;
            const float d0 = xy.d[0] + kf.d[0], d1 = xy.d[1] + kf.d[1], d2 = xy.d[2] + kf.d[2], d3 = xy.d[3] + kf.d[3],
                        d4 = xy.d[4] + kf.d[4];
            g[0] = k * d0 * 2.f * sx * (1.f - sx);
            g[1] = k * d1 * 2.f * sy * (1.f - sy);
            g[2] = k * d2 * 8.f * sw * sw * (1.f - sw) * aw;
            g[3] = k * d3 * 8.f * sh * sh * (1.f - sh) * ah;
            g[4] = k * d4 * 1.1f * sa * (1.f - sa);
        }
        f[5] = score;
        atomicMax(&s.owner[cell], e);                                        // last writer (largest e) wins
        if (p.compute_grad) {
            float* gb = s.gbox + (int64_t)e * 8;
            for (int k = 0; k < 5; k++) gb[k] = g[k];
            s.next[e] = atomicExch(&s.head[cell], e);                        // link into the cell's chain (any order: K2b sorts)
        }
    }
    f[6] = reg_a;
    f[7] = reg_b;
}

// K2b, class / angle-bin BCE terms: G lanes per match — 64 (a wave per match), or 16 when the classes fit (nc <= 16, no angle bins: four matches
// per wave; the xor butterfly over 16 lanes adds the same values in the same order as the 64-lane one did, whose upper lanes held zeros) — and
// folds the match's box terms into the partial sums: one row of part_match per FOUR consecutive matches, as before.
__global__ __launch_bounds__(256) void loss_match_kernel(const LossParams p, ScaleWs s0, ScaleWs s1, ScaleWs s2, int G)
{
    const int scale = blockIdx.y;
    const ScaleWs s = scale == 0 ? s0 : (scale == 1 ? s1 : s2);
    __shared__ float blk[16][4];
    const int per = 256 / G;                                                      // matches per workgroup: 4 or 16
    const int n = *s.count;
    if ((int)blockIdx.x * per >= n) return;                                          // (the grid covers the CAPACITY; finalize reads ceil(n / 4) rows)
    const int sub = threadIdx.x / G, lane = threadIdx.x - sub * G;
    const int e = blockIdx.x * per + sub;
    const int attrs = p.nc + (p.mode == 0 ? 185 : 6);
    float reg_a = 0.f, reg_b = 0.f, clsl = 0.f, thl = 0.f;
    if (e < n) {
        const int* r = s.rec + (int64_t)e * 8;
        const float* f = s.frec + (int64_t)e * 8;
        const int cell = r[6];
        const float* ps = p.head[scale] + (int64_t)cell * attrs;
        if (lane == 0) { reg_a = f[6]; reg_b = f[7]; }
        // class BCE (nc > 1 only, lib/loss.py:223 / :399)
        const int c0 = p.mode == 0 ? 5 : 6;
        if (p.nc > 1) {
            const int tc = r[4];
            for (int k = lane; k < p.nc; k += G) {
                const float x = ps[c0 + k], t = (k == tc) ? 1.f : 0.f;
                clsl += fl_val(x, t, p.cls_pw, p.fl_gamma, p.fl_alpha);
            }
        }
        if (p.mode == 0) {                                                       // CSL theta BCE, lib/loss.py:231  (G = 64)
            const float* tg = p.targets + (int64_t)r[5] * p.tcols + 7;
            for (int k = lane; k < 180; k += G) {
                const float x = ps[5 + p.nc + k], t = tg[k];
                thl += fl_val(x, t, 1.0f, p.fl_gamma, p.fl_alpha);
            }
        }
    }
    for (int o = G >> 1; o > 0; o >>= 1) { clsl += __shfl_xor(clsl, o, 64); thl += __shfl_xor(thl, o, 64); }
    if (lane == 0) { blk[sub][0] = reg_a; blk[sub][1] = reg_b; blk[sub][2] = clsl; blk[sub][3] = thl; }
    __syncthreads();
    if ((int)threadIdx.x < per) {                                                     // thread (row j, item q)
        const int j = threadIdx.x >> 2, q = threadIdx.x & 3;
        const float v = blk[4 * j][q] + blk[4 * j + 1][q] + blk[4 * j + 2][q] + blk[4 * j + 3][q];
        const int row = blockIdx.x * (per >> 2) + j;
        if (row < s.nblk_match) s.part_match[(int64_t)row * 4 + q] = v;
    }
}

// ------------------------------------------------------------------------------------------------ K2b per-cell gradient
// d(loss)/d(logits) of the matched cells.  A cell matched by several targets (the reference gathers pi[b, a, gj, gi] with repeated
// indices, autograd scatters with index_put_(accumulate=True)) gets the SUM of its matches' terms: the cell's owner (largest match
// index) walks the chain K2 linked and adds the members in ASCENDING match order — a fixed order, so the gradient is bitwise
// reproducible (float atomics in arrival order differed in the last bits whenever three or more targets shared a cell) — and writes
// plain stores.  Chains are short (1 for almost every cell); the next member is found by selection (O(d^2) walks, no storage bound).
__global__ __launch_bounds__(256) void loss_match_grad_kernel(const LossParams p, ScaleWs s0, ScaleWs s1, ScaleWs s2)
{
    const int scale = blockIdx.y;
    const ScaleWs s = scale == 0 ? s0 : (scale == 1 ? s1 : s2);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = blockIdx.x * 4 + wave;
    const int n = *s.count;
    if (e >= n) return;
    const int cell = s.rec[(int64_t)e * 8 + 6];
    if (s.owner[cell] != e) return;                                              // one wave per matched cell
    const int attrs = p.nc + (p.mode == 0 ? 185 : 6);
    const float* ps = p.head[scale] + (int64_t)cell * attrs;
    float* gp = p.grad[scale] + (int64_t)cell * attrs;
    const float inv_n = 1.0f / (float)n;
    const int c0 = p.mode == 0 ? 5 : 6;
    const float kc = p.cls * inv_n / (float)(p.nc > 0 ? p.nc : 1);
    const float kt = p.theta_gain * inv_n / 180.f;
    // (r05: the gradient map is no longer cleared beforehand — this kernel defines every element of an owned row except the objectness one,
    // loss_obj_kernel the rest of the map)
    if (p.nc == 1 && lane == 0) gp[c0] = 0.f;                                    // a single class carries no class term (lib/loss.py:223,399)
    if (p.nc > 64)
        for (int k = lane; k < p.nc; k += 64) gp[c0 + k] = 0.f;                  // wide class heads accumulate in memory below (same lane, program order)
    float gbox[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    float gcls = 0.f, gth[3] = {0.f, 0.f, 0.f};
    const float xc = (p.nc > 1 && lane < p.nc) ? ps[c0 + lane] : 0.f;             // nc <= 64 classes per lane pass (looped below if more)
    float xt[3] = {0.f, 0.f, 0.f};
    if (p.mode == 0)
        for (int q = 0; q < 3; q++) { const int k = lane + 64 * q; if (k < 180) xt[q] = ps[5 + p.nc + k]; }
    int prev = -1;
    for (;;) {
        int m = 0x7fffffff;
        if (lane == 0)
            for (int c = s.head[cell]; c >= 0; c = s.next[c])
                if (c > prev && c < m) m = c;
        m = __shfl(m, 0, 64);
        if (m == 0x7fffffff) break;
        prev = m;
        const int* r = s.rec + (int64_t)m * 8;
        if (lane == 0) {
            const float* gb = s.gbox + (int64_t)m * 8;
            for (int k = 0; k < 5; k++) gbox[k] += gb[k];
        }
        if (p.nc > 1) {
            const int tc = r[4];
            if (p.nc <= 64) {
                if (lane < p.nc) gcls += kc * fl_grad(xc, lane == tc ? 1.f : 0.f, p.cls_pw, p.fl_gamma, p.fl_alpha);
            } else {
                for (int k = lane; k < p.nc; k += 64)                                  // wide class heads: accumulate in memory, same order
                    gp[c0 + k] += kc * fl_grad(ps[c0 + k], k == tc ? 1.f : 0.f, p.cls_pw, p.fl_gamma, p.fl_alpha);
            }
        }
        if (p.mode == 0) {
            const float* tg = p.targets + (int64_t)r[5] * p.tcols + 7;
            for (int q = 0; q < 3; q++) { const int k = lane + 64 * q; if (k < 180) gth[q] += kt * fl_grad(xt[q], tg[k], 1.0f, p.fl_gamma, p.fl_alpha); }
        }
    }
    if (lane == 0) {
        for (int k = 0; k < 4; k++) gp[k] = gbox[k];
        if (p.mode != 0) gp[4] = gbox[4];
    }
    if (p.nc > 1 && p.nc <= 64 && lane < p.nc) gp[c0 + lane] = gcls;
    if (p.mode == 0)
        for (int q = 0; q < 3; q++) { const int k = lane + 64 * q; if (k < 180) gp[5 + p.nc + k] = gth[q]; }
}

// ------------------------------------------------------------------------------------------------ K3 / K4
__global__ __launch_bounds__(256) void loss_obj_kernel(const LossParams p, ScaleWs s0, ScaleWs s1, ScaleWs s2)
{
    __shared__ float red[4];
    const int scale = blockIdx.y;
    const ScaleWs s = scale == 0 ? s0 : (scale == 1 ? s1 : s2);
    if ((int)blockIdx.x >= s.nblk_obj) return;                                       // (one launch for the three scales: the grid is the largest one's)
    const int attrs = p.nc + (p.mode == 0 ? 185 : 6);
    const int och = p.mode == 0 ? 4 : 5;
    const float kg = p.obj / (float)s.cells;
    const float rattrs = 1.0f / (float)attrs;
    const int lane = threadIdx.x & 63;
    float acc = 0.f;
    // r05: this kernel also DEFINES the gradient map (it used to be cleared by a 1.3 GB hipMemsetAsync at the benchmark size, long before this pass
    // scattered one float per 88-byte row into it).  A wave owns 64 consecutive cells = one contiguous run of 64 * attrs floats: it writes the whole
    // run — zeros with the objectness gradient in place — as 16-byte stores; rows of MATCHED cells (owner >= 0) keep what loss_match_grad_kernel
    // wrote before this launch, so a run that holds one takes the element-wise path.  Same cell -> thread mapping and summation order as before:
    // loss values are bit-identical.
    for (int ib = blockIdx.x * 256; ib < s.cells; ib += s.nblk_obj * 256) {         // (uniform trip count: the shuffles below need every lane)
        const int i = ib + threadIdx.x;
        const bool valid = i < s.cells;
        float g = 0.f;
        int own = -1;
        if (valid) {
            const float x = p.headobj[scale] ? p.headobj[scale][i] : p.head[scale][(int64_t)i * attrs + och];
            own = s.owner[i];
            const float t = own >= 0 ? s.frec[(int64_t)own * 8 + 5] : 0.f;           // gr = 1.0: tconf = the owner's score (lib/loss.py:221), else 0
            acc += fl_val(x, t, p.obj_pw, p.fl_gamma, p.fl_alpha);
            if (p.compute_grad) {
                g = kg * fl_grad(x, t, p.obj_pw, p.fl_gamma, p.fl_alpha);
                if (p.objgrad[scale]) p.objgrad[scale][i] = g;
            }
        }
        if (!p.compute_grad) continue;
        const int w0 = ib + (int)(threadIdx.x & ~63u);                              // first cell of this wave's run
        if (w0 >= s.cells) continue;                                                // (wave-uniform)
        const int ncell = min(64, s.cells - w0);
        const bool plain = __ballot(own >= 0) == 0ull && ncell == 64;
        float* const base = p.grad[scale] + (int64_t)w0 * attrs;
        const int nflo = ncell * attrs;
        if (plain) {
            const int trips = (nflo + 255) >> 8;                                      // uniform: every lane takes part in every shuffle
            for (int k = 0; k < trips; k++) {                                        // nflo = 64 * attrs: a multiple of 4, base 16-byte aligned
                const int f0 = 256 * k + 4 * lane;
                int c = (int)((float)f0 * rattrs);
                if (c * attrs > f0) c--;
                if ((c + 1) * attrs <= f0) c++;
                const int e1 = och - (f0 - c * attrs);                              // position of cell c's objectness element inside this quad
                const float gv = __shfl(g, c < 64 ? c : 63, 64);                    // (attrs >= 7 > 4: a quad holds at most one objectness element,
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);                         //  and never the NEXT cell's: that one sits >= och + 1 >= 5 further)
                if (e1 == 0) v.x = gv; else if (e1 == 1) v.y = gv; else if (e1 == 2) v.z = gv; else if (e1 == 3) v.w = gv;
                if (f0 < nflo) *reinterpret_cast<float4*>(base + f0) = v;
            }
        } else {
            const int trips = (nflo + 63) >> 6;
            for (int k = 0; k < trips; k++) {
                const int f = 64 * k + lane;
                int c = (int)((float)f * rattrs);
                if (c * attrs > f) c--;
                if ((c + 1) * attrs <= f) c++;
                const int cs = c < 64 ? c : 63;
                const float gv = __shfl(g, cs, 64);
                const int oc = __shfl(own, cs, 64);
                if (f < nflo) {
                    if (f - c * attrs == och) base[f] = gv;
                    else if (oc < 0) base[f] = 0.f;
                }
            }
        }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) s.part_obj[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// ------------------------------------------------------------------------------------------------ K5 finalize
// (r05: 1024 threads, 16-byte loads of the partial rows and shuffle reductions — the 256-thread form walked ~30 dependent global loads per thread
// and 135 barriers: 72 us at the benchmark size for a few kilobytes of sums)
__global__ __launch_bounds__(1024) void loss_finalize_kernel(const LossParams p, ScaleWs s0, ScaleWs s1, ScaleWs s2)
{
    __shared__ double red[16][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double part[3][5];                            // per scale: reg_a, reg_b, cls, theta (per-match sums), objectness
    int nn[3];
    for (int i = 0; i < 3; i++) {
        const ScaleWs s = i == 0 ? s0 : (i == 1 ? s1 : s2);
        const int n = *s.count;
        nn[i] = n;
        const int nb = (n + 3) / 4;
        for (int q = 0; q < 5; q++) part[i][q] = 0.0;
        for (int b = threadIdx.x; b < nb; b += 1024) {
            const float4 v = *reinterpret_cast<const float4*>(s.part_match + (int64_t)b * 4);
            part[i][0] += (double)v.x; part[i][1] += (double)v.y; part[i][2] += (double)v.z; part[i][3] += (double)v.w;
        }
        for (int b = threadIdx.x; b < s.nblk_obj; b += 1024) part[i][4] += (double)s.part_obj[b];
    }
    int bad = 0;
    for (int t = threadIdx.x; t < p.nt; t += 1024) {
        const int tb = (int)p.targets[(int64_t)t * p.tcols];
        bad += (tb < 0 || tb >= p.batch) ? 1 : 0;
    }
    for (int i = 0; i < 3; i++)
        for (int q = 0; q < 5; q++) {
            const double v = wave_sum_d(part[i][q]);
            if (lane == 0) red[wave][i * 5 + q] = v;
        }
    {
        const double v = wave_sum_d((double)bad);
        if (lane == 0) red[wave][15] = v;
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        double v = 0.0;
        for (int w = 0; w < 16; w++) v += red[w][threadIdx.x];
        red[0][threadIdx.x] = v;                  // (thread q only touches column q)
    }
    __syncthreads();
    double tot[4] = {0.0, 0.0, 0.0, 0.0};       // reg, conf, cls, theta (unscaled, summed over scales)
    for (int i = 0; i < 3; i++) {
        const ScaleWs s = i == 0 ? s0 : (i == 1 ? s1 : s2);
        const int n = nn[i];
        if (n > 0) {
            tot[0] += (red[0][i * 5 + 0] + red[0][i * 5 + 1]) / (double)n;
            if (p.nc > 1) tot[2] += red[0][i * 5 + 2] / ((double)n * p.nc);
            if (p.mode == 0) tot[3] += red[0][i * 5 + 3] / ((double)n * 180.0);
        }
        tot[1] += red[0][i * 5 + 4] / (double)s.cells;
    }
    if (threadIdx.x == 0) {
        const float reg = p.box * (float)tot[0], conf = p.obj * (float)tot[1], cls = p.cls * (float)tot[2],
                    th = p.theta_gain * (float)tot[3];
        p.items[0] = reg; p.items[1] = conf; p.items[2] = cls; p.items[3] = th;
        p.items[4] = reg + conf + cls + th;
        p.items[5] = (float)red[0][15];
    }
}

// ------------------------------------------------------------------------------------------------ C ABI
extern "C" int ryolo_loss_workspace_bytes(const LossParams* pp, size_t* bytes)
{
    if (!pp || !bytes) return RY_ERR_ARG;
    LossParams p = *pp;
    p.ws = nullptr;
    ScaleWs s[3];
    carve(p, s, bytes);
    return RY_OK;
}

extern "C" int ryolo_loss(const LossParams* pp, hipStream_t stream)
{
    if (!pp) return RY_ERR_ARG;
    const LossParams& p = *pp;
    if (p.na < 1 || p.na > LOSS_MAX_NA || p.nt < 0 || p.batch < 1 || p.nc < 0 || !p.items || !p.ws) return RY_ERR_ARG;
    if (p.nt > 0 && !p.targets) return RY_ERR_ARG;
    if (p.tcols < (p.mode == 0 ? 187 : 7) && p.nt > 0) return RY_ERR_ARG;
    ScaleWs s[3];
    size_t need, ff_off, ff_bytes;
    carve(p, s, &need, &ff_off, &ff_bytes);
    if (p.ws_bytes < need) return RY_ERR_WORKSPACE;
    for (int i = 0; i < 3; i++)
        if (!p.head[i] || (p.compute_grad && !p.grad[i])) return RY_ERR_ARG;
    // two fills: the match counters (zero) and the owner / chain-head grids (-1); everything else is written before it is read, and the
    // gradient maps are defined by loss_match_grad_kernel + loss_obj_kernel
    if (hipMemsetAsync(p.ws, 0, 768, stream) != hipSuccess) return RY_ERR_LAUNCH;
    if (hipMemsetAsync(reinterpret_cast<char*>(p.ws) + ff_off, 0xff, p.compute_grad ? ff_bytes : ff_bytes / 2, stream) != hipSuccess) return RY_ERR_LAUNCH;
    int max_obj = 1;
    for (int i = 0; i < 3; i++) max_obj = s[i].nblk_obj > max_obj ? s[i].nblk_obj : max_obj;
    if (p.nt > 0) {                                           // (one launch per pass for the three scales: blockIdx.y; cap is the same for all)
        hipLaunchKernelGGL(loss_targets_count_kernel, dim3(3, LT_BLOCKS), dim3(1024), 0, stream, p, s[0], s[1], s[2]);
        hipLaunchKernelGGL(loss_targets_kernel, dim3(3, LT_BLOCKS), dim3(1024), 0, stream, p, s[0], s[1], s[2]);
        hipLaunchKernelGGL(loss_match_box_kernel, dim3((unsigned)ry_cdiv(s[0].cap, 256), 3), dim3(256), 0, stream, p, s[0], s[1], s[2]);
        const int G = (p.mode != 0 && p.nc <= 16) ? 16 : 64;
        hipLaunchKernelGGL(loss_match_kernel, dim3((unsigned)ry_cdiv(s[0].cap, 256 / G), 3), dim3(256), 0, stream, p, s[0], s[1], s[2], G);
        if (p.compute_grad) hipLaunchKernelGGL(loss_match_grad_kernel, dim3(s[0].nblk_match, 3), dim3(256), 0, stream, p, s[0], s[1], s[2]);
    }
    hipLaunchKernelGGL(loss_obj_kernel, dim3(max_obj, 3), dim3(256), 0, stream, p, s[0], s[1], s[2]);
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(1024), 0, stream, p, s[0], s[1], s[2]);
    RY_CHECK_LAUNCH();
    return RY_OK;
}

// the owner grids of the last ryolo_loss call on this workspace: owner[i][cell] >= 0 iff the cell of scale i was matched (its grad[] row is dense);
// valid until the next ryolo_loss on the same workspace
extern "C" int ryolo_loss_owner_grids(const LossParams* pp, const int** owner)
{
    if (!pp || !owner || !pp->ws) return RY_ERR_ARG;
    LossParams p = *pp;
    ScaleWs s[3];
    size_t need;
    carve(p, s, &need);
    if (p.ws_bytes < need) return RY_ERR_WORKSPACE;
    for (int i = 0; i < 3; i++) owner[i] = s[i].owner;
    return RY_OK;
}

// where the match counters and records of the last ryolo_loss call on this workspace live: count[i] -> one int, rec[i] -> [count][8] ints
// (b, a, gj, gi, cls, target row, cell, pad) in the reference's enumeration order (tests: target-assignment parity)
extern "C" int ryolo_loss_match_records(const LossParams* pp, const int** count, const int** rec)
{
    if (!pp || !count || !rec || !pp->ws) return RY_ERR_ARG;
    LossParams p = *pp;
    ScaleWs s[3];
    size_t need;
    carve(p, s, &need);
    if (p.ws_bytes < need) return RY_ERR_WORKSPACE;
    for (int i = 0; i < 3; i++) { count[i] = s[i].count; rec[i] = s[i].rec; }
    return RY_OK;
}

// grad *= *scale unless *scale == 1 (uniform early exit: one scalar load per workgroup)
__global__ __launch_bounds__(256) void loss_grad_scale_kernel(float* __restrict__ g, int64_t n, const float* __restrict__ scale)
{
    const float s = *scale;
    if (s == 1.0f) return;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) g[i] *= s;
}

// the same for up to 8 arrays in ONE launch (the three gradient maps and their compact objectness copies): blockIdx.y = array
struct GradScaleSet { float* g[8]; int64_t n[8]; };
__global__ __launch_bounds__(256) void loss_grad_scale_multi_kernel(const GradScaleSet a, const float* __restrict__ scale)
{
    const float s = *scale;
    if (s == 1.0f) return;
    float* const g = a.g[blockIdx.y];
    const int64_t n = a.n[blockIdx.y], stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) g[i] *= s;
}

extern "C" int ryolo_loss_grad_scale_multi(float* const* grads, const int64_t* n, int count, const float* scale, hipStream_t stream)
{
    if (!scale || !grads || !n || count < 1 || count > 8) return RY_ERR_ARG;
    GradScaleSet a;
    int64_t most = 0;
    for (int i = 0; i < 8; i++) {
        a.g[i] = i < count ? grads[i] : nullptr;
        a.n[i] = i < count ? n[i] : 0;
        if (i < count && (n[i] < 0 || (n[i] > 0 && !grads[i]))) return RY_ERR_ARG;
        most = a.n[i] > most ? a.n[i] : most;
    }
    if (most == 0) return RY_OK;
    const int64_t blocks = (most + 255) / 256;
    hipLaunchKernelGGL(loss_grad_scale_multi_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096), count), dim3(256), 0, stream, a, scale);
    RY_CHECK_LAUNCH();
    return RY_OK;
}

extern "C" int ryolo_loss_grad_scale(float* grad, int64_t n, const float* scale, hipStream_t stream)
{
    if (!scale || n < 0 || (n > 0 && !grad)) return RY_ERR_ARG;
    if (n == 0) return RY_OK;
    const int64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(loss_grad_scale_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, stream, grad, n, scale);
    RY_CHECK_LAUNCH();
    return RY_OK;
}
