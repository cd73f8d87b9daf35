// Image-side augmentations of the reference's loader as device kernels over uint8 HWC (BGR) images resident in HBM — SURVEY.md
// §8(f) N2: mosaic-4 / mosaic-9 assembly (datasets/base_dataset.py:224-330), the perspective warp of random_warping
// (lib/augmentations.py:45-74), hsv (:8-21) and mixup (:24-28).  At ~700 img/s per GPU the reference's 8 cv2 worker processes
// (lib/load.py:19) cannot feed one MI355X; here a batch is byte movement at HBM speed:
//   paste_rects   canvas filled with 114, then every source rectangle copied in the reference's paste order (later rectangles win,
//                 as img9[y1:y2, x1:x2] = ... overwrites): one launch per batch of canvases, 4 pixels (12 bytes) per thread;
//   warp          inverse perspective map per destination pixel + bilinear taps with OpenCV's fixed-point layout (5 fractional
//                 coordinate bits, 15-bit weights, round-to-nearest), border value 114;
//   hsv_gain      BGR -> HSV (OpenCV's 8-bit integer path: 12-bit division tables), three LUTs, HSV -> BGR (float path), in place;
//   mixup         uint8(a * r + b * (1 - r)) in double, truncation like numpy's astype(uint8).
// PARITY: paste and mixup are pure index / IEEE arithmetic and are pinned to the imported reference (fixture g11, the real
// load_mosaic / load_mosaic9 / mixup ran); warp and hsv restate OpenCV (third-party, absent here, version un-pinned by the reference):
// "parity unpinned" — checked only against this build's own numpy restatement (oracle/ref_data.py).
#include "common.h"

struct PasteRect {               // one `canvas[dy:dy+h, dx:dx+w] = src[sy:sy+h, sx:sx+w]`
    int64_t src_off;             // byte offset of the source image in the pool
    int src_w;                   // source row pitch in pixels
    int sx, sy, dx, dy, w, h;
    int canvas;                  // index of the destination canvas in the batch
};

// `first` (optional, [ncanvas + 1]): the rectangles of canvas b are rects[first[b] .. first[b + 1]) — the batch assembler emits them canvas by
// canvas.  Without it every thread had to walk the whole table: ~370 rectangles x 190 M canvas pixels per batch of 64 mosaics = 20.6 ms, two
// thirds of the assembler's GPU time and — because concurrent kernels stretch each other on this part — a quarter of a loader-fed training step.
// A thread owns 4 consecutive canvas pixels (12 bytes = three dword stores; CW is a multiple of 4 for every canvas the loader builds, other widths
// take the byte path at the row end).
__global__ __launch_bounds__(256) void paste_rects_kernel(const uint8_t* __restrict__ pool, const PasteRect* __restrict__ rects, int nrect,
                                                          const int* __restrict__ first, uint8_t* __restrict__ canvas, int CH, int CW, int fill)
{
    const int b = blockIdx.z;
    const int y = blockIdx.y;
    const int x0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (x0 >= CW) return;
    const int r_lo = first ? first[b] : 0, r_hi = first ? first[b + 1] : nrect;
    uint8_t px[12];
#pragma unroll
    for (int k = 0; k < 12; k++) px[k] = (uint8_t)fill;
    unsigned todo = 0xfu;                                           // pixels of this thread not yet covered
    for (int r = r_hi - 1; r >= r_lo && todo; r--) {                // the LAST paste that covers a pixel wins
        const PasteRect q = rects[r];
        const int ry = y - q.dy;
        if (q.canvas != b || (unsigned)ry >= (unsigned)q.h) continue;
        const uint8_t* srow = pool + q.src_off + ((int64_t)(q.sy + ry) * q.src_w + q.sx) * 3;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int rx = x0 + k - q.dx;
            if (((todo >> k) & 1u) && (unsigned)rx < (unsigned)q.w) {
                const uint8_t* sp = srow + rx * 3;
                px[3 * k] = sp[0]; px[3 * k + 1] = sp[1]; px[3 * k + 2] = sp[2];
                todo &= ~(1u << k);
            }
        }
    }
    uint8_t* d = canvas + (((int64_t)b * CH + y) * CW + x0) * 3;
    if (x0 + 4 <= CW && (((uintptr_t)d) & 3) == 0) {
        unsigned w[3];
#pragma unroll
        for (int k = 0; k < 3; k++) w[k] = (unsigned)px[4 * k] | ((unsigned)px[4 * k + 1] << 8) | ((unsigned)px[4 * k + 2] << 16) | ((unsigned)px[4 * k + 3] << 24);
        unsigned* dw = reinterpret_cast<unsigned*>(d);
        dw[0] = w[0]; dw[1] = w[1]; dw[2] = w[2];
    } else {
        for (int k = 0; k < 12 && x0 + k / 3 < CW; k++) d[k] = px[k];
    }
}

// Two horizontally adjacent BGR pixels (6 bytes) as ONE unaligned 8-byte load (hipcc emits a single global_load_dwordx2 for it on gfx950); the
// caller guarantees 8 readable bytes (pixel index + 3 <= pixels of the image).  Per-channel byte loads made the bilinear kernels request-bound:
// 12 loads per output pixel for 4 taps x 3 channels, now 2.
__device__ __forceinline__ unsigned long long load_px2(const uint8_t* p)
{
    unsigned long long v;
    __builtin_memcpy(&v, p, 8);
    return v;
}
__device__ __forceinline__ int px2_byte(unsigned long long v, int k) { return (int)((v >> (8 * k)) & 255ull); }

// dst[b] = warpPerspective(src[b], M[b]) with flags INTER_LINEAR, borderMode CONSTANT, borderValue (114, 114, 114).
// Minv [B][9] double: the INVERSE of the matrix the caller passed to cv2.warpPerspective (OpenCV inverts it itself).
__global__ __launch_bounds__(256) void warp_perspective_kernel(const uint8_t* __restrict__ src, int SH, int SW, const double* __restrict__ Minv,
                                                               uint8_t* __restrict__ dst, int DH, int DW, int border)
{
    const int b = blockIdx.z, y = blockIdx.y;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= DW) return;
    const double* m = Minv + (int64_t)b * 9;
    // cv::warpPerspective: W = M20 x + M21 y + M22; fX = (M00 x + M01 y + M02) / W scaled by INTER_TAB_SIZE = 32 and rounded
    // (saturate_cast<int> of a double = cvRound: round half to even)
    double W = m[6] * x + m[7] * y + m[8];
    W = W ? 32.0 / W : 0.0;
    const double fX = fmax((double)INT_MIN, fmin((double)INT_MAX, (m[0] * x + m[1] * y + m[2]) * W));
    const double fY = fmax((double)INT_MIN, fmin((double)INT_MAX, (m[3] * x + m[4] * y + m[5]) * W));
    const int X = (int)rint(fX), Y = (int)rint(fY);
    const int sx = X >> 5, sy = Y >> 5, ax = X & 31, ay = Y & 31;
    // bilinear weights: OpenCV's table holds (1 - a/32, a/32) products scaled by 2^15 and rounded, corrected so the four sum to 2^15
    const float fx1 = ax * (1.f / 32.f), fy1 = ay * (1.f / 32.f);
    const float wf[4] = {(1.f - fy1) * (1.f - fx1), (1.f - fy1) * fx1, fy1 * (1.f - fx1), fy1 * fx1};
    int w[4], sum = 0, imax = 0, imin = 0;
    for (int k = 0; k < 4; k++) {
        w[k] = (int)rintf(wf[k] * 32768.f);
        sum += w[k];
        if (w[k] > w[imax]) imax = k;
        if (w[k] < w[imin]) imin = k;
    }
    // (initInterTab2D: the rounding error of the sum is pushed into the largest / smallest weight)
    if (sum != 32768) {
        const int diff = sum - 32768;
        if (diff < 0) w[imax] -= diff; else w[imin] -= diff;
    }
    const uint8_t* sb = src + (int64_t)b * SH * SW * 3;
    uint8_t* d = dst + (((int64_t)b * DH + y) * DW + x) * 3;
    if (sx >= 0 && sx + 1 < SW && sy >= 0 && sy + 1 < SH && (int64_t)(sy + 1) * SW + sx + 3 <= (int64_t)SH * SW) {
        // all four taps inside the image (and 8 readable bytes behind the lower pair): the same integer sums from two 8-byte loads
        const unsigned long long r0 = load_px2(sb + ((int64_t)sy * SW + sx) * 3), r1 = load_px2(sb + ((int64_t)(sy + 1) * SW + sx) * 3);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int acc = px2_byte(r0, c) * w[0] + px2_byte(r0, c + 3) * w[1] + px2_byte(r1, c) * w[2] + px2_byte(r1, c + 3) * w[3];
            d[c] = (uint8_t)((acc + (1 << 14)) >> 15);
        }
        return;
    }
    for (int c = 0; c < 3; c++) {
        int acc = 0;
        for (int k = 0; k < 4; k++) {
            const int px = sx + (k & 1), py = sy + (k >> 1);
            const int v = ((unsigned)px < (unsigned)SW && (unsigned)py < (unsigned)SH) ? sb[((int64_t)py * SW + px) * 3 + c] : border;
            acc += v * w[k];
        }
        d[c] = (uint8_t)((acc + (1 << 14)) >> 15);
    }
}

// cv2.cvtColor(BGR2HSV) on uint8 -> LUTs -> cv2.cvtColor(HSV2BGR), in place (lib/augmentations.py:8-21).  lut [3][256] uint8.
__global__ __launch_bounds__(256) void hsv_gain_kernel(uint8_t* __restrict__ img, int64_t npix, const uint8_t* __restrict__ lut)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    uint8_t* p = img + i * 3;
    const int b = p[0], g = p[1], r = p[2];
    // RGB2HSV_b (OpenCV, hrange 180): 12-bit fixed-point division tables
    int v = max(b, max(g, r)), vmin = min(b, min(g, r));
    const int diff = v - vmin;
    const int vr = v == r ? -1 : 0, vg = v == g ? -1 : 0;
    const int sdiv = v ? (int)rint((255 << 12) / (double)v) : 0;
    const int hdiv = diff ? (int)rint((180 << 12) / (6.0 * diff)) : 0;
    const int s = (diff * sdiv + (1 << 11)) >> 12;
    int h = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
    h = (h * hdiv + (1 << 11)) >> 12;
    h += h < 0 ? 180 : 0;
    const int H = lut[h & 255], S = lut[256 + s], V = lut[512 + v];
    // HSV2RGB_b: 8-bit values through the float converter (h in degrees / 2, s and v scaled by 1/255), result * 255 rounded
    float hf = (float)H * (6.f / 180.f), sf = (float)S * (1.f / 255.f), vf = (float)V * (1.f / 255.f);
    float bb, gg, rr;
    if (sf == 0.f) {
        bb = gg = rr = vf;
    } else {
        static const int sector[6][3] = {{1, 3, 0}, {1, 0, 2}, {3, 0, 1}, {0, 2, 1}, {0, 1, 3}, {2, 1, 0}};
        if (hf < 0.f) { do hf += 6.f; while (hf < 0.f); }
        else if (hf >= 6.f) { do hf -= 6.f; while (hf >= 6.f); }
        const int sec = (int)floorf(hf);
        hf -= (float)sec;
        const int sc = (unsigned)sec >= 6u ? 0 : sec;
        if ((unsigned)sec >= 6u) hf = 0.f;
        float tab[4];
        tab[0] = vf;
        tab[1] = vf * (1.f - sf);
        tab[2] = vf * (1.f - sf * hf);
        tab[3] = vf * (1.f - sf * (1.f - hf));
        bb = tab[sector[sc][0]];
        gg = tab[sector[sc][1]];
        rr = tab[sector[sc][2]];
    }
    p[0] = (uint8_t)min(255, max(0, (int)rintf(bb * 255.f)));
    p[1] = (uint8_t)min(255, max(0, (int)rintf(gg * 255.f)));
    p[2] = (uint8_t)min(255, max(0, (int)rintf(rr * 255.f)));
}

// out = (a * r + b * (1 - r)).astype(uint8): double arithmetic, truncation toward zero (lib/augmentations.py:24-28)
__global__ __launch_bounds__(256) void mixup_kernel(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, double r, int64_t n,
                                                    uint8_t* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = (double)a[i] * r + (double)b[i] * (1.0 - r);
    out[i] = (uint8_t)(int)v;
}

// cv::resize's whole-number test (`is_area_fast`, imgproc/src/resize.cpp): scale = 1 / ((double) dst / src) per axis, iscale = saturate_cast<int>
// (round half even); both |scale - iscale| < DBL_EPSILON.  INTER_AREA downscales by whole numbers take the integer block sum, and INTER_LINEAR
// at exactly 2 x 2 is switched to that path by OpenCV itself.  ix = iy = 0: generic path.
__device__ __forceinline__ void area_fast_scales(int SH, int SW, int NH, int NW, int& ix, int& iy)
{
    const double sx = 1.0 / ((double)NW / (double)SW), sy = 1.0 / ((double)NH / (double)SH);
    const int rx = (int)rint(sx), ry = (int)rint(sy);
    const bool fast = fabs(sx - rx) < 2.220446049250313e-16 && fabs(sy - ry) < 2.220446049250313e-16 && rx >= 1 && ry >= 1 && (rx > 1 || ry > 1);
    ix = fast ? rx : 0;
    iy = fast ? ry : 0;
}
// cv::resizeAreaFast_ for uint8: the iy x ix block summed in int; 2 x 2 -> (sum + 2) >> 2 (the vector form: round half up), otherwise
// saturate_cast<uchar>(sum * (1.f / area)) (float product, round half even)
__device__ __forceinline__ void area_fast_pixel(const uint8_t* __restrict__ src, int SW, int x, int y, int ix, int iy, int* px)
{
    int sum[3] = {0, 0, 0};
    for (int r = 0; r < iy; r++) {
        const uint8_t* rp = src + ((int64_t)(y * iy + r) * SW + (int64_t)x * ix) * 3;
        for (int q = 0; q < ix; q++)
            for (int c = 0; c < 3; c++) sum[c] += rp[q * 3 + c];
    }
    if (ix == 2 && iy == 2) { for (int c = 0; c < 3; c++) px[c] = (sum[c] + 2) >> 2; }
    else {
        const float scale = 1.f / (float)(ix * iy);
        for (int c = 0; c < 3; c++) px[c] = min(255, max(0, (int)rintf((float)sum[c] * scale)));
    }
}

// Letterbox of pad_to_square (datasets/base_dataset.py:33-56): cv2.resize(img, (NW, NH), INTER_LINEAR) placed at (top, left) of an
// OH x OW canvas filled with `fill` (cv2.copyMakeBorder, BORDER_CONSTANT).  OpenCV's 8-bit linear resize: source coordinate
// (x + 0.5) * scale - 0.5, 11-bit coefficients (cvRound), horizontal pass in int, vertical pass ((b0 * (S0 >> 4)) >> 16) + ... + 2 >> 2.
__global__ __launch_bounds__(256) void letterbox_kernel(const uint8_t* __restrict__ src, int SH, int SW, int NH, int NW, int top, int left,
                                                        uint8_t* __restrict__ dst, int OH, int OW, int fill)
{
    const int y = blockIdx.y, x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= OW) return;
    uint8_t* d = dst + ((int64_t)y * OW + x) * 3;
    const int rx = x - left, ry = y - top;
    if ((unsigned)rx >= (unsigned)NW || (unsigned)ry >= (unsigned)NH) { d[0] = d[1] = d[2] = (uint8_t)fill; return; }
    if (NH == SH && NW == SW) {                                     // `if shape[::-1] != new_unpad` — no resize
        const uint8_t* sp = src + ((int64_t)ry * SW + rx) * 3;
        d[0] = sp[0]; d[1] = sp[1]; d[2] = sp[2];
        return;
    }
    int fx, fy;
    area_fast_scales(SH, SW, NH, NW, fx, fy);
    if (fx == 2 && fy == 2) {                                       // exactly half size: OpenCV's INTER_LINEAR is its INTER_AREA block mean here
        int px[3];
        area_fast_pixel(src, SW, rx, ry, 2, 2, px);
        d[0] = (uint8_t)px[0]; d[1] = (uint8_t)px[1]; d[2] = (uint8_t)px[2];
        return;
    }
    auto coef = [](int o, int dn, int sn, int& s0, int& a0, int& a1) {
        const double scale = (double)sn / (double)dn;
        float f = (float)((o + 0.5) * scale - 0.5);
        int si = (int)floorf(f);
        f -= (float)si;
        if (si < 0) { f = 0.f; si = 0; }
        if (si >= sn - 1) { f = 0.f; si = sn - 1; }
        s0 = si;
        a0 = (int)rintf((1.f - f) * 2048.f);
        a1 = (int)rintf(f * 2048.f);
    };
    int sx, ax0, ax1, sy, by0, by1;
    coef(rx, NW, SW, sx, ax0, ax1);
    coef(ry, NH, SH, sy, by0, by1);
    const int sx1 = min(sx + 1, SW - 1), sy1 = min(sy + 1, SH - 1);
    for (int c = 0; c < 3; c++) {
        const int r0 = src[((int64_t)sy * SW + sx) * 3 + c] * ax0 + src[((int64_t)sy * SW + sx1) * 3 + c] * ax1;
        const int r1 = src[((int64_t)sy1 * SW + sx) * 3 + c] * ax0 + src[((int64_t)sy1 * SW + sx1) * 3 + c] * ax1;
        d[c] = (uint8_t)((((by0 * (r0 >> 4)) >> 16) + ((by1 * (r1 >> 4)) >> 16) + 2) >> 2);
    }
}

// ---- batched source-image stage of load_image (datasets/base_dataset.py:170-186): cv2.resize to (NW, NH) [+ hsv gain in place] for EVERY
// source image a batch uses, one launch.  An item reads one image of the resident pool and writes its resized copy into a staging pool;
// interp 0 = INTER_LINEAR (the 8-bit two-pass integer form of letterbox_kernel above), 1 = INTER_AREA (whole-number scale factors: the integer
// block sum of cv::resizeAreaFast_, which INTER_LINEAR at exactly 2 x 2 also takes; otherwise the float accumulation form of
// cv::ResizeArea_: per axis a leading partial cell, whole cells of weight 1 / cellWidth, a trailing partial cell; weights and sums in
// float, cvRound at the end), 2 = copy (r == 1: the reference skips cv2.resize).  lut >= 0: the three 256-entry tables of hsv() for this
// image (lib/augmentations.py:8-21) are applied to the resized pixel before it is stored (resize, then hsv in place: same result).
struct ResizeItem {
    int64_t src_off, dst_off;    // byte offsets into the source pool / the staging pool
    int SH, SW, NH, NW;
    int interp, lut;
};

// divtab (optional, LDS): [0..255] = OpenCV's sdiv_table, [256..511] = hdiv_table180 — the two double-precision divisions per pixel as table
// reads (what cv::cvtColor's 8-bit BGR2HSV does itself); same integers as the expressions below
__device__ __forceinline__ void hsv_lut_pixel(int& b, int& g, int& r, const uint8_t* __restrict__ lut, const int* divtab = nullptr)
{
    int v = max(b, max(g, r)), vmin = min(b, min(g, r));
    const int diff = v - vmin;
    const int vr = v == r ? -1 : 0, vg = v == g ? -1 : 0;
    const int sdiv = divtab ? divtab[v] : (v ? (int)rint((255 << 12) / (double)v) : 0);
    const int hdiv = divtab ? divtab[256 + diff] : (diff ? (int)rint((180 << 12) / (6.0 * diff)) : 0);
    const int s = (diff * sdiv + (1 << 11)) >> 12;
    int h = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
    h = (h * hdiv + (1 << 11)) >> 12;
    h += h < 0 ? 180 : 0;
    const int H = lut[h & 255], S = lut[256 + s], V = lut[512 + v];
    float hf = (float)H * (6.f / 180.f), sf = (float)S * (1.f / 255.f), vf = (float)V * (1.f / 255.f);
    float bb, gg, rr;
    if (sf == 0.f) {
        bb = gg = rr = vf;
    } else {
        static const int sector[6][3] = {{1, 3, 0}, {1, 0, 2}, {3, 0, 1}, {0, 2, 1}, {0, 1, 3}, {2, 1, 0}};
        if (hf < 0.f) { do hf += 6.f; while (hf < 0.f); }
        else if (hf >= 6.f) { do hf -= 6.f; while (hf >= 6.f); }
        const int sec = (int)floorf(hf);
        hf -= (float)sec;
        const int sc = (unsigned)sec >= 6u ? 0 : sec;
        if ((unsigned)sec >= 6u) hf = 0.f;
        float tab[4];
        tab[0] = vf;
        tab[1] = vf * (1.f - sf);
        tab[2] = vf * (1.f - sf * hf);
        tab[3] = vf * (1.f - sf * (1.f - hf));
        bb = tab[sector[sc][0]];
        gg = tab[sector[sc][1]];
        rr = tab[sector[sc][2]];
    }
    b = min(255, max(0, (int)rintf(bb * 255.f)));
    g = min(255, max(0, (int)rintf(gg * 255.f)));
    r = min(255, max(0, (int)rintf(rr * 255.f)));
}

// one axis of cv::computeResizeAreaTab for destination index d: source cells [s_lo, s_hi] with weights (first, 1/cell ..., last)
struct AreaAxis { int lo, hi; float wlo, wmid, whi; bool has_lo, has_hi; };
__device__ __forceinline__ AreaAxis area_axis(int d, int dn, int sn)
{
    const double scale = (double)sn / (double)dn;
    const double f1 = d * scale, f2 = f1 + scale;
    const double cell = fmin(scale, (double)sn - f1);
    int s1 = (int)ceil(f1), s2 = (int)floor(f2);
    s2 = min(s2, sn - 1);
    s1 = min(s1, s2);
    AreaAxis a;
    a.has_lo = s1 - f1 > 1e-3;
    a.wlo = (float)((s1 - f1) / cell);
    a.lo = s1;                                     // whole cells s1 .. s2 - 1; the leading partial cell is s1 - 1
    a.hi = s2;
    a.wmid = (float)(1.0 / cell);
    a.has_hi = f2 - s2 > 1e-3;
    a.whi = (float)(fmin(fmin(f2 - s2, 1.0), cell) / cell);
    return a;
}

__global__ __launch_bounds__(256) void resize_hsv_batch_kernel(const uint8_t* __restrict__ pool, const ResizeItem* __restrict__ items,
                                                               const uint8_t* __restrict__ luts, uint8_t* __restrict__ stage)
{
    const ResizeItem it = items[blockIdx.y];
    __shared__ int divtab[512];
    if (it.lut >= 0) {                                              // (block-uniform)
        const int t = threadIdx.x;
        divtab[t] = t ? (int)rint((255 << 12) / (double)t) : 0;
        divtab[256 + t] = t ? (int)rint((180 << 12) / (6.0 * t)) : 0;
        __syncthreads();
    }
    const int64_t npix = (int64_t)it.NH * it.NW;
    __shared__ uint8_t slut[768];                                   // this image's three hsv tables: three scattered GLOBAL byte reads per pixel otherwise
    if (it.lut >= 0) {
        for (int t = threadIdx.x; t < 768; t += 256) slut[t] = luts[(int64_t)it.lut * 768 + t];
        __syncthreads();
    }
    int fx = 0, fy = 0;                                             // whole-number downscale: OpenCV's integer block path (block-uniform)
    if (it.interp != 2) {
        area_fast_scales(it.SH, it.SW, it.NH, it.NW, fx, fy);
        if (it.interp == 0 && !(fx == 2 && fy == 2)) fx = fy = 0;   // INTER_LINEAR only switches at exactly 2 x 2
    }
    // (pixel index in 32 bits: the host rejects images of 2^31 pixels; a 64-bit division per pixel was a third of this kernel's instructions)
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < (unsigned)npix; i += gridDim.x * 256u) {
        const int y = (int)(i / (unsigned)it.NW), x = (int)(i - (unsigned)y * (unsigned)it.NW);
        const uint8_t* src = pool + it.src_off;
        int px[3];
        if (it.interp == 2) {
            const uint8_t* sp = src + ((int64_t)y * it.SW + x) * 3;
            px[0] = sp[0]; px[1] = sp[1]; px[2] = sp[2];
        } else if (fx) {
            area_fast_pixel(src, it.SW, x, y, fx, fy, px);
        } else if (it.interp == 0) {
            auto coef = [](int o, int dn, int sn, int& s0, int& a0, int& a1) {
                const double scale = (double)sn / (double)dn;
                float f = (float)((o + 0.5) * scale - 0.5);
                int si = (int)floorf(f);
                f -= (float)si;
                if (si < 0) { f = 0.f; si = 0; }
                if (si >= sn - 1) { f = 0.f; si = sn - 1; }
                s0 = si;
                a0 = (int)rintf((1.f - f) * 2048.f);
                a1 = (int)rintf(f * 2048.f);
            };
            int sx, ax0, ax1, sy, by0, by1;
            coef(x, it.NW, it.SW, sx, ax0, ax1);
            coef(y, it.NH, it.SH, sy, by0, by1);
            const int sx1 = min(sx + 1, it.SW - 1), sy1 = min(sy + 1, it.SH - 1);
            if (sx1 == sx + 1 && (int64_t)sy1 * it.SW + sx + 3 <= (int64_t)it.SH * it.SW) {      // the pair of each row in one 8-byte load (load_px2)
                const unsigned long long q0 = load_px2(src + ((int64_t)sy * it.SW + sx) * 3), q1 = load_px2(src + ((int64_t)sy1 * it.SW + sx) * 3);
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const int r0 = px2_byte(q0, c) * ax0 + px2_byte(q0, c + 3) * ax1;
                    const int r1 = px2_byte(q1, c) * ax0 + px2_byte(q1, c + 3) * ax1;
                    px[c] = (((by0 * (r0 >> 4)) >> 16) + ((by1 * (r1 >> 4)) >> 16) + 2) >> 2;
                }
            } else
            for (int c = 0; c < 3; c++) {
                const int r0 = src[((int64_t)sy * it.SW + sx) * 3 + c] * ax0 + src[((int64_t)sy * it.SW + sx1) * 3 + c] * ax1;
                const int r1 = src[((int64_t)sy1 * it.SW + sx) * 3 + c] * ax0 + src[((int64_t)sy1 * it.SW + sx1) * 3 + c] * ax1;
                px[c] = (((by0 * (r0 >> 4)) >> 16) + ((by1 * (r1 >> 4)) >> 16) + 2) >> 2;
            }
        } else {
            const AreaAxis ax = area_axis(x, it.NW, it.SW), ay = area_axis(y, it.NH, it.SH);
            float acc[3] = {0.f, 0.f, 0.f};
            auto row = [&](int sy, float beta) {                  // horizontal pass of one source row into float, then * beta
                float rsum[3] = {0.f, 0.f, 0.f};
                const uint8_t* rp = src + (int64_t)sy * it.SW * 3;
                if (ax.has_lo) for (int c = 0; c < 3; c++) rsum[c] += rp[(ax.lo - 1) * 3 + c] * ax.wlo;
                for (int sx = ax.lo; sx < ax.hi; sx++) for (int c = 0; c < 3; c++) rsum[c] += rp[sx * 3 + c] * ax.wmid;
                if (ax.has_hi) for (int c = 0; c < 3; c++) rsum[c] += rp[ax.hi * 3 + c] * ax.whi;
                for (int c = 0; c < 3; c++) acc[c] += beta * rsum[c];
            };
            if (ay.has_lo) row(ay.lo - 1, ay.wlo);
            for (int sy = ay.lo; sy < ay.hi; sy++) row(sy, ay.wmid);
            if (ay.has_hi) row(ay.hi, ay.whi);
            for (int c = 0; c < 3; c++) px[c] = min(255, max(0, (int)rintf(acc[c])));
        }
        if (it.lut >= 0) hsv_lut_pixel(px[0], px[1], px[2], slut, divtab);
        uint8_t* d = stage + it.dst_off + (int64_t)i * 3;
        d[0] = (uint8_t)px[0]; d[1] = (uint8_t)px[1]; d[2] = (uint8_t)px[2];
    }
}

extern "C" int ryolo_resize_item_bytes(int* bytes) { if (!bytes) return RY_ERR_ARG; *bytes = (int)sizeof(ResizeItem); return RY_OK; }

extern "C" int ryolo_resize_hsv_batch(const uint8_t* pool, const void* items_dev, int nitems, int64_t max_pixels, const uint8_t* luts, uint8_t* stage,
                                      hipStream_t stream)
{
    if (nitems < 0 || max_pixels < 0) return RY_ERR_ARG;
    if (nitems == 0 || max_pixels == 0) return RY_OK;
    if (!pool || !items_dev || !stage || nitems > 65535 || max_pixels >= (1ll << 31)) return RY_ERR_ARG;
    // 8 pixels per thread: a workgroup's set-up (the hsv division tables: two double-precision divisions per thread; the whole-number test; the LUT
    // copy) cost more than its 1-2 pixels per thread did with one workgroup per 256 pixels
    const int64_t bx = ry_cdiv(max_pixels, 256 * 8);
    hipLaunchKernelGGL(resize_hsv_batch_kernel, dim3((unsigned)(bx < 1024 ? bx : 1024), (unsigned)nitems), dim3(256), 0, stream, pool,
                       reinterpret_cast<const ResizeItem*>(items_dev), luts, stage);
    RY_CHECK_LAUNCH();
    return RY_OK;
}

// ---- label side of the sample composition (datasets/base_dataset.py:188-222 load_target, :318-330 the mosaic-9 crop, lib/augmentations.py:
// 67-74 the warp of the vertices) for every label row a batch uses, one launch, element-wise: a row that a filter drops gets NaN vertices
// and is removed — in order — by the compaction of ryolo_encode_labels (its bounds test is false for NaN), so the surviving rows keep
// the reference's order without a scan here.  fp32 operations in the reference's order (-ffp-contract=off): x / w0, * w_, filter on
// the mean vertex (strict), + pad; [crop filter, - crop origin]; [M (double 3x3) applied in double, stored as float].
struct LabelRow {
    float poly[8];               // as parsed from the label file
    float cls;
    int slot;                    // image slot of the batch (column 0 of the result)
    float w0, h0;                // original image size; 0: labels are already normalised (normalized_labels)
    float w1, h1;                // size after load_image's resize
    float bx1, bx2, by1, by2;    // `boarder` of load_target (source-image coordinates); bx2 < 0: no filter
    float padw, padh;
    float cx1, cx2, cy1, cy2;    // mosaic-9 crop window on the 3s canvas; cx2 < 0: none.  The origin (cx1, cy1) is subtracted afterwards
    int mat;                     // index into the warp matrices, -1: none
};

__global__ __launch_bounds__(256) void label_stage_kernel(const LabelRow* __restrict__ rows, int64_t n, const double* __restrict__ mats,
                                                          float* __restrict__ out /*[n][10]*/)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const LabelRow r = rows[i];
    float q[8];
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
        float x = r.poly[k], y = r.poly[k + 1];
        if (r.w0 > 0.f) { x = x / r.w0; y = y / r.h0; }
        q[k] = x * r.w1;
        q[k + 1] = y * r.h1;
    }
    auto mean_ok = [&](float x1, float x2, float y1, float y2) {
        const float mx = (((q[0] + q[2]) + q[4]) + q[6]) / 4.0f;
        const float my = (((q[1] + q[3]) + q[5]) + q[7]) / 4.0f;
        return (mx > x1) && (mx < x2) && (my > y1) && (my < y2);
    };
    if (r.bx2 >= 0.f) ok = mean_ok(r.bx1, r.bx2, r.by1, r.by2);
#pragma unroll
    for (int k = 0; k < 8; k += 2) { q[k] = q[k] + r.padw; q[k + 1] = q[k + 1] + r.padh; }
    if (ok && r.cx2 >= 0.f) {
        ok = mean_ok(r.cx1, r.cx2, r.cy1, r.cy2);
#pragma unroll
        for (int k = 0; k < 8; k += 2) { q[k] = q[k] - r.cx1; q[k + 1] = q[k + 1] - r.cy1; }
    }
    if (r.mat >= 0) {
        const double* m = mats + (int64_t)r.mat * 9;
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
            const double x = (double)q[k], y = (double)q[k + 1];
            q[k] = (float)((m[0] * x + m[1] * y) + m[2]);
            q[k + 1] = (float)((m[3] * x + m[4] * y) + m[5]);
        }
    }
    float* o = out + i * 10;
    o[0] = (float)r.slot;
    o[1] = r.cls;
    const float nanv = __int_as_float(0x7fc00000);
#pragma unroll
    for (int k = 0; k < 8; k++) o[2 + k] = ok ? q[k] : nanv;
}

extern "C" int ryolo_label_row_bytes(int* bytes) { if (!bytes) return RY_ERR_ARG; *bytes = (int)sizeof(LabelRow); return RY_OK; }

extern "C" int ryolo_label_stage(const void* rows_dev, int64_t nrows, const double* mats, float* out, hipStream_t stream)
{
    if (nrows < 0) return RY_ERR_ARG;
    if (nrows == 0) return RY_OK;
    if (!rows_dev || !out) return RY_ERR_ARG;
    hipLaunchKernelGGL(label_stage_kernel, dim3((unsigned)ry_cdiv(nrows, 256)), dim3(256), 0, stream, reinterpret_cast<const LabelRow*>(rows_dev), nrows,
                       mats, out);
    RY_CHECK_LAUNCH();
    return RY_OK;
}

extern "C" int ryolo_letterbox_u8(const uint8_t* src, int SH, int SW, int NH, int NW, int top, int left, uint8_t* dst, int OH, int OW, int fill,
                                  hipStream_t stream)
{
    if (SH <= 0 || SW <= 0 || NH <= 0 || NW <= 0 || OH <= 0 || OW <= 0) return RY_ERR_ARG;
    if (!src || !dst) return RY_ERR_ARG;
    hipLaunchKernelGGL(letterbox_kernel, dim3((unsigned)ry_cdiv(OW, 256), OH), dim3(256), 0, stream, src, SH, SW, NH, NW, top, left, dst, OH, OW, fill);
    RY_CHECK_LAUNCH();
    return RY_OK;
}

static int paste_launch(const uint8_t* pool, const void* rects_dev, int nrect, const int* first, uint8_t* canvas, int ncanvas, int CH, int CW, int fill,
                        hipStream_t stream)
{
    if (ncanvas < 0 || CH < 0 || CW < 0 || nrect < 0) return RY_ERR_ARG;
    if (ncanvas == 0 || CH == 0 || CW == 0) return RY_OK;
    if (!canvas || (nrect && (!pool || !rects_dev))) return RY_ERR_ARG;
    hipLaunchKernelGGL(paste_rects_kernel, dim3((unsigned)ry_cdiv(CW, 1024), CH, ncanvas), dim3(256), 0, stream, pool,
                       reinterpret_cast<const PasteRect*>(rects_dev), nrect, first, canvas, CH, CW, fill);
    RY_CHECK_LAUNCH();
    return RY_OK;
}

extern "C" int ryolo_paste_rects(const uint8_t* pool, const void* rects_dev, int nrect, uint8_t* canvas, int ncanvas, int CH, int CW, int fill,
                                 hipStream_t stream)
{
    return paste_launch(pool, rects_dev, nrect, nullptr, canvas, ncanvas, CH, CW, fill, stream);
}

// the same with the rectangles grouped by canvas: first_dev [ncanvas + 1] ints, rectangles of canvas b = rects[first[b] .. first[b + 1])
extern "C" int ryolo_paste_rects_grouped(const uint8_t* pool, const void* rects_dev, int nrect, const int* first_dev, uint8_t* canvas, int ncanvas,
                                         int CH, int CW, int fill, hipStream_t stream)
{
    if (!first_dev) return RY_ERR_ARG;
    return paste_launch(pool, rects_dev, nrect, first_dev, canvas, ncanvas, CH, CW, fill, stream);
}

extern "C" int ryolo_paste_rect_bytes(int* bytes) { if (!bytes) return RY_ERR_ARG; *bytes = (int)sizeof(PasteRect); return RY_OK; }

extern "C" int ryolo_warp_perspective_u8(const uint8_t* src, int batch, int SH, int SW, const double* Minv, uint8_t* dst, int DH, int DW,
                                         int border, hipStream_t stream)
{
    if (batch < 0 || SH <= 0 || SW <= 0 || DH < 0 || DW < 0) return RY_ERR_ARG;
    if (batch == 0 || DH == 0 || DW == 0) return RY_OK;
    if (!src || !Minv || !dst) return RY_ERR_ARG;
    hipLaunchKernelGGL(warp_perspective_kernel, dim3((unsigned)ry_cdiv(DW, 256), DH, batch), dim3(256), 0, stream, src, SH, SW, Minv, dst, DH, DW,
                       border);
    RY_CHECK_LAUNCH();
    return RY_OK;
}

extern "C" int ryolo_hsv_gain_u8(uint8_t* img, int64_t npix, const uint8_t* lut, hipStream_t stream)
{
    if (npix < 0) return RY_ERR_ARG;
    if (npix == 0) return RY_OK;
    if (!img || !lut) return RY_ERR_ARG;
    hipLaunchKernelGGL(hsv_gain_kernel, dim3((unsigned)ry_cdiv(npix, 256)), dim3(256), 0, stream, img, npix, lut);
    RY_CHECK_LAUNCH();
    return RY_OK;
}

extern "C" int ryolo_mixup_u8(const uint8_t* a, const uint8_t* b, double r, int64_t n, uint8_t* out, hipStream_t stream)
{
    if (n < 0) return RY_ERR_ARG;
    if (n == 0) return RY_OK;
    if (!a || !b || !out) return RY_ERR_ARG;
    hipLaunchKernelGGL(mixup_kernel, dim3((unsigned)ry_cdiv(n, 256)), dim3(256), 0, stream, a, b, r, n, out);
    RY_CHECK_LAUNCH();
    return RY_OK;
}
