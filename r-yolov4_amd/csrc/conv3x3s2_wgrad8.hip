// 3x3 STRIDE-2 (pad 1) WEIGHT GRADIENT on parity-plane rings, 8 waves — gfx950 only.
// Reference rows served: SURVEY.md §8a M1 (`Conv`, model/utils.py:6-32: autograd of nn.Conv2d(k = 3, s = 2) w.r.t. its weight — the five
// down-sampling convolutions of yolov7's backbone / neck with more than 64 output channels, model/backbone.py:69-101, model/utils.py:146-160).
//
// dW[co][dh, dw][ci] = sum over output pixels p = (n, oh, ow) of dY[p][co] * X[n, 2 oh + dh, 2 ow + dw][ci].  The tapped kernels of conv.hip
// gather a fresh X tile per tap (every input pixel crosses L2 -> LDS 2.25 times, every dY tile once per (tap, 32-channel chunk) group) and ran
// these layers at 450-520 TF/s.  Here the stride disappears: X is read as its four PARITY PLANES P_ab[n, i, j] = X[n, 2 i + a, 2 j + b], each the
// size of the output grid, and on the planes the nine taps are offsets of 0 or -1:
//     plane 00: (0, 0)                                        tap (dh, dw) = (0, 0)
//     plane 01: (0, -1) (0, 0)                                taps (0, -1) (0, +1)
//     plane 10: (-1, 0) (0, 0)                                taps (-1, 0) (+1, 0)
//     plane 11: (-1, -1) (-1, 0) (0, -1) (0, 0)               taps (-1, -1) (-1, +1) (+1, -1) (+1, +1)
// so the halo-ring form of conv3x3_wgrad8.hip applies with one ring PER PLANE (instead of per input-channel chunk) over output coordinates
// padded by one row on top and one column on the left: a tap is a constant ring-row offset 0, -1, -PWq or -PWq - 1, padding comes from the zero
// page, every input pixel is requested exactly once per 128 output channels and dY once per 32 input channels.
// One 8-wave workgroup per CU owns 128 output channels x 32 input channels x 9 taps:
//     wave w:  h  = w >> 2        tap half: planes 11 + 00 (5 taps) or planes 01 + 10 (4 taps); waves w and w + 4 share a SIMD
//              pr = (w >> 1) & 1  output-channel pair (quarters 2 pr, 2 pr + 1)
//              kh = w & 1         pixel half of the 64-pixel step: two slabs per K range (the deterministic reduce of conv.hip adds them)
//   per 16-pixel slice 2 dY fragments + 5 (4) X fragments for 10 (8) MFMAs; per wave and step 2 ring pieces + 2 dY pieces by LDS-DMA.
// Everything else — slot-major padded dY stage, transposed fragment reads with counted lgkmcnt, mirrored ring head, scalar ring positions,
// multiply-high decomposition of padded indices, CU-exclusive workgroups on part of the chip — is conv3x3_wgrad8.hip's.
#include "conv_internal.h"
#include <stdlib.h>
#include <type_traits>

extern __shared__ __attribute__((aligned(1024))) unsigned char ws2_lds[];

#define WS2_MIRROR 32

template <int K> __device__ __forceinline__ void ws2_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K) : "memory"); }

template <int U, int N> struct WS2Unroll {
    template <class F> static __device__ __forceinline__ void run(F& f)
    {
        f(std::integral_constant<int, U>{});
        WS2Unroll<U + 1, N>::run(f);
    }
};
template <int N> struct WS2Unroll<N, N> {
    template <class F> static __device__ __forceinline__ void run(F&) {}
};
template <int N> __device__ __forceinline__ void ws2_wait6(ry_s16x4& a, ry_s16x4& b, ry_s16x4& c, ry_s16x4& d, ry_s16x4& e, ry_s16x4& f)
{
    asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "n"(N));
}

struct WS2Geom {
    int PWq, HPq;              // padded output width / height (OW + 1, OH + 1)
    int64_t Mp;                // NB * HPq * PWq
    int RX;                    // ring rows (multiple of 64) of each of the four plane rings
    int gx, gc, splitk, slabs;
    int64_t kchunk;
    int tap_of[9];             // caller's tap index of tap (dh, dw) at 3 (dh + 1) + (dw + 1)
    unsigned m_img, s_img, m_row, s_row;
    unsigned lds_bytes;
};

// the wave's tap list: (plane = 2 a + b, di, dj) per unit j of tap half H
__host__ __device__ constexpr int ws2_plane(int H, int j) { return H == 0 ? (j < 4 ? 3 : 0) : (j < 2 ? 1 : 2); }
__host__ __device__ constexpr int ws2_di(int H, int j) { return H == 0 ? (j < 2 ? -1 : 0) : (j == 2 ? -1 : 0); }
__host__ __device__ constexpr int ws2_dj(int H, int j) { return H == 0 ? ((j == 0 || j == 2) ? -1 : 0) : (j == 0 ? -1 : 0); }
__host__ __device__ constexpr int ws2_group(int H, int j) { return H == 0 ? (j < 2 ? 0 : (j < 4 ? 1 : 2)) : (j < 2 ? 0 : j - 1); }
__host__ __device__ constexpr bool ws2_first(int H, int j) { return j == 0 || ws2_group(H, j) != ws2_group(H, j - 1); }
__host__ __device__ constexpr int ws2_tap(int H, int j)                      // 3 (dh + 1) + (dw + 1)
{
    const int pl = ws2_plane(H, j), a = pl >> 1, b = pl & 1;
    const int dh = a == 0 ? 0 : (ws2_di(H, j) == -1 ? -1 : 1), dw = b == 0 ? 0 : (ws2_dj(H, j) == -1 ? -1 : 1);
    return 3 * (dh + 1) + (dw + 1);
}

__global__ __launch_bounds__(512, 1) void conv3x3s2_wgrad8_kernel(const WgradParams p, const WS2Geom g)
{
    constexpr int DSL = 1024 + 64, DQS = 4 * DSL, DYS = 4 * DQS;    // dY stage: [4 quarters][4 slots of 8 channels, 64 B of padding each][64 px][16 B]
    constexpr int NKS = 2;                                           // 16-pixel slices a wave multiplies per step (its pixel half)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = wave >> 2, pr = (wave >> 1) & 1, kh = wave & 1;
    const int q0 = 2 * pr;
    const int t_id = xcd_remap(blockIdx.x, gridDim.x);
    const int bx = t_id % g.gx, bc = (t_id / g.gx) % g.gc, bz = t_id / (g.gx * g.gc);
    const int i0 = bx * 128, ci0 = bc * 32;
    const int64_t kbeg = (int64_t)bz * g.kchunk;
    const int64_t kend = min(g.Mp, kbeg + g.kchunk);
    if (kbeg >= kend) return;
    const int nk = (int)((kend - kbeg + 63) >> 6);
    const int OH = p.OH, OW = p.OW, IH = p.IH, IW = p.IW, PWq = g.PWq, HPq = g.HPq;
    const int HALO = PWq + 1;
    const int RX = g.RX;
    const unsigned RB = (unsigned)(RX + WS2_MIRROR) * 64u;
    unsigned char* const dyst = ws2_lds + 4u * RB;
    const int kend32 = (int)kend, Mp32 = (int)g.Mp;
    const unsigned per = (unsigned)(HPq * PWq);
    // padded index q -> (image, padded row ip, padded column jp); row 0 / column 0 are the padding
    auto split = [&](int q, unsigned& img, unsigned& ip, unsigned& jp) {
        const unsigned uq = (unsigned)q;
        img = __umulhi(uq, g.m_img) >> g.s_img;
        const unsigned rem = uq - img * per;
        ip = __umulhi(rem, g.m_row) >> g.s_row;
        jp = rem - ip * (unsigned)PWq;
    };
    // ---- LDS-DMA of a wave per step: ring pieces 2 (w & 1), 2 (w & 1) + 1 (16 rows x 64 B each) of plane w >> 1; dY quarter w >> 1, slots
    // 2 (w & 1), 2 (w & 1) + 1 (a lane owns pixel row `lane`)
    const int xpl = wave >> 1, xa = xpl >> 1, xb = xpl & 1;
    const int xr0 = 32 * (wave & 1);
    const int dqr = wave >> 1, ds0 = 2 * (wave & 1);
    const bf16_t* const x_base = p.X + ci0 + (lane & 3) * 8;
    const bf16_t* const dy_base = p.dY + i0 + 32 * dqr + ds0 * 8;
    const bool d_ok0 = (i0 + 32 * dqr + ds0 * 8) < p.CoutPad, d_ok1 = (i0 + 32 * dqr + ds0 * 8 + 8) < p.CoutPad;
    const int x0 = (int)(((kbeg - HALO) >> 6) << 6);
    const int pro_iters = ((int)kbeg + 64 + 16 - x0 + 63) >> 6;
    int xq = x0 + xr0 + (lane >> 2);                                  // this lane's padded position in piece 0 of the next ring request (piece 1: + 16)
    int xslot = xr0;                                                  // (scalar) ring row piece 0 lands on
    int dq = (int)kbeg + lane;
    const bf16_t *xsrc0, *xsrc1, *dsrc;                               // nullptr: padding / out of range -> the zero page
    auto locate_x = [&](int q) -> const bf16_t* {
        unsigned img, ip, jp;
        split(q, img, ip, jp);
        const unsigned ih = 2u * (ip - 1u) + (unsigned)xa, iw = 2u * (jp - 1u) + (unsigned)xb;      // (ip = 0 / jp = 0 wrap to huge values: rejected below)
        const bool ok = (unsigned)q < (unsigned)Mp32 && ip >= 1u && jp >= 1u && ih < (unsigned)IH && iw < (unsigned)IW;
        return ok ? x_base + (int64_t)((img * (unsigned)IH + ih) * (unsigned)IW + iw) * p.ldX : nullptr;
    };
    auto prep_x0 = [&]() { xsrc0 = locate_x(xq); };
    auto prep_x1 = [&]() { xsrc1 = locate_x(xq + 16); };
    auto prep_dy = [&]() {
        unsigned img, ip, jp;
        split(dq, img, ip, jp);
        const bool ok = (unsigned)dq < (unsigned)kend32 && ip >= 1u && jp >= 1u;
        dsrc = ok ? dy_base + (int64_t)((img * (unsigned)OH + ip - 1u) * (unsigned)OW + jp - 1u) * p.ldY : nullptr;
    };
    unsigned char* const xring_w = ws2_lds + (unsigned)xpl * RB;
    auto issue_x = [&](int u) {
        const bf16_t* s_ = u ? xsrc1 : xsrc0;
        const bf16_t* src = s_ ? s_ : p.zeros;
        const int sl = xslot + 16 * u;                                // pieces never straddle the wrap: RX and the slots are multiples of 16
        __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(xring_w + (unsigned)sl * 64u), 16, 0, 0);
        if (sl < WS2_MIRROR)
            __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(xring_w + (unsigned)(sl + RX) * 64u), 16, 0, 0);
        if (u == 1) {
            xq += 64;
            xslot += 64;
            if (xslot >= RX) xslot -= RX;
        }
    };
    auto issue_dy1 = [&](int stage, int u) {
        const bf16_t* src = (dsrc && (u ? d_ok1 : d_ok0)) ? dsrc + u * 8 : p.zeros;
        __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(dyst + stage * DYS + dqr * DQS + (ds0 + u) * DSL), 16, 0, 0);
        if (u == 1) dq += 64;
    };
    for (int it = 0; it < pro_iters; it++) { prep_x0(); prep_x1(); issue_x(0); issue_x(1); }      // rows of step 0 with the halo in front
    prep_dy();
    issue_dy1(0, 0);
    issue_dy1(0, 1);
    prep_x0();                                                        // the requests of step 0 (operands of step 1)
    prep_x1();
    prep_dy();

    const int s16 = lane & 15, grp = lane >> 4;
    const int fr_row = (grp >> 1) * 8 + (s16 >> 2);
    const int fr_col = (16 * (grp & 1) + 4 * (s16 & 3)) * 2;
    const unsigned xr_l = lds_addr(ws2_lds) + (unsigned)(fr_row * 64 + fr_col);                    // lane constant inside a ring (plane base added per group)
    const unsigned da_l = lds_addr(dyst) + (unsigned)(q0 * DQS + (fr_col >> 4) * DSL + (32 * kh + fr_row) * 16 + (fr_col & 15));
    int rp = (int)kbeg + 32 * kh - x0;                                // (scalar) ring row of this wave's first pixel of the step
    if (rp >= RX) rp -= RX;

    auto body = [&](auto hc) {
        constexpr int HH = decltype(hc)::value;
        constexpr int NT = 5 - HH;
        constexpr int NU = NKS * NT;
        constexpr int PF = NT;
        f32x16 acc[NT][2];
#pragma unroll
        for (int j = 0; j < NT; j++)
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int e = 0; e < 16; e++) acc[j][a][e] = 0.f;
        for (int s = 0; s < nk; s++) {
            ws2_wait_vm<0>();
            __builtin_amdgcn_s_barrier();
            const bool more = s + 1 < nk;
            const unsigned da_s = da_l + (unsigned)((s & 1) * DYS);
            ry_s16x4 al[2][2], ah[2][2], bl[NU], bh[NU];
            unsigned gaddr[3 * NKS];
            auto read_u = [&](auto uc) {
                constexpr int U = decltype(uc)::value;
                constexpr int ks = U / NT, j = U % NT, gi = 3 * ks + ws2_group(HH, j);
                if constexpr (j == 0) {
                    constexpr unsigned ao = (unsigned)(16 * ks * 16);
                    al[ks & 1][0] = lds_tr16_off<ao>(da_s);
                    ah[ks & 1][0] = lds_tr16_off<ao + 64>(da_s);
                    al[ks & 1][1] = lds_tr16_off<ao + DQS>(da_s);
                    ah[ks & 1][1] = lds_tr16_off<ao + DQS + 64>(da_s);
                }
                if constexpr (ws2_first(HH, j)) {                     // base of a (plane, row offset) group at ITS first tap's column offset, wrapped once
                    int v = rp + 16 * ks + ws2_di(HH, j) * PWq + ws2_dj(HH, j);
                    if (v < 0) v += RX;
                    if (v >= RX) v -= RX;
                    gaddr[gi] = xr_l + (unsigned)ws2_plane(HH, j) * RB + ((unsigned)v << 6);
                }
                constexpr int jf = (ws2_first(HH, j) ? j : j - 1);    // groups hold at most two taps (dj = -1, 0)
                constexpr int imm = (ws2_dj(HH, j) - ws2_dj(HH, jf)) * 64;
                bl[U] = lds_tr16_off<imm>(gaddr[gi]);
                bh[U] = lds_tr16_off<imm + 256>(gaddr[gi]);
            };
            WS2Unroll<0, PF>::run(read_u);
            bf16x8 af[2][2];
            auto unit = [&](auto uc) {
                constexpr int U = decltype(uc)::value;
                constexpr int ks = U / NT, j = U % NT;
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (U + PF < NU) read_u(std::integral_constant<int, U + PF>{});
                constexpr int ahead = (U + PF < NU ? U + PF : NU - 1);
                constexpr int N = 2 * (ahead - U) + ((ahead / NT > U / NT) ? 4 : 0);
                static_assert(N <= 15, "lgkmcnt is a 4-bit counter");
                if constexpr (j == 0) {
                    ws2_wait6<N>(al[ks & 1][0], ah[ks & 1][0], al[ks & 1][1], ah[ks & 1][1], bl[U], bh[U]);
                    af[ks & 1][0] = join_halves(al[ks & 1][0], ah[ks & 1][0]);
                    af[ks & 1][1] = join_halves(al[ks & 1][1], ah[ks & 1][1]);
                } else {
                    lds_wait_h<N>(bl[U], bh[U]);
                }
                __builtin_amdgcn_sched_barrier(0);
                const bf16x8 bf = join_halves(bl[U], bh[U]);
                acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks & 1][0], bf, acc[j][0], 0, 0, 0);
                acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks & 1][1], bf, acc[j][1], 0, 0, 0);
                if constexpr (U <= 3) {                               // the step's four LDS-DMA requests behind the first units' MFMAs
                    __builtin_amdgcn_sched_barrier(0);
                    if (more) {
                        if constexpr (U < 2) issue_x(U);
                        else issue_dy1((s + 1) & 1, U - 2);
                    }
                }
                if constexpr (U == NU / 2 || U == NU / 2 + 1 || U == NU / 2 + 2) {     // ... the next step's source pointers behind later ones
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (U == NU / 2) prep_x0();
                    else if constexpr (U == NU / 2 + 1) prep_x1();
                    else prep_dy();
                }
            };
            WS2Unroll<0, NU>::run(unit);
            __builtin_amdgcn_sched_barrier(0);
            rp += 64;
            if (rp >= RX) rp -= RX;
        }
        const int NK = 9 * p.Cin;
        float* part = p.partial + ((int64_t)bz * 2 + kh) * p.Cout * NK;
#pragma unroll
        for (int j = 0; j < NT; j++) {
            const int kc = g.tap_of[ws2_tap(HH, j)] * p.Cin + ci0 + (lane & 31);
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const int co = i0 + 32 * (q0 + a) + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                    if (co < p.Cout) part[(int64_t)co * NK + kc] = acc[j][a][e];
                }
        }
    };
    if (h == 0) body(std::integral_constant<int, 0>{});
    else body(std::integral_constant<int, 1>{});
}

static bool ws2_geom(const WgradParams& p, WS2Geom& g)
{
    // OFF by default (RYOLO_WS2_8W=1 selects it; tests/test_gpu_wgrad_taps.py runs its cases that way).  Measured: alone on whole-chip grids +15-25 %
    // over the tapped LDS-DMA kernel (530-557 vs 440-500 TF/s on the five layers), in the training step NEUTRAL at every grid size (96 workgroups
    // -0.9 %, 128 / 160 / 192 / 256 +-0.2 %, alternating same-box runs): with four plane rings in LDS a step is 64 pixels x 32 input channels — 36
    // MFMAs per SIMD between two barriers for 4 LDS-DMA requests and 3 index decompositions per wave, a quarter of the matrix work per step of
    // conv3x3_wgrad8.hip — and what it saves in L2 traffic the tapped kernel, thin on the whole chip, was already hiding under the main stream.
    static const int on = getenv("RYOLO_WS2_8W") ? atoi(getenv("RYOLO_WS2_8W")) : 0;
    g = WS2Geom{};
    if (!on || !p.zeros || p.ntaps != 9 || p.sh != 2 || p.sw != 2) return false;
    if (p.OH != (p.IH - 1) / 2 + 1 || p.OW != (p.IW - 1) / 2 + 1) return false;                       // k = 3, pad 1
    if (p.Cin % 32 || p.Cout <= 64 || p.ldX % 8 || p.ldY % 8 || p.CoutPad % 8 || p.CoutPad < p.Cout || p.CoutPad > p.ldY) return false;
    if ((reinterpret_cast<uintptr_t>(p.dY) | reinterpret_cast<uintptr_t>(p.X)) & 15) return false;
    unsigned seen = 0;
    for (int t = 0; t < 9; t++) {
        if (p.dh[t] < -1 || p.dh[t] > 1 || p.dw[t] < -1 || p.dw[t] > 1) return false;
        seen |= 1u << ((p.dh[t] + 1) * 3 + p.dw[t] + 1);
        g.tap_of[(p.dh[t] + 1) * 3 + p.dw[t] + 1] = t;
    }
    if (seen != 0x1ffu) return false;
    g.PWq = p.OW + 1;
    g.HPq = p.OH + 1;
    g.Mp = (int64_t)p.NB * g.HPq * g.PWq;
    if (g.Mp >= (1ll << 31) || (int64_t)p.NB * p.IH * p.IW >= (1ll << 31)) return false;
    const int need = (g.PWq + 1) + 209;                              // one halo (in front) + this step + the next + alignment slack
    g.RX = (int)ry_cdiv(need, 64) * 64;
    g.lds_bytes = 4u * (unsigned)(g.RX + WS2_MIRROR) * 64u + 2u * 4u * 4352u;
    if (g.lds_bytes > 160u * 1024u) return false;                    // (the 32 -> 64 layer at 800 -> 400 would not fit either way: Cout <= 64)
    g.gx = (int)ry_cdiv(p.Cout, 128);
    g.gc = p.Cin / 32;
    static const int target = getenv("RYOLO_WS2_8W_BLOCKS") ? atoi(getenv("RYOLO_WS2_8W_BLOCKS")) : 96;     // part of the chip: conv3x3_wgrad8.hip
    int64_t sk = ry_cdiv(target, (int64_t)g.gx * g.gc);
    const int64_t maxsplit = g.Mp / (16 * 64);                       // at least 16 steps per split (a one-sided halo prologue: cheaper than the stride-1 ring's)
    if (sk > maxsplit) sk = maxsplit;
    if (sk < 1) sk = 1;
    static const bool force = getenv("RYOLO_WS2_8W_FORCE") != nullptr;
    if ((int64_t)g.gx * g.gc * sk < 48 && !force) return false;
    g.kchunk = ry_cdiv(ry_cdiv(g.Mp, sk), 64) * 64;
    g.splitk = (int)ry_cdiv(g.Mp, g.kchunk);
    g.slabs = g.splitk * 2;
    auto magic = [](unsigned d, unsigned& m, unsigned& sh) {         // n / d == mulhi(n, m) >> sh for 0 <= n < 2^31, d >= 2
        unsigned l = 0;
        while ((1ull << l) < d) l++;
        m = (unsigned)((((unsigned long long)1 << (31 + l)) + d - 1) / d);
        sh = l - 1;
    };
    if (g.PWq < 2 || g.HPq * g.PWq < 2) return false;
    magic((unsigned)(g.HPq * g.PWq), g.m_img, g.s_img);
    magic((unsigned)g.PWq, g.m_row, g.s_row);
    return true;
}

// eligibility + what ryolo_conv_wgrad_plan / _grid report (slabs = fp32 partial tiles the reduce adds, workgroups of the launch)
bool ws2_geometry(const WgradParams& p, int* slabs, int* workgroups)
{
    WS2Geom g;
    if (!ws2_geom(p, g)) return false;
    *slabs = g.slabs;
    *workgroups = g.gx * g.gc * g.splitk;
    return true;
}

int ws2_launch(const WgradParams& p, hipStream_t stream)
{
    WS2Geom g;
    if (!ws2_geom(p, g)) return RY_ERR_ARG;
    static RyLdsAttr attr;
    if (ry_max_dynamic_lds(attr, reinterpret_cast<const void*>(&conv3x3s2_wgrad8_kernel), 160 * 1024)) return RY_ERR_LAUNCH;
    const dim3 grid((unsigned)((int64_t)g.gx * g.gc * g.splitk));
    hipLaunchKernelGGL(conv3x3s2_wgrad8_kernel, grid, dim3(512), g.lds_bytes, stream, p, g);
    return hipGetLastError() == hipSuccess ? RY_OK : RY_ERR_LAUNCH;
}
