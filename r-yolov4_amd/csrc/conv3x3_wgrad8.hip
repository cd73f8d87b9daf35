// 3x3 stride-1 WEIGHT GRADIENT, 8-wave form of the halo-ring kernel (conv3x3.hip) — gfx950 only.
// Reference rows served: SURVEY.md §8a M1 (`Conv`, model/utils.py:6-32: autograd of nn.Conv2d(k = 3, s = 1) w.r.t. its weight).
//
// dW[co][tap][ci] = sum_p dY[p][co] * X[p + tap][ci], K = padded pixels, one sliding ring of input rows per 32-channel chunk, padding from a
// zero page: everything conv3x3.hip says about the ring form holds.  What round 5 changes is the SHAPE of the work per wave.
//
// The 4-wave kernels give wave w one 32-channel quarter of dY and all 9 taps: per 16-pixel slice 1 dY fragment + 9 X fragments for 9 MFMAs —
// 1.11 KiB of ds_read_b64_tr_b16 traffic per MFMA, every X fragment read by all four waves.  Per SIMD and 64-pixel step (two workgroups per
// CU) that is 80 KiB of LDS reads + 10 LDS-DMA requests for 72 MFMAs; the phase stamps of round 4 said 3400 cycles per step against 2304 of
// matrix time with the waves never waiting — the loop was issuing, and most of what it issued were those reads and requests.
// Here ONE 8-wave workgroup per CU owns 128 (64) output channels x 64 input channels x 9 taps, and a wave owns a 2 x (5 | 4) block of the
// (output-channel quarter) x (tap) grid for one input-channel chunk:
//     wave w:  h  = w >> 2        tap half — taps 0..4 or 5..8 in row-major (dh, dw) order; waves w and w + 4 share a SIMD (waves are dealt
//                                 to the SIMDs cyclically), so every SIMD gets 5 + 4 taps: 18 accumulator tiles, like two waves of the old form
//              cc = w & 1         input-channel chunk (its own ring)
//              NCO = 4 (Cout > 64):  pr = (w >> 1) & 1   output-channel PAIR (quarters 2 pr, 2 pr + 1), all 64 pixels of a step
//              NCO = 2 (Cout <= 64): kh = (w >> 1) & 1   pixel HALF of the step (both quarters), two slabs per K range
//   per 16-pixel slice: 2 dY fragments + 5 (4) X fragments for 10 (8) MFMAs — 0.72 KiB of LDS reads per MFMA instead of 1.11;
//   LDS-DMA requests per wave and step: 1 ring piece + 2 (1) dY pieces = 3 (2) instead of 5 (3): 24 requests per 288 MFMAs instead of 40;
//   dY is shared by the two input-channel chunks (the old form streamed it once per chunk), X by the two output-channel pairs.
// Rings of ANY length (a multiple of 64 rows, not a power of two): a ring position is a scalar carried from step to step with one
// conditional subtract, the eight (slice, kernel row) bases of a step are scalar adds + one wrap each, and a lane adds its constant row
// offset — one VALU instruction per six transposed reads (the three taps of a kernel row and the +4-row second read are immediates).
// An immediate cannot wrap: the ring is followed by a copy of its first 32 rows (the two pieces that land there are requested twice).
// W = 25 / 50 maps need 320 rows (the power-of-two form took 512), W = 100 448, W = 200 640.
// Split-K slabs + the deterministic reduce of conv.hip, XCD-aware order, padded-coordinate decomposition by multiply-high: unchanged.
#include "conv_internal.h"
#include <stdlib.h>
#include <type_traits>

extern __shared__ __attribute__((aligned(1024))) unsigned char w8_lds[];

#define W8_MIRROR 32                                               // ring rows repeated behind the ring (lane row <= 11, +2 taps, +4 second read)

template <int K> __device__ __forceinline__ void w8_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K) : "memory"); }

template <int U, int N> struct W8Unroll {
    template <class F> static __device__ __forceinline__ void run(F& f)
    {
        f(std::integral_constant<int, U>{});
        W8Unroll<U + 1, N>::run(f);
    }
};
template <int N> struct W8Unroll<N, N> {
    template <class F> static __device__ __forceinline__ void run(F&) {}
};

// waits tied to fragment halves (conv_internal.h: the consumer cannot move above the wait, the halves are joined behind it)
template <int N> __device__ __forceinline__ void w8_wait6(ry_s16x4& a, ry_s16x4& b, ry_s16x4& c, ry_s16x4& d, ry_s16x4& e, ry_s16x4& f)
{
    asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "n"(N));
}

template <int NCO>
__global__ __launch_bounds__(512, 1) void conv3x3_wgrad8_kernel(const WgradParams p, const W3Geom g)
{
    // one dY stage: [NCO quarters][4 slots of 8 channels][64 px][16 B] with 64 bytes of padding behind every slot: a transposed fragment read
    // touches 4 rows x 16 B of each of the 4 slots, and with 1024-byte slots all four sit on the same 16 banks — a 4-way conflict on every dY
    // read (PMC of the first cut: SQ_LDS_BANK_CONFLICT 48 % of SQ_LDS_IDX_ACTIVE; the 4-wave kernels have the same layout but read dY a
    // quarter as often).  1088-byte slots put them on banks 0 / 16 / 32 / 48.
    constexpr int DSL = 1024 + 64, DQS = 4 * DSL, DYS = NCO * DQS;
    constexpr int NKS = NCO == 4 ? 4 : 2;                            // 16-pixel slices a wave multiplies per step
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = wave >> 2, cc = wave & 1, sel = (wave >> 1) & 1;
    const int q0 = NCO == 4 ? 2 * sel : 0;                           // first output-channel quarter of this wave
    const int kh = NCO == 4 ? 0 : sel;                               // pixel half of the step (NCO = 2)
#ifdef W3_TIMING
    const unsigned long long T0 = __builtin_readcyclecounter();
#endif
    const int t_id = xcd_remap(blockIdx.x, gridDim.x);
    const int bx = t_id % g.gx, bc = (t_id / g.gx) % g.gc, bz = t_id / (g.gx * g.gc);
    const int i0 = bx * 128, ci0 = bc * 64;
    const int64_t kbeg = (int64_t)bz * g.kchunk;
    const int64_t kend = min(g.Mp, kbeg + g.kchunk);
    if (kbeg >= kend) return;
    const int nk = (int)((kend - kbeg + 63) >> 6);
    const int H = p.OH, W = p.OW, PWp = g.PWp, HPp = g.HPp;
    const int HALO = PWp + 1;
    const int RX = g.RX;
    // PD = prefetch distance in steps (g.pd): the requests issued during step s bring the operands of step s + PD into dY stage (s + PD) % (PD + 1)
    // and PD * 64 rows further down the rings.  All 8 waves of the CU meet at ONE barrier per step, so whatever a step waits for at its top is
    // exposed on every SIMD at once (the 4-wave kernels ran two independent workgroups per CU): PD = 1 left 27 % of the wave cycles parked in
    // s_waitcnt / s_barrier (PMC, profiles/r05_pmc_wgrad_ring) — a step of ~1 us does not cover an HBM round trip under load; PD = 2 does,
    // for one more dY stage and 64 more ring rows of LDS (the CU is this workgroup's alone anyway).
    const int PD = g.pd;
    const unsigned RB = (unsigned)(RX + W8_MIRROR) * 64u;             // bytes of one ring with its mirrored head
    unsigned char* const dyst = w8_lds + 2u * RB;
    const int kend32 = (int)kend, Mp32 = (int)g.Mp;
    const unsigned per = (unsigned)(HPp * PWp);
    auto locate = [&](int q, int limit, bool& ok) -> int {           // pixel index of padded position q (valid iff ok)
        const unsigned uq = (unsigned)q;
        const unsigned img = __umulhi(uq, g.m_img) >> g.s_img;
        const unsigned rem = uq - img * per;
        const unsigned ihp = __umulhi(rem, g.m_row) >> g.s_row;
        const unsigned iwp = rem - ihp * (unsigned)PWp;
        ok = uq < (unsigned)limit && (ihp - 1u) < (unsigned)H && (iwp - 1u) < (unsigned)W;
        return (int)((img * (unsigned)H + ihp - 1u) * (unsigned)W + iwp - 1u);
    };
    // ---- LDS-DMA requests of a wave per step.  Ring: piece xp = w & 3 (16 rows x 64 B: lane -> row lane >> 2, 16-byte slot lane & 3) of
    // chunk xc = w >> 2.  dY (slot-major stage: a lane owns pixel row `lane`, an instruction moves one 8-channel slot of all 64 rows):
    // NCO = 4: quarter w >> 1, slots 2 (w & 1) and 2 (w & 1) + 1; NCO = 2: quarter w >> 2, slot w & 3.  ONE padded-pixel decomposition
    // per stream, lane and step.
    const int xc = wave >> 2, xp = wave & 3;
    const int dqr = NCO == 4 ? wave >> 1 : wave >> 2;                 // dY quarter this wave stages
    const int ds0 = NCO == 4 ? 2 * (wave & 1) : (wave & 3);           // first (only) slot
    constexpr int NDY = NCO == 4 ? 2 : 1;
    const bf16_t* const x_base = p.X + ci0 + 32 * xc + (lane & 3) * 8;
    const bf16_t* const dy_base = p.dY + i0 + 32 * dqr + ds0 * 8;
    const bool d_ok0 = (i0 + 32 * dqr + ds0 * 8) < p.CoutPad, d_ok1 = (i0 + 32 * dqr + ds0 * 8 + 8) < p.CoutPad;
    const int x0 = (int)(((kbeg - HALO) >> 6) << 6);                 // ring origin: aligned down to 64 (arithmetic shift: also for negatives)
    const int pro_iters = ((int)kbeg + 64 * PD + HALO + 16 - x0 + 63) >> 6;
    int xq = x0 + 16 * xp + (lane >> 2);                              // this lane's padded row of the next ring request
    int xslot = 16 * xp;                                              // (scalar) ring row the next piece lands on
    int dq = (int)kbeg + lane;                                        // this lane's dY pixel of the next request
    const bf16_t *xsrc, *dsrc;                                        // nullptr: padding / out of range -> the zero page
    auto prep_x = [&]() {
        bool ok;
        const int pix = locate(xq, Mp32, ok);
        xsrc = ok ? x_base + (int64_t)pix * p.ldX : nullptr;
    };
    auto prep_dy = [&]() {
        bool ok;
        const int pix = locate(dq, kend32, ok);
        dsrc = ok ? dy_base + (int64_t)pix * p.ldY : nullptr;
    };
    unsigned char* const xring_w = w8_lds + (unsigned)xc * RB;        // the ring this wave fills
    int n_issued = 0;                                                 // (scalar) LDS-DMA instructions of the current step so far
    auto issue_x = [&]() {
        const bf16_t* src = xsrc ? xsrc : p.zeros;
        __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(xring_w + (unsigned)xslot * 64u), 16, 0, 0);
        n_issued++;
        if (xslot < W8_MIRROR) {                                      // wave-uniform: twice per lap of the ring
            __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(xring_w + (unsigned)(xslot + RX) * 64u), 16, 0, 0);
            n_issued++;
        }
        xq += 64;
        xslot += 64;
        if (xslot >= RX) xslot -= RX;
    };
    auto issue_dy1 = [&](int stage, int u) {
        const bf16_t* src = (dsrc && (u ? d_ok1 : d_ok0)) ? dsrc + u * 8 : p.zeros;
        __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(dyst + stage * DYS + dqr * DQS + (ds0 + u) * DSL), 16, 0, 0);
        n_issued++;
        if (u == NDY - 1) dq += 64;
    };
    for (int it = 0; it < pro_iters; it++) { prep_x(); issue_x(); }   // rows of steps 0 .. PD - 1 with both halos
    prep_dy();
#pragma unroll
    for (int u = 0; u < NDY; u++) issue_dy1(0, u);
    int pend = 0;                                                     // (scalar) LDS-DMA instructions this wave issued for a LATER step than the next one to run
    if (PD == 2 && nk > 1) {                                          // dY of step 1: the only requests that may still be in flight when step 0 starts
        prep_dy();
#pragma unroll
        for (int u = 0; u < NDY; u++) issue_dy1(1, u);
        pend = NDY;
    }
    prep_x();                                                         // the requests of step 0 (operands of step PD)
    prep_dy();
    auto wait_pending = [&]() {                                       // "at most `pend` of my requests in flight": in-order return = everything older has landed
        if (pend == 0) w8_wait_vm<0>();
        else if (pend == 1) w8_wait_vm<1>();
        else if (pend == 2) w8_wait_vm<2>();
        else if (pend == 3) w8_wait_vm<3>();
        else w8_wait_vm<4>();
    };

    // ---- fragment addressing: transposed reads, lane -> (pixel row, channel) inside a 16-lane group (conv.hip)
    const int s16 = lane & 15, grp = lane >> 4;
    const int fr_row = (grp >> 1) * 8 + (s16 >> 2);
    const int fr_col = (16 * (grp & 1) + 4 * (s16 & 3)) * 2;
    const unsigned xr_l = lds_addr(w8_lds) + (unsigned)cc * RB + (unsigned)(fr_row * 64 + fr_col);     // lane constant inside this wave's ring
    // dY fragment (slot-major stage): channel fr_col / 2 = 8 * slot + c of quarter q0 (+ 1), row 32 kh + 16 ks + fr_row (+ 4)
    const unsigned da_l = lds_addr(dyst) + (unsigned)(q0 * DQS + (fr_col >> 4) * DSL + (32 * kh + fr_row) * 16 + (fr_col & 15));
    int rp = (int)kbeg + 32 * kh - x0;                                // (scalar) ring row of this wave's first pixel of the step, < RX:
    if (rp >= RX) rp -= RX;                                           //   kbeg - x0 < HALO + 64 and RX >= 2 HALO + 209
#ifdef W3_TIMING
    const unsigned long long T1 = __builtin_readcyclecounter();
    unsigned long long t_wait = 0, t_bar = 0;
#endif

    auto body = [&](auto hc) {
        constexpr int HH = decltype(hc)::value;                       // tap half
        constexpr int NT = 5 - HH;                                    // taps of this wave: row-major indices 5 HH ... 5 HH + NT - 1
        constexpr int NU = NKS * NT;                                  // units (slice, tap) per step: one X fragment, two MFMAs each
        constexpr int PF = NT;                                        // units of read-ahead: exactly one dY pair among any PF consecutive units
        f32x16 acc[NT][2];
#pragma unroll
        for (int j = 0; j < NT; j++)
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int e = 0; e < 16; e++) acc[j][a][e] = 0.f;
        int st_cur = 0, st_nxt = PD == 2 ? 2 : 1;                     // dY stage of this step, stage the step's requests fill: (s + PD) % (PD + 1)
        for (int s = 0; s < nk; s++) {
#ifdef W3_TIMING
            const unsigned long long tw0 = __builtin_readcyclecounter();
#endif
            wait_pending();                                           // this wave's requests for step s have landed (those for step s + 1 may fly on: PD = 2)
#ifdef W3_TIMING
            const unsigned long long tw1 = __builtin_readcyclecounter();
#endif
            __builtin_amdgcn_s_barrier();                             // ... and everybody else's; step s - 1 fully consumed
#ifdef W3_TIMING
            t_wait += tw1 - tw0;
            t_bar += __builtin_readcyclecounter() - tw1;
#endif
            const bool more = s + PD < nk;
            const unsigned da_s = da_l + (unsigned)(st_cur * DYS);
            n_issued = 0;
            ry_s16x4 al[2][2], ah[2][2], bl[NU], bh[NU];              // [slice parity][quarter]: the dY pair of slice ks + 1 is read under slice ks
            unsigned gaddr[2 * NKS];                                  // lane address of the dw = -1 fragment of a (slice, kernel row) group
            auto read_u = [&](auto uc) {
                constexpr int U = decltype(uc)::value;
                constexpr int ks = U / NT, j = U % NT, t = 5 * HH + j, dhi = t / 3, dwi = t % 3, gi = 2 * ks + (dhi - HH);
                if constexpr (j == 0) {                               // the dY pair of this slice first
                    constexpr unsigned ao = (unsigned)(16 * ks * 16);
                    al[ks & 1][0] = lds_tr16_off<ao>(da_s);
                    ah[ks & 1][0] = lds_tr16_off<ao + 64>(da_s);
                    al[ks & 1][1] = lds_tr16_off<ao + DQS>(da_s);
                    ah[ks & 1][1] = lds_tr16_off<ao + DQS + 64>(da_s);
                }
                if constexpr (j == 0 || dwi == 0) {                   // first tap of a kernel row in this wave's list: its base, wrapped once
                    int v = rp + 16 * ks + (dhi - 1) * PWp - 1;       // scalar; |16 ks + (dh) PWp - 1| < RX
                    if (v < 0) v += RX;
                    if (v >= RX) v -= RX;
                    gaddr[gi] = xr_l + ((unsigned)v << 6);
                }
                bl[U] = lds_tr16_off<dwi * 64>(gaddr[gi]);
                bh[U] = lds_tr16_off<dwi * 64 + 256>(gaddr[gi]);
            };
            W8Unroll<0, PF>::run(read_u);
            bf16x8 af[2][2];
            auto unit = [&](auto uc) {
                constexpr int U = decltype(uc)::value;
                constexpr int ks = U / NT, j = U % NT;
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (U + PF < NU) read_u(std::integral_constant<int, U + PF>{});
                // LDS returns in order: "at most N later reads in flight" = fragment U (and everything read before it) has landed.
                // N = 2 per X fragment read after U + 4 if a dY pair was read among them.
                constexpr int ahead = (U + PF < NU ? U + PF : NU - 1);
                constexpr int N = 2 * (ahead - U) + ((ahead / NT > U / NT) ? 4 : 0);
                static_assert(N <= 15, "lgkmcnt is a 4-bit counter");
                if constexpr (j == 0) {
                    w8_wait6<N>(al[ks & 1][0], ah[ks & 1][0], al[ks & 1][1], ah[ks & 1][1], bl[U], bh[U]);
                    af[ks & 1][0] = join_halves(al[ks & 1][0], ah[ks & 1][0]);
                    af[ks & 1][1] = join_halves(al[ks & 1][1], ah[ks & 1][1]);
                } else {
                    lds_wait_h<N>(bl[U], bh[U]);
                }
                __builtin_amdgcn_sched_barrier(0);
                const bf16x8 bf = join_halves(bl[U], bh[U]);
                acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks & 1][0], bf, acc[j][0], 0, 0, 0);
                acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks & 1][1], bf, acc[j][1], 0, 0, 0);
                // the step's LDS-DMA requests behind the first units' MFMAs (a request stalls the issuing wave ~100 cycles: in an MFMA's
                // shadow), the next request's source pointers behind units of the second half
                if constexpr (U <= NDY) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (more) {
                        if constexpr (U == 0) issue_x();
                        else issue_dy1(st_nxt, U - 1);
                    }
                }
                if constexpr (U == NU / 2 + 1 || U == NU / 2 + 3) {
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (U == NU / 2 + 1) prep_x();
                    else prep_dy();
                }
            };
            W8Unroll<0, NU>::run(unit);
            __builtin_amdgcn_sched_barrier(0);
            rp += 64;
            if (rp >= RX) rp -= RX;
            pend = PD == 2 ? n_issued : 0;                            // PD = 1: the next step needs what this one requested
            st_cur = st_cur + 1 > PD ? 0 : st_cur + 1;
            st_nxt = st_nxt + 1 > PD ? 0 : st_nxt + 1;
        }
#ifdef W3_TIMING
        const unsigned long long T2 = __builtin_readcyclecounter();
#endif
        // split-K partial tile -> workspace [slab][Cout][9 * Cin] (NCO = 2: two slabs per K range, one per pixel half; the deterministic
        // reduce of conv.hip adds them)
        const int NK = 9 * p.Cin;
        float* part = p.partial + ((int64_t)bz * (NCO == 2 ? 2 : 1) + kh) * p.Cout * NK;
#pragma unroll
        for (int j = 0; j < NT; j++) {
            const int kc = g.tap_of[5 * HH + j] * p.Cin + ci0 + 32 * cc + (lane & 31);
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const int co = i0 + 32 * (q0 + a) + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                    if (co < p.Cout) part[(int64_t)co * NK + kc] = acc[j][a][e];
                }
        }
#ifdef W3_TIMING
        if (tid == 0) {   // debug build only: timestamps into the tail of the slab workspace (tools/bench_wgrad.py reads them)
            unsigned long long* dbg = reinterpret_cast<unsigned long long*>(p.partial + (size_t)g.slabs * p.Cout * NK) + (size_t)blockIdx.x * 4;
            dbg[0] = t_wait; dbg[1] = T2 - T1; dbg[2] = t_bar; dbg[3] = nk;
        }
#endif
    };
    if (h == 0) body(std::integral_constant<int, 0>{});
    else body(std::integral_constant<int, 1>{});
}

// Geometry of the 8-wave form on top of a W3Geom already filled by w3_geometry (PWp, HPp, Mp, toff, tap_of, magic numbers): ring length,
// grid, split, LDS.  false: stay on the 4-wave kernels.
bool w8_geometry(const WgradParams& p, W3Geom& g)
{
    static const int on = getenv("RYOLO_W3_V8") ? atoi(getenv("RYOLO_W3_V8")) : 1;          // A/B knob: 0 = the 4-wave kernels of r03 / r04
    if (!on || p.Cin % 64) return false;
    const int nco = p.Cout <= 64 ? 2 : 4;
    static const int pd_max = getenv("RYOLO_W3_V8_PD") ? atoi(getenv("RYOLO_W3_V8_PD")) : 2;       // A/B knob: 1 = the first cut (one step of prefetch)
    int pd = pd_max >= 2 ? 2 : 1, rx = 0;
    unsigned lds = 0;
    for (; pd >= 1; pd--) {                                          // two steps of prefetch where the LDS has the room (every map of the 800 x 800 step but 400 x 400)
        const int need = 2 * (g.PWp + 1) + 209 + 64 * (pd - 1);      // two halos + this step + the pd steps in flight + alignment slack (conv3x3.hip)
        rx = (int)ry_cdiv(need, 64) * 64;
        lds = 2u * (unsigned)(rx + W8_MIRROR) * 64u + (unsigned)(pd + 1) * (unsigned)nco * 4352u;     // (dY stages: 4 slots of 1088 B per quarter)
        if (lds <= 160u * 1024u) break;
    }
    if (pd < 1) return false;
    // An 8-wave workgroup holds 2 x 224 of a SIMD's 512 registers: no wave of the main stream's matrix kernels fits beside it, the CU is this
    // workgroup's alone whatever its LDS share — so the ring may take the whole 160 KiB (W = 400: 1024 rows, 148 KiB), and the grid is sized
    // to HALF the chip: measured on the step (same box, alternating, img/s) 256 workgroups 879 (the 4-wave kernels at 256: 883), 192 888,
    // 160 904*, 128 894 / 908*, 96 898*, 64 891* (* = a faster box).  The side stream then owns 128 CUs at full speed and the main stream the
    // other 128 undisturbed, instead of both sharing every CU's issue slots, registers and LDS.
    static const int max_kib = getenv("RYOLO_W3_V8_LDS") ? atoi(getenv("RYOLO_W3_V8_LDS")) : 160;
    if (lds > (unsigned)max_kib * 1024u) return false;
    const int gx = (int)ry_cdiv(p.Cout, 128), gc = p.Cin / 64;
    static const int target = getenv("RYOLO_W3_V8_BLOCKS") ? atoi(getenv("RYOLO_W3_V8_BLOCKS")) : 96;
    int64_t sk = ry_cdiv(target, (int64_t)gx * gc);
    static const int minsteps = getenv("RYOLO_W3_MINSTEPS") ? atoi(getenv("RYOLO_W3_MINSTEPS")) : 24;
    const int64_t maxsplit = g.Mp / ((int64_t)minsteps * 32);
    if (sk > maxsplit) sk = maxsplit;
    if (sk < 1) sk = 1;
    static const bool force = getenv("RYOLO_W3_FORCE") != nullptr;
    if ((int64_t)gx * gc * sk < 48 && !force) return false;           // small problems: the generic kernel's finer tiles fill the chip better
    g.v8 = nco;
    g.pd = pd;
    g.co64 = nco == 2 ? 1 : 0;
    g.step64 = 1;
    g.mirror = 1;
    g.RX = rx;
    g.gx = gx;
    g.gc = gc;
    g.kchunk = ry_cdiv(ry_cdiv(g.Mp, sk), 64) * 64;
    g.splitk = (int)ry_cdiv(g.Mp, g.kchunk);
    g.slabs = g.splitk * (nco == 2 ? 2 : 1);
    g.lds_bytes = lds;
    return true;
}

int w8_launch(const WgradParams& p, const W3Geom& g, hipStream_t stream)
{
    static RyLdsAttr attr2, attr4;
    if (ry_max_dynamic_lds(attr2, reinterpret_cast<const void*>(&conv3x3_wgrad8_kernel<2>), 160 * 1024) ||
        ry_max_dynamic_lds(attr4, reinterpret_cast<const void*>(&conv3x3_wgrad8_kernel<4>), 160 * 1024))
        return RY_ERR_LAUNCH;
    const dim3 grid((unsigned)((int64_t)g.gx * g.gc * g.splitk));
    if (g.v8 == 2) hipLaunchKernelGGL((conv3x3_wgrad8_kernel<2>), grid, dim3(512), g.lds_bytes, stream, p, g);
    else hipLaunchKernelGGL((conv3x3_wgrad8_kernel<4>), grid, dim3(512), g.lds_bytes, stream, p, g);
    return hipGetLastError() == hipSuccess ? RY_OK : RY_ERR_LAUNCH;
}
