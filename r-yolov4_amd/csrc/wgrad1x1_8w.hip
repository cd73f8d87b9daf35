// Pointwise (1x1, stride 1) WEIGHT GRADIENT on 256 x 256 tiles with 8 waves — gfx950 only.
// Reference rows served: SURVEY.md §8a M1 (`Conv`, model/utils.py:6-32: autograd of nn.Conv2d(k = 1) w.r.t. its weight) for the layers with
// Cin >= 256 and Cout > 128 (30 of yolov7's 40 pointwise weight gradients, ~4.9 of the 5.4 ms the class takes alone at batch 64).
//
// dW[co][ci] = sum_p dY[p][co] * X[p][ci].  wgrad1x1_dma_kernel (conv.hip) gives a 4-wave workgroup a 128 x 128 tile: per 64-pixel K step it
// streams 32 KiB through L2 -> LDS for 64 MFMAs, each wave reads 1 KiB of fragments per MFMA, and dY crosses L2 Cin / 128 times (X: Cout /
// 128 times).  The class ran at 570 TF/s = 23 % of the matrix peak and 37 % of the HBM peak: bound by neither, by its L2 -> LDS stream.
// Here (the shape of conv3x3_wgrad8.hip and gemm256.hip): ONE 8-wave workgroup per CU owns 256 output x 256 input channels; wave (wm, wn)
// of a 2 x 4 grid owns 128 x 64 = 4 x 2 accumulator tiles: per 16-pixel slice 4 + 2 fragments for 8 MFMAs (0.75 KiB per MFMA), per 64-pixel
// step 64 KiB for 256 MFMAs (half the L2 bytes and half the LDS-DMA requests per MFMA), every operand byte crosses L2 Cin / 256 (Cout / 256)
// times.  Two 64-KiB stages ([64 px][512 B] of dY, then of X), both requested one step ahead; a workgroup holds 2 x ~200 registers per SIMD
// lane, so its CU is its own and the grid is sized to a PART of the chip (conv3x3_wgrad8.hip: the side stream owns those CUs, the main
// stream's kernels the rest).
// LDS image: plain [pixel][512 B] rows, lane-linear LDS-DMA pieces of two whole rows; four consecutive rows of one 64-byte channel block
// would sit on the same banks (the bank pattern repeats every 256 B), so block e of row r is stored at position e ^ (r & 3) — the swizzle is
// applied on the SOURCE address of the request, the transposed fragment reads apply the same XOR (their rows keep r & 3 per lane).
// Split-K slabs [z][Cout][Cin] fp32 + the deterministic reduce of conv.hip.
#include "conv_internal.h"
#include <stdlib.h>
#include <type_traits>

extern __shared__ __attribute__((aligned(1024))) unsigned char w1x8_lds[];

template <int K> __device__ __forceinline__ void w1x8_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K) : "memory"); }
template <int N> __device__ __forceinline__ void w1x8_wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }

struct W1x8Geom {
    int gx, gy, splitk;
    int64_t kchunk;
};

__global__ __launch_bounds__(512, 1) void wgrad1x1_8w_kernel(const WgradParams p, const W1x8Geom g)
{
    constexpr int OPB = 64 * 512;                                    // bytes of one operand in a stage: [64 px][512 B]
    constexpr int STB = 2 * OPB;                                     // stage: dY then X
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;                         // 128 output channels x 64 input channels per wave
    const int64_t M = (int64_t)p.NB * p.OH * p.OW;
    const int t_id = xcd_remap(blockIdx.x, gridDim.x);
    const int bx = t_id % g.gx, by = (t_id / g.gx) % g.gy, bz = t_id / (g.gx * g.gy);
    const int i0 = bx * 256, j0 = by * 256;
    const int64_t kbeg = (int64_t)bz * g.kchunk;
    const int64_t kend = min(M, kbeg + g.kchunk);
    if (kbeg >= kend) return;
    const int nk = (int)((kend - kbeg + 63) >> 6);
    const int npix = (int)(kend - kbeg);

    // ---- LDS-DMA: 64 pieces of 1 KiB per stage (two whole 512-byte rows each), 8 per wave: waves 0-3 bring dY, waves 4-7 X.
    // lane -> (row rr = lane >> 5 of the piece, 64-byte position (lane >> 2) & 7, 16-byte slot lane & 3); piece k of an operand = rows 2k, 2k + 1.
    const int rr = lane >> 5, pos = (lane >> 2) & 7, slot = lane & 3;
    const bool is_x = wave >= 4;
    const int kp0 = 8 * (wave & 3);                                   // first piece of this wave inside its operand
    const bf16_t* const op = is_x ? p.X : p.dY;
    const int64_t ld = is_x ? p.ldX : p.ldY;
    const int c0 = is_x ? j0 : i0, climit = is_x ? p.Cin : p.CoutPad;
    // even pieces hold rows with r & 3 = rr, odd pieces rows with r & 3 = 2 + rr: two source columns per lane
    const int e_even = pos ^ rr, e_odd = pos ^ (2 + rr);
    const bool ok_even = c0 + 32 * e_even + slot * 8 < climit, ok_odd = c0 + 32 * e_odd + slot * 8 < climit;
    const int64_t row0 = 2 * kp0 + rr;                                // this lane's row (pixel relative to the step) in piece 0 of the wave
    const bf16_t* src_even = op + (kbeg + row0) * ld + c0 + 32 * e_even + slot * 8;
    const bf16_t* src_odd = op + (kbeg + row0 + 2) * ld + c0 + 32 * e_odd + slot * 8;
    const int64_t pair_step = 4 * ld, k_step = 64 * ld;
    int pix_l = (int)row0;                                            // pixel (relative to kbeg) of this lane's row in piece 0 of the next request
    unsigned char* const my_lds = w1x8_lds + (is_x ? OPB : 0) + kp0 * 1024;
    auto issue_piece = [&](int stage, int u) {                        // u = 0 .. 7 (compile-time at every call site)
        const bool odd = u & 1;
        const int pix = pix_l + 2 * u;
        const bool ok = (odd ? ok_odd : ok_even) && pix < npix;
        const bf16_t* s_ = ok ? (odd ? src_odd : src_even) + (u >> 1) * pair_step : p.zeros;
        __builtin_amdgcn_global_load_lds((gbl_void_t*)s_, (lds_void_t*)(my_lds + stage * STB + u * 1024), 16, 0, 0);
    };
    auto advance = [&]() { src_even += k_step; src_odd += k_step; pix_l += 64; };
#pragma unroll
    for (int u = 0; u < 8; u++) issue_piece(0, u);
    advance();

    // ---- fragments: lane -> pixel row (grp >> 1) * 8 + (s16 >> 2) [+ 16 ks, + 4 for the second read], channels 16 (grp & 1) + 4 (s16 & 3) .. + 3
    // of a 32-channel block; block e of row r at position e ^ (r & 3); row stride 512 B
    const int s16 = lane & 15, grp = lane >> 4;
    const unsigned fr_row = (unsigned)((grp >> 1) * 8 + (s16 >> 2));
    const unsigned fr_col = (unsigned)((16 * (grp & 1) + 4 * (s16 & 3)) * 2);
    unsigned fa[4], fb[2];
#pragma unroll
    for (int i = 0; i < 4; i++) fa[i] = lds_addr(w1x8_lds) + fr_row * 512u + (unsigned)(((4 * wm + i) ^ (int)(fr_row & 3u)) * 64) + fr_col;
#pragma unroll
    for (int j = 0; j < 2; j++) fb[j] = lds_addr(w1x8_lds) + (unsigned)OPB + fr_row * 512u + (unsigned)(((2 * wn + j) ^ (int)(fr_row & 3u)) * 64) + fr_col;
    // Layers narrower than the tile (Cout <= 128: the wm = 1 waves; Cin = 128: the wn >= 2 waves) keep the 8-wave shape: the waves without a
    // block still bring their share of the stage, the others compute — those layers are HBM-bound (<= 128 flop/B), the matrix rate of two or
    // four waves per CU covers their bytes, and they stay on the side stream's CUs instead of spreading thin over the chip.
    const bool active = (i0 + 128 * wm < p.Cout) && (j0 + 64 * wn < p.Cin);
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

    for (int s = 0; s < nk; s++) {
        w1x8_wait_vm<0>();                                            // everything this wave requested one step ago has landed
        __builtin_amdgcn_s_barrier();                                 // ... and everybody else's; step s - 1 fully consumed
        const bool more = s + 1 < nk;
        const unsigned sb = (unsigned)((s & 1) * STB);
        ry_s16x4 al[2][4], ah[2][4], bl[2][2], bh[2][2];              // [slice parity]: slice ks + 1 is read under the MFMAs of slice ks
        // a slice's 12 reads go out as 8 (dY blocks 0, 1 and both X blocks) + 4 (dY blocks 2, 3); lgkmcnt is a 4-bit counter: never more than
        // 12 of this wave's reads in flight
        auto read_first = [&](auto kc) {
            constexpr int ks = decltype(kc)::value;
            constexpr unsigned off = (unsigned)(ks * 16 * 512);
#pragma unroll
            for (int i = 0; i < 2; i++) {
                al[ks & 1][i] = lds_tr16_off<off>(fa[i] + sb);
                ah[ks & 1][i] = lds_tr16_off<off + 2048>(fa[i] + sb);
            }
#pragma unroll
            for (int j = 0; j < 2; j++) {
                bl[ks & 1][j] = lds_tr16_off<off>(fb[j] + sb);
                bh[ks & 1][j] = lds_tr16_off<off + 2048>(fb[j] + sb);
            }
        };
        auto read_second = [&](auto kc) {
            constexpr int ks = decltype(kc)::value;
            constexpr unsigned off = (unsigned)(ks * 16 * 512);
#pragma unroll
            for (int i = 2; i < 4; i++) {
                al[ks & 1][i] = lds_tr16_off<off>(fa[i] + sb);
                ah[ks & 1][i] = lds_tr16_off<off + 2048>(fa[i] + sb);
            }
        };
        if (active) {
            read_first(std::integral_constant<int, 0>{});
            read_second(std::integral_constant<int, 0>{});
        }
        auto slice = [&](auto kc) {
            constexpr int ks = decltype(kc)::value;
            constexpr bool last = ks == 3;
            __builtin_amdgcn_sched_barrier(0);
            // in flight here: the 12 reads of slice ks.  First 8 landed <=> at most 4 outstanding.
            w1x8_wait_lgkm<4>();
            asm volatile("" : "+v"(al[ks & 1][0]), "+v"(ah[ks & 1][0]), "+v"(al[ks & 1][1]), "+v"(ah[ks & 1][1]));
            asm volatile("" : "+v"(bl[ks & 1][0]), "+v"(bh[ks & 1][0]), "+v"(bl[ks & 1][1]), "+v"(bh[ks & 1][1]));
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!last) read_first(std::integral_constant<int, ks + 1>{});          // 4 + 8 in flight
            __builtin_amdgcn_sched_barrier(0);
            bf16x8 bf[2];
#pragma unroll
            for (int j = 0; j < 2; j++) bf[j] = join_halves(bl[ks & 1][j], bh[ks & 1][j]);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (i == 2) {
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (!last) w1x8_wait_lgkm<8>(); else w1x8_wait_lgkm<0>();     // dY blocks 2, 3 of slice ks (issued before the 8 of slice ks + 1)
                    asm volatile("" : "+v"(al[ks & 1][2]), "+v"(ah[ks & 1][2]), "+v"(al[ks & 1][3]), "+v"(ah[ks & 1][3]));
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (!last) read_second(std::integral_constant<int, ks + 1>{}); // 8 + 4 in flight
                    __builtin_amdgcn_sched_barrier(0);
                }
                const bf16x8 af = join_halves(al[ks & 1][i], ah[ks & 1][i]);
#pragma unroll
                for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf[j], acc[i][j], 0, 0, 0);
                // two of the step's eight LDS-DMA requests per slice, behind the MFMAs of accumulator rows 0 and 2
                if (i == 0 || i == 2) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (more) issue_piece((s + 1) & 1, 2 * ks + (i >> 1));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        if (active) {
            slice(std::integral_constant<int, 0>{});
            slice(std::integral_constant<int, 1>{});
            slice(std::integral_constant<int, 2>{});
            slice(std::integral_constant<int, 3>{});
        } else if (more) {                                            // a wave whose 128 x 64 block lies outside the layer only moves data
#pragma unroll
            for (int u = 0; u < 8; u++) issue_piece((s + 1) & 1, u);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (more) advance();
    }
    // split-K partial tile -> workspace [z][Cout][Cin] fp32
    float* part = p.partial + (int64_t)bz * p.Cout * p.Cin;
    if (!active) return;
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int ci = j0 + 64 * wn + 32 * j + (lane & 31);
        if (ci >= p.Cin) continue;
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int co = i0 + 128 * wm + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (co < p.Cout) part[(int64_t)co * p.Cin + ci] = acc[i][j][e];
            }
    }
}

// eligibility + grid of the 8-wave pointwise form; *splitk / *kchunk as ryolo_conv_wgrad_plan reports them.  false: the 4-wave kernels of conv.hip.
bool w1x8_geometry(const WgradParams& p, int* splitk, int64_t* kchunk, int* gx_out, int* gy_out)
{
    static const int on = getenv("RYOLO_WGRAD_8W") ? atoi(getenv("RYOLO_WGRAD_8W")) : 1;       // A/B knob
    if (!on || !p.zeros) return false;
    if (p.ntaps != 1 || p.dh[0] != 0 || p.dw[0] != 0 || p.sh != 1 || p.sw != 1 || p.IH != p.OH || p.IW != p.OW) return false;
    // Cin >= 256 and Cout > 128 by default.  Narrower layers (down to 128 channels: RYOLO_WGRAD_8W_MINC=128) run correctly with idle waves (`active`
    // in the kernel; tests/test_gpu_wgrad1x1.py forces it) but cost the step 1 % (same box, three alternating runs: 905.9 vs 894.4 img/s): their
    // 4-wave launches are short, HBM-bound and overlap well as they are.
    static const int min_c = getenv("RYOLO_WGRAD_8W_MINC") ? atoi(getenv("RYOLO_WGRAD_8W_MINC")) : 256;
    if (p.Cin < min_c || p.Cin % 32 || p.Cout <= min_c / 2) return false;
    if (p.ldX % 8 || p.ldY % 8 || p.CoutPad % 8 || ((reinterpret_cast<uintptr_t>(p.dY) | reinterpret_cast<uintptr_t>(p.X)) & 15)) return false;
    const int64_t M = (int64_t)p.NB * p.OH * p.OW;
    if (M >= (1ll << 31) || M < 64) return false;
    const int gx = (int)ry_cdiv(p.Cout, 256), gy = (int)ry_cdiv(p.Cin, 256);
    static const int target = getenv("RYOLO_WGRAD_8W_BLOCKS") ? atoi(getenv("RYOLO_WGRAD_8W_BLOCKS")) : 96;
    int64_t sk = ry_cdiv(target, (int64_t)gx * gy);
    const int64_t maxsplit = ry_cdiv(M, 16 * 64);                     // at least 16 steps per split
    if (sk > maxsplit) sk = maxsplit;
    if (sk < 1) sk = 1;
    static const bool force = getenv("RYOLO_WGRAD_8W_FORCE") != nullptr;
    if ((int64_t)gx * gy * sk < 48 && !force) return false;
    const int64_t kc = ry_cdiv(ry_cdiv(M, sk), 64) * 64;
    *kchunk = kc;
    *splitk = (int)ry_cdiv(M, kc);
    *gx_out = gx;
    *gy_out = gy;
    return true;
}

int w1x8_launch(const WgradParams& p, hipStream_t stream)
{
    W1x8Geom g;
    if (!w1x8_geometry(p, &g.splitk, &g.kchunk, &g.gx, &g.gy)) return RY_ERR_ARG;
    static RyLdsAttr attr;
    if (ry_max_dynamic_lds(attr, reinterpret_cast<const void*>(&wgrad1x1_8w_kernel), 160 * 1024)) return RY_ERR_LAUNCH;
    const dim3 grid((unsigned)((int64_t)g.gx * g.gy * g.splitk));
    hipLaunchKernelGGL(wgrad1x1_8w_kernel, grid, dim3(512), 2 * 2 * 64 * 512, stream, p, g);
    return hipGetLastError() == hipSuccess ? RY_OK : RY_ERR_LAUNCH;
}
