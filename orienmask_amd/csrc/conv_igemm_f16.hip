// Fused convolution for gfx950, fp16 activations / fp16 weights / fp32 accumulate (BASELINE.json configs[4],
// SURVEY.md section 8d "Config 5"): implicit GEMM on v_mfma_f32_32x32x16_f16.
//
//   out[m][n] = f16( act( (sum_k A[m][k] * Wt[n][k]) * scale[n] + shift[n] ) (+ res[m][n]) )
//   m = (b, oy, ox) output pixel, n = output channel, k = (kh, kw, ci); A, Wt, res, out fp16; sum, scale, shift fp32
//
// Same fusion as conv_igemm.hip (Conv2d -> BatchNorm2d(eval) -> LeakyReLU(0.1), residual add, nearest upsample,
// concat-by-slice: /root/reference/model/base.py:95-137, backbone/darknet.py:14-15, orienmask_yolo_fpnplus.py:78-86);
// the reference itself has no reduced-precision path, so this file's parity bar is its own oracle
// (oracle/orienmask_ref.py:forward_f16) plus agreement with the fp32 path at detection level.
//
// What changes against the f32 kernel, and why (MI355X):
//   * the f16 matrix pipe is 16x faster (2.5 PFLOP/s dense) while every byte rate stays where it was, so a k-step
//     (64 halfs = one 128-byte row per tile row, the same LDS image and XOR swizzle as the f32 kernel) is only
//     4*TM*TN MFMAs of 32 cycles.  The operand ring is therefore 3 deep with the LDS-DMA running two k-steps ahead
//     (counted vmcnt, raw s_barrier), and the big layers use a 256x128 tile with 128x64 per wave: 0.75 KiB of
//     ds_read_b128 per MFMA instead of 1 KiB, 12 DMA pieces per 32 MFMAs.
//   * one ds_read_b128 (8 halfs of a row) is exactly one MFMA operand: lanes 0-31 supply k 0..7, lanes 32-63 k 8..15.
//   * cin = 32 layers (conv2.0, conv2.1.conv.1) read TWO taps per k-step (chunks 0-3 = tap 2j, 4-7 = tap 2j+1; the
//     10th tap of a 3x3 is zero in both operands) so that conv1's output stays 32 channels wide in HBM.
//   * epilogue: C tile through LDS in fp32, then 8 channels per thread: one 16-byte fp16 store (and residual load).
//     The four head convolutions write fp32 (what the postprocess kernels read).
#include <cstdlib>

#include "om_common.h"

namespace om {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct IgemmHParams {
    const _Float16* in;
    const _Float16* w;
    const float* scale;
    const float* shift;
    const _Float16* res;
    void* out;            // fp16 (out_f32 = 0) or fp32 (out_f32 = 1)
    int* ticket;
    int H, W, cin, in_pix_stride;
    int Ho, Wo, HoWo, cout;
    int ks, stride, pad;
    int M, kc, ksteps, taps;      // PAIR: ksteps = (taps + 1) / 2, weights stored with taps rounded up to even
    int n_tiles, total_tiles;
    int leaky, res_pix_stride, out_pix_stride, out_mode, up, out_f32;
    int vec_io;
    int total_in_pixels;
    int w_bytes;
};

template <int BM, int BN, int WM, int WN, bool PAIR>
__global__ __launch_bounds__(256) void conv_igemm_f16_kernel(const IgemmHParams p) {
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int NWN = BN / WN;
    constexpr int A_CH = BM / 32, B_CH = BN / 32, NP = A_CH + B_CH;
    constexpr int NBUF = 3;
    constexpr int STAGE = (BM + BN) * 8;          // f32x4 (16-byte) units per ring stage
    constexpr int PPS = (NP + 2) / 3;             // DMA pieces issued in each of the first three slots of a k-step
    constexpr int CH8 = BN / 8;                   // 8-channel chunks per C-tile row
    constexpr int RP = 256 / CH8;                 // C-tile rows per epilogue pass
    constexpr int CH = BN / 4;                    // f32x4 chunks per C-tile row
    static_assert((BM / WM) * (BN / WN) == 4, "four waves per workgroup");
    static_assert(BM * BN / 4 <= NBUF * STAGE, "fp32 C tile must fit in the operand ring");
    __shared__ f32x4 smem[NBUF * STAGE + 1];      // ONE LDS object (see conv_igemm.hip)
    int* const s_ticket = reinterpret_cast<int*>(smem + NBUF * STAGE);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave / NWN, wn = wave % NWN;
    const int lrow = tid >> 3, lcol = tid & 7;
    const int scol = lcol ^ ((lrow >> 1) & 7);     // logical 16-byte chunk this lane fetches (LDS stays lane-linear)
    const int fi = lane & 31, fk = lane >> 5;
    const int fsw = (fi >> 1) & 7;

    for (;;) {
        int tile;
        if (p.ticket) {
            if (tid == 0) *s_ticket = atomicAdd(p.ticket, 1);
            __syncthreads();
            tile = *s_ticket;
        } else {
            tile = blockIdx.x;
        }
        if (tile >= p.total_tiles) break;
        tile = __builtin_amdgcn_readfirstlane(tile);
        const int tile_n = tile % p.n_tiles;
        const int tile_m = tile / p.n_tiles;
        const int m0 = tile_m * BM, n0 = tile_n * BN;

        // ---- loader role: thread -> (row lrow + 32*j, chunk lcol); offsets relative to the tile's first image
        int pixbase[A_CH], iy0[A_CH], ix0[A_CH];
        unsigned mokmask = 0;
        const int b_first = (m0 < p.M ? m0 : p.M - 1) / p.HoWo;
#pragma unroll
        for (int j = 0; j < A_CH; ++j) {
            int m = m0 + lrow + 32 * j;
            const bool ok = m < p.M;
            if (!ok) m = p.M - 1;
            const int b = m / p.HoWo;
            const int rr = m - b * p.HoWo;
            const int oy = rr / p.Wo;
            const int ox = rr - oy * p.Wo;
            pixbase[j] = (b - b_first) * p.H * p.W;
            iy0[j] = oy * p.stride - p.pad;
            ix0[j] = ox * p.stride - p.pad;
            mokmask |= (ok ? 1u : 0u) << j;
        }
        const _Float16* in_base = p.in + (size_t)b_first * p.H * p.W * p.in_pix_stride;
        const size_t in_left = ((size_t)p.total_in_pixels - (size_t)b_first * p.H * p.W) * p.in_pix_stride * 2;
        const auto rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(in_base), 0,
                                                             in_left < 0x7FFFFFFFull ? (int)in_left : 0x7FFFFFFF, 0x00020000);
        const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.w), 0, p.w_bytes, 0x00020000);

        int n_kh = 0, n_kw = 0, n_cc = 0;          // step being fetched: (tap row, tap col, 64-channel chunk); PAIR: n_cc = tap pair
        auto advance = [&]() {
            if constexpr (PAIR) {
                ++n_cc;
            } else {
                if (++n_cc == p.kc) {
                    n_cc = 0;
                    if (++n_kw == p.ks) { n_kw = 0; ++n_kh; }
                }
            }
        };
        auto issue_piece = [&](int piece, int buf, bool live) {
            f32x4* dst = smem + buf * STAGE + wave_u * 64;
            if (piece < A_CH) {
                const int j = piece;
                int kh, kw, coff;
                bool tap_ok = true;
                if constexpr (PAIR) {
                    const int tap = 2 * n_cc + (scol >> 2);
                    kh = (tap * 11) >> 5;          // tap / 3 for tap < 12
                    kw = tap - 3 * kh;
                    coff = (scol & 3) * 8;
                    tap_ok = tap < p.taps;
                } else {
                    kh = n_kh; kw = n_kw;
                    coff = n_cc * 64 + scol * 8;
                }
                const int iy = iy0[j] + kh, ix = ix0[j] + kw;
                const bool ok = live & tap_ok & (((mokmask >> j) & 1u) != 0) & ((unsigned)iy < (unsigned)p.H) &
                                ((unsigned)ix < (unsigned)p.W);
                const int voff = (((pixbase[j] + iy * p.W + ix) * p.in_pix_stride + coff) * 2) | (ok ? 0 : (int)0x80000000);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr_t)(dst + j * 256), 16, voff, 0, 0, 0);
            } else {
                const int j = piece - A_CH;
                int koff;
                if constexpr (PAIR) koff = n_cc * 64 + scol * 8;
                else koff = (n_kh * p.ks + n_kw) * p.cin + n_cc * 64 + scol * 8;
                const int row_halfs = PAIR ? ((p.taps + 1) / 2) * 64 : p.taps * p.cin;
                const int voff = (((n0 + lrow + 32 * j) * row_halfs + koff) * 2) | (live ? 0 : (int)0x80000000);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(dst + BM * 8 + j * 256), 16, voff, 0, 0, 0);
            }
        };

        f32x16 acc[TM][TN];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

        const f32x4* fragA = smem + (wm * WM + fi) * 8;
        const f32x4* fragB = smem + BM * 8 + (wn * WN + fi) * 8;
        f32x4 ca[TM], cb[TN], na[TM], nb[TN];
        auto read_frags = [&](f32x4(&fa)[TM], f32x4(&fb)[TN], int buf, int q) {
            const int ch = (2 * q + fk) ^ fsw;
            const int bo = buf * STAGE;
#pragma unroll
            for (int a = 0; a < TM; ++a) fa[a] = fragA[bo + a * 32 * 8 + ch];
#pragma unroll
            for (int b = 0; b < TN; ++b) fb[b] = fragB[bo + b * 32 * 8 + ch];
        };

        // prologue: steps 0 and 1 in flight, step 0 waited for
#pragma unroll
        for (int piece = 0; piece < NP; ++piece) issue_piece(piece, 0, true);
        advance();
#pragma unroll
        for (int piece = 0; piece < NP; ++piece) issue_piece(piece, 1, 1 < p.ksteps);
        advance();
        asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NP) : "memory");
        __builtin_amdgcn_s_barrier();
        read_frags(ca, cb, 0, 0);
        int buf = 0;
        for (int s = 0; s < p.ksteps; ++s) {
            const int buf1 = buf == NBUF - 1 ? 0 : buf + 1;
            const int buf2 = buf1 == NBUF - 1 ? 0 : buf1 + 1;
            const bool live2 = s + 2 < p.ksteps;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        // weights first: D[i = channel][j = pixel]
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                            __builtin_bit_cast(f16x8, cb[b]), __builtin_bit_cast(f16x8, ca[a]), acc[a][b], 0, 0, 0);
                if (q < 3) {
                    read_frags(na, nb, buf, q + 1);
#pragma unroll
                    for (int i = 0; i < PPS; ++i)
                        if (q * PPS + i < NP) issue_piece(q * PPS + i, buf2, live2);
                } else {
                    read_frags(na, nb, buf1, 0);
                }
                if (q == 2) {
                    // operands of step s+1 have landed (everything older than this step's NP pieces) and my reads of
                    // the current buffer are done -> the ring slot two steps back may be overwritten after the barrier
                    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"i"(NP) : "memory");
                    __builtin_amdgcn_s_barrier();
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int a = 0; a < TM; ++a) ca[a] = na[a];
#pragma unroll
                for (int b = 0; b < TN; ++b) cb[b] = nb[b];
            }
            buf = buf1;
            advance();
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();

        // ---- epilogue, phase 1: accumulators -> LDS C tile [m][n] fp32, 16-byte chunk index swizzled with m & 7.
        // Transposed 32x32 D layout: pixel = lane & 31, channel = 8*(r>>2) + 4*(lane>>5) + (r&3).
        f32x4* sC = smem;
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            const int ml = wm * WM + a * 32 + fi;
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n4 = (wn * WN + b * 32) / 4 + 2 * g + fk;
                    f32x4 v = {acc[a][b][4 * g], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]};
                    sC[ml * CH + (n4 ^ (ml & 7))] = v;
                }
        }
        __syncthreads();

        // ---- phase 2: 8 channels per thread
        if (p.out_mode != 2) {
            const int n8 = tid % CH8, r0 = tid / CH8;
            const int n = n0 + n8 * 8;
            float sc[8], sh[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { sc[k] = p.scale[n + k]; sh[k] = p.shift[n + k]; }   // scale/shift padded to cout_pad
            const int nvalid = p.cout - n;
            const bool vec = p.vec_io && nvalid >= 8;
#pragma unroll 2
            for (int ps = 0; ps < BM / RP; ++ps) {
                const int ml = ps * RP + r0;
                const int m = m0 + ml;
                if (m >= p.M || nvalid <= 0) continue;
                const f32x4 v0 = sC[ml * CH + ((2 * n8) ^ (ml & 7))];
                const f32x4 v1 = sC[ml * CH + ((2 * n8 + 1) ^ (ml & 7))];
                float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float t = fmaf(v[k], sc[k], sh[k]);
                    v[k] = p.leaky ? (t > 0.f ? t : t * 0.1f) : t;
                }
                if (p.out_f32) {
                    float* o = static_cast<float*>(p.out) + (size_t)m * p.out_pix_stride + n;
                    if (vec) {
                        *reinterpret_cast<f32x4*>(o) = f32x4{v[0], v[1], v[2], v[3]};
                        *reinterpret_cast<f32x4*>(o + 4) = f32x4{v[4], v[5], v[6], v[7]};
                    } else {
                        for (int k = 0; k < 8 && k < nvalid; ++k) o[k] = v[k];
                    }
                    continue;
                }
                _Float16* const outh = static_cast<_Float16*>(p.out);
                if (p.out_mode == 0) {
                    _Float16* o = outh + (size_t)m * p.out_pix_stride + n;
                    if (p.res) {
                        const _Float16* rp = p.res + (size_t)m * p.res_pix_stride + n;
                        if (vec) {
                            const f16x8 rv = *reinterpret_cast<const f16x8*>(rp);
#pragma unroll
                            for (int k = 0; k < 8; ++k) v[k] += (float)rv[k];
                        } else {
                            for (int k = 0; k < 8 && k < nvalid; ++k) v[k] += (float)rp[k];
                        }
                    }
                    if (vec) {
                        f16x8 hv;
#pragma unroll
                        for (int k = 0; k < 8; ++k) hv[k] = (_Float16)v[k];
                        *reinterpret_cast<f16x8*>(o) = hv;
                    } else {
                        for (int k = 0; k < 8 && k < nvalid; ++k) o[k] = (_Float16)v[k];
                    }
                } else {
                    f16x8 hv;
#pragma unroll
                    for (int k = 0; k < 8; ++k) hv[k] = (_Float16)v[k];
                    const int bi = m / p.HoWo;
                    const int rr = m - bi * p.HoWo;
                    const int oy = rr / p.Wo, ox = rr - oy * p.Wo;
                    const int Wu = p.Wo * p.up;
                    const size_t base = ((size_t)bi * p.Ho * p.up + (size_t)oy * p.up) * Wu + (size_t)ox * p.up;
                    for (int dy = 0; dy < p.up; ++dy)
                        for (int dx = 0; dx < p.up; ++dx) {
                            _Float16* o = outh + (base + (size_t)dy * Wu + dx) * p.out_pix_stride + n;
                            if (vec) *reinterpret_cast<f16x8*>(o) = hv;
                            else
                                for (int k = 0; k < 8 && k < nvalid; ++k) o[k] = hv[k];
                        }
                }
            }
        } else {
            // NCHW fp32 output (orientation head): consecutive threads walk pixels of one channel
            const float* sCf = reinterpret_cast<const float*>(smem);
            float* const outf = static_cast<float*>(p.out);
            const int nch = min(BN, p.cout - n0);
            for (int idx = tid; idx < nch * BM; idx += 256) {
                const int nl = idx / BM, ml = idx - nl * BM;
                const int m = m0 + ml;
                if (m >= p.M) continue;
                const int n = n0 + nl;
                float t = fmaf(sCf[(ml * CH + ((nl >> 2) ^ (ml & 7))) * 4 + (nl & 3)], p.scale[n], p.shift[n]);
                if (p.leaky) t = t > 0.f ? t : t * 0.1f;
                const int bi = m / p.HoWo;
                const int rr = m - bi * p.HoWo;
                outf[((size_t)bi * p.cout + n) * p.HoWo + rr] = t;
            }
        }
        if (!p.ticket) break;
        __syncthreads();      // the C tile is dead before the next tile's operands land in LDS
    }
}

struct TileChoiceH { int bm, bn; };

// Same cost model as the f32 kernel: ceil(tiles / CUs) tile-times weighted by how well the shape feeds the pipe.
static TileChoiceH choose_tile_f16(int M, int cout_pad) {
    struct Cand { int bm, bn; double eff; };
    const Cand cands[] = {{256, 128, 1.00}, {128, 128, 0.85}, {128, 64, 0.70}, {64, 64, 0.55}, {128, 32, 0.45}};
    TileChoiceH best{0, 0};
    double best_cost = 1e300;
    for (const Cand& c : cands) {
        if (cout_pad % c.bn) continue;
        const long long tiles = (long long)((M + c.bm - 1) / c.bm) * (cout_pad / c.bn);
        const long long rounds = (tiles + 255) / 256;
        const double cost = (double)rounds * c.bm * c.bn / c.eff;
        if (cost < best_cost) { best_cost = cost; best = TileChoiceH{c.bm, c.bn}; }
    }
    return best;
}

void conv_tile_for_f16(int M, int cout_pad, int cin, int* bm, int* bn) {
    if (cin == 32) { *bm = 128; *bn = cout_pad % 64 == 0 ? 64 : 32; return; }
    const TileChoiceH t = choose_tile_f16(M, cout_pad);
    *bm = t.bm; *bn = t.bn;
}

template <int BM, int BN, int WM, int WN, bool PAIR>
static int launch_tile_f16(IgemmHParams p, int cout_pad, int blocks_per_cu, hipStream_t stream) {
    const int m_tiles = (p.M + BM - 1) / BM;
    p.n_tiles = cout_pad / BN;
    const long long total = (long long)m_tiles * p.n_tiles;
    OM_REQUIRE(total > 0 && total < (1ll << 31), OM_EINVAL, "conv f16: %lld tiles out of range", total);
    p.total_tiles = (int)total;
    long long grid = total;
    if (p.ticket) grid = total < 256ll * blocks_per_cu ? total : 256ll * blocks_per_cu;
    hipLaunchKernelGGL((conv_igemm_f16_kernel<BM, BN, WM, WN, PAIR>), dim3((unsigned)grid), dim3(256), 0, stream, p);
    OM_CHECK_HIP(hipGetLastError());
    return OM_OK;
}

int launch_conv_igemm_f16(const ConvArgsH& a, hipStream_t stream) {
    OM_REQUIRE(a.in && a.w && a.scale && a.shift && a.out, OM_EINVAL, "conv f16: null pointer");
    OM_REQUIRE(a.cin == 32 || (a.cin % 64 == 0 && a.cin >= 64), OM_EINVAL, "conv f16: cin=%d must be 32 or a multiple of 64", a.cin);
    OM_REQUIRE(a.ks == 1 || a.ks == 3, OM_EINVAL, "conv f16: ksize=%d not supported", a.ks);
    OM_REQUIRE(a.stride == 1 || a.stride == 2, OM_EINVAL, "conv f16: stride=%d not supported", a.stride);
    OM_REQUIRE(a.in_pix_stride % 8 == 0 && (reinterpret_cast<uintptr_t>(a.in) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(a.w) & 15) == 0,
               OM_EINVAL, "conv f16: input view / weights must be 16-byte aligned");
    OM_REQUIRE(a.cout_pad % 32 == 0 && a.cout <= a.cout_pad, OM_EINVAL, "conv f16: cout_pad=%d", a.cout_pad);
    OM_REQUIRE((long long)a.B * a.H * a.W < (1ll << 31) && (long long)a.B * a.Ho * a.Wo < (1ll << 31), OM_EINVAL,
               "conv f16: more than 2^31 pixels");
    OM_REQUIRE(!(a.res && (a.out_mode != 0 || a.out_f32)), OM_EINVAL, "conv f16: residual only with plain NHWC fp16 output");
    OM_REQUIRE(!(a.out_mode == 2 && !a.out_f32) && !(a.out_mode == 1 && a.out_f32), OM_EINVAL,
               "conv f16: NCHW output is fp32, up-sampled output is fp16");
    const bool pair = a.cin == 32;
    IgemmHParams p;
    p.in = static_cast<const _Float16*>(a.in); p.w = static_cast<const _Float16*>(a.w);
    p.scale = a.scale; p.shift = a.shift; p.res = static_cast<const _Float16*>(a.res); p.out = a.out;
    p.ticket = a.ticket;
    p.H = a.H; p.W = a.W; p.cin = a.cin; p.in_pix_stride = a.in_pix_stride;
    p.Ho = a.Ho; p.Wo = a.Wo; p.HoWo = a.Ho * a.Wo; p.cout = a.cout;
    p.ks = a.ks; p.stride = a.stride; p.pad = a.ks / 2;
    p.M = a.B * a.Ho * a.Wo; p.taps = a.ks * a.ks;
    p.kc = pair ? 1 : a.cin / 64;
    p.ksteps = pair ? (p.taps + 1) / 2 : p.taps * p.kc;
    p.leaky = a.leaky; p.res_pix_stride = a.res_pix_stride; p.out_pix_stride = a.out_pix_stride;
    p.out_mode = a.out_mode; p.up = a.up; p.out_f32 = a.out_f32;
    p.n_tiles = 0; p.total_tiles = 0;
    p.total_in_pixels = a.B * a.H * a.W;
    p.w_bytes = (int)(conv_f16_weight_halfs(a.cout_pad, a.ks, a.cin) * 2);
    const int esz = a.out_f32 ? 4 : 2;
    p.vec_io = (a.out_mode != 2 && (a.out_pix_stride * esz) % 16 == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0 &&
                (!a.res || (a.res_pix_stride % 8 == 0 && (reinterpret_cast<uintptr_t>(a.res) & 15) == 0)))
                   ? 1 : 0;
    int bm, bn;
    conv_tile_for_f16(p.M, a.cout_pad, a.cin, &bm, &bn);
    static const int force_tile = [] { const char* e = getenv("OM_CONV16_TILE"); return e ? atoi(e) : 0; }();   // e.g. 128128
    if (force_tile > 0 && !pair && a.cout_pad % (force_tile % 1000) == 0) { bm = force_tile / 1000; bn = force_tile % 1000; }
    if (pair) {
        if (bn == 64) return launch_tile_f16<128, 64, 64, 32, true>(p, a.cout_pad, 2, stream);
        return launch_tile_f16<128, 32, 32, 32, true>(p, a.cout_pad, 2, stream);
    }
    if (bm == 256 && bn == 128) return launch_tile_f16<256, 128, 128, 64, false>(p, a.cout_pad, 1, stream);
    if (bm == 128 && bn == 128) return launch_tile_f16<128, 128, 64, 64, false>(p, a.cout_pad, 1, stream);
    if (bm == 128 && bn == 64) return launch_tile_f16<128, 64, 64, 32, false>(p, a.cout_pad, 2, stream);
    if (bm == 64 && bn == 64) return launch_tile_f16<64, 64, 32, 32, false>(p, a.cout_pad, 3, stream);
    return launch_tile_f16<128, 32, 32, 32, false>(p, a.cout_pad, 2, stream);
}

}  // namespace om
