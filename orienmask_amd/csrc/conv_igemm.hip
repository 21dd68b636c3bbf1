// Fused convolution for gfx950: implicit GEMM on the f32-input matrix cores.
//
//   out[m][n] = act( (sum_k A[m][k] * Wt[n][k]) * scale[n] + shift[n] ) (+ res[m][n])
//   m = (b, oy, ox) output pixel, n = output channel, k = (kh, kw, ci)
//
// Replaces, per layer, the reference's three separate ops Conv2d -> BatchNorm2d(eval) ->
// LeakyReLU(0.1) (/root/reference/model/base.py:104-137) plus, where present, the residual add
// of _DarkNetBlock (/root/reference/model/backbone/darknet.py:14-15), the nearest upsample of
// a route (/root/reference/model/base.py:95-101, written replicated into the concat buffer) and
// the torch.cat of /root/reference/model/orienmask_yolo_fpnplus.py:78-86 (producers write
// straight into channel slices of the concatenated NHWC buffer, consumers read strided views).
//
// Design (MI355X):
//   * NHWC activations, weights stored [cout][kh*kw][cin]: for a fixed tap both operands are
//     k-contiguous, so every global load is a full 128-byte line (32 floats) per tile row.
//   * v_mfma_f32_32x32x2_f32: exact fp32 (an fmaf chain), 64 cycles per instruction per SIMD,
//     157 TFLOP/s peak.  One wave owns a (WM x WN) output tile = TM x TN MFMA tiles.
//     The two k-slots of the instruction are fed from lanes 0-31 / 32-63; each lane reads a
//     16-byte k-quad of its row, so one ds_read_b128 per operand feeds four MFMAs.  (Which k
//     goes to which slot is a free permutation as long as A and B agree.)  The weights are the
//     FIRST operand, so the accumulator holds the tile transposed: a lane owns one pixel and
//     runs of 4 consecutive channels -> 16-byte epilogue accesses.
//   * K loop = taps x cin/32.  Register-staged double buffer: the loads of step s+1 are issued
//     before the 16*TM*TN MFMAs of step s (>= 2048 matrix-pipe cycles per wave: HBM/L2 latency is
//     hidden), written to the other LDS buffer after them; one barrier per step.
//   * LDS rows are 128 B (32 floats); the 16-B chunk index is XOR-swizzled with (row>>1)&7 so
//     that the 16 rows a ds_read_b128 lane group touches land on 16 distinct bank quads.
//   * Epilogue through LDS: the C tile is written [m][n] (16-B chunks XOR-swizzled with m&7),
//     then every thread streams rows as float4: scale/shift/LeakyReLU, residual read and the
//     store are full 512-byte rows per 32 lanes.
//   * Persistent workgroups + one atomic ticket counter per launch: tiles are handed out
//     dynamically, so every CU stays busy until the queue is empty (a static grid left the last
//     partial round at 2 workgroups per CU on a fraction of the chip: 57 % efficiency on the
//     17x17 layers).  Consecutive tickets share an A panel (N tiles fastest).
#include <cstdlib>

#include "om_common.h"

namespace om {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct IgemmParams {
    const float* in;
    const float* w;
    const float* scale;
    const float* shift;
    const float* res;
    float* out;
    int* ticket;          // zeroed before the launch; nullptr = one tile per workgroup (static grid)
    int H, W, cin, in_pix_stride;
    int Ho, Wo, HoWo, cout;
    int ks, stride, pad;
    int M, kc, ksteps, taps;
    int n_tiles, total_tiles;
    int leaky, res_pix_stride, out_pix_stride, out_mode, up;
    int vec_io;           // 1: out/res rows are 16-byte aligned (float4 epilogue)
    int total_in_pixels;  // B*H*W
    int w_bytes;          // cout_pad*taps*cin*4
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// Operands go global -> LDS directly (buffer_load_dwordx4 ... lds); zero padding comes from the buffer descriptor's bounds
// check (offset 0x80000000 is out of range -> zeros).
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_igemm_f32_kernel(const IgemmParams p) {
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int NWN = BN / WN;
    constexpr int A_CH = BM / 32;   // 16-byte chunks of the A tile each thread stages
    constexpr int B_CH = BN / 32;
    constexpr int CH = BN / 4;      // float4 chunks per C-tile row
    constexpr int RP = 256 / CH;    // C-tile rows per epilogue pass
    static_assert((BM / WM) * (BN / WN) == 4, "four waves per workgroup");
    static_assert(BM * BN <= 2 * (BM + BN) * 32, "C tile must fit in the operand buffers");
    // ONE LDS object (a second __shared__ variable makes hipcc drain the LDS-DMA queue before every
    // ds_read); the last 16 bytes hold the tile ticket.
    __shared__ f32x4 smem[2 * (BM + BN) * 8 + 1];
    int* const s_ticket = reinterpret_cast<int*>(smem + 2 * (BM + BN) * 8);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / NWN, wn = wave % NWN;
    const int lrow = tid >> 3, lcol = tid & 7;
    const int fi = lane & 31, fk = lane >> 5;
    const int fsw = (fi >> 1) & 7;

    for (;;) {
        int tile;
        if (p.ticket) {
            // lane 0's wave pays the ticket's round trip; the others wait at a raw barrier and do not drain the previous tile's
            // output stores (a __syncthreads() would make every wave wait for them: ~2 us per tile)
            if (tid == 0) *s_ticket = atomicAdd(p.ticket, 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            tile = *s_ticket;
        } else {
            // static grid: XCD-aware, bijective remap of the workgroup id (block b runs on XCD b % 8)
            const int nblk = gridDim.x, bid = blockIdx.x;
            const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, i = bid >> 3;
            tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
        }
        if (tile >= p.total_tiles) break;
        tile = __builtin_amdgcn_readfirstlane(tile);
        const int tile_n = tile % p.n_tiles;
        const int tile_m = tile / p.n_tiles;
        const int m0 = tile_m * BM, n0 = tile_n * BN;

        // ---- loader role: thread -> (row lrow + 32*j, 16-byte chunk lcol) ----------------
        int pixbase[A_CH], iy0[A_CH], ix0[A_CH];
        unsigned mokmask = 0;
        // offsets are relative to the first image the tile touches (keeps them < 2^31 bytes)
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
        // ---- descriptors and the issue of ONE 1-KiB piece (8 rows x 128 B) per call
        constexpr int NP = A_CH + B_CH;                     // pieces per wave per k-step
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);
        const float* in_base = p.in + (size_t)b_first * p.H * p.W * p.in_pix_stride;
        const int scol = lcol ^ ((lrow >> 1) & 7);          // logical chunk this lane fetches (LDS stays lane-linear)
        const size_t in_left = ((size_t)p.total_in_pixels - (size_t)b_first * p.H * p.W) * p.in_pix_stride * 4;
        const auto rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in_base), 0,
                                                             in_left < 0x7FFFFFFFull ? (int)in_left : 0x7FFFFFFF, 0x00020000);
        const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, p.w_bytes, 0x00020000);
        int n_kh = 0, n_kw = 0, n_cc = 0;                   // (tap row, tap col, cin chunk) of the step being fetched
        auto advance = [&]() {
            if (++n_cc == p.kc) {
                n_cc = 0;
                if (++n_kw == p.ks) { n_kw = 0; ++n_kh; }
            }
        };
        auto issue_piece = [&](int piece, int buf, bool live) {
            {
                const int coff = n_cc * 32 + scol * 4;
                f32x4* dst = smem + buf * (BM + BN) * 8 + wave_u * 64;
                if (piece < A_CH) {
                    const int j = piece;
                    const int iy = iy0[j] + n_kh, ix = ix0[j] + n_kw;
                    const bool ok = live & (((mokmask >> j) & 1u) != 0) & ((unsigned)iy < (unsigned)p.H) &
                                    ((unsigned)ix < (unsigned)p.W);      // bitwise: no branches around the loads
                    // branch-free: an invalid row just gets bit 31 set (>= num_records -> the DMA writes zeros)
                    const int voff = (((pixbase[j] + iy * p.W + ix) * p.in_pix_stride + coff) * 4) | (ok ? 0 : (int)0x80000000);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr_t)(dst + j * 256), 16, voff, 0, 0, 0);
                } else {
                    const int j = piece - A_CH;
                    const int tap = n_kh * p.ks + n_kw;
                    const int voff = ((((n0 + lrow + 32 * j) * p.taps + tap) * p.cin + coff) * 4) | (live ? 0 : (int)0x80000000);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(dst + BM * 8 + j * 256), 16, voff, 0, 0, 0);
                }
            }
        };

        f32x16 acc[TM][TN];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

        {
            // Software pipeline with ONE barrier per k-step and nothing outside the MFMA stream:
            //   slots 0..NP-1   : one DMA piece of step s+1 each
            //   slot (q, t=1)   : ds_reads of fragment group q+1
            //   end of slot 11  : barrier (this wave's DMA has landed, all reads of buffer s&1 are done)
            //   slot 12         : ds_reads of step s+1's group 0 from the other buffer
            // Inside a slot the address VALU, the DMA and the ds_reads are spread BETWEEN the slot's MFMAs
            // (sched_group_barrier): a 64-cycle f32 MFMA hides ~10 issue slots, but only if the other
            // instructions do not sit in one clump behind the last MFMA.
            const f32x4* fragA = smem + (wm * WM + fi) * 8;
            const f32x4* fragB = smem + BM * 8 + (wn * WN + fi) * 8;
            f32x4 ca[TM], cb[TN], na[TM], nb[TN];
            auto read_frags = [&](f32x4(&fa)[TM], f32x4(&fb)[TN], int buf, int q) {
                const int ch = (2 * q + fk) ^ fsw;
                const int bo = buf * (BM + BN) * 8;
#pragma unroll
                for (int a = 0; a < TM; ++a) fa[a] = fragA[bo + a * 32 * 8 + ch];
#pragma unroll
                for (int b = 0; b < TN; ++b) fb[b] = fragB[bo + b * 32 * 8 + ch];
            };
#pragma unroll
            for (int piece = 0; piece < NP; ++piece) issue_piece(piece, 0, true);
            advance();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            read_frags(ca, cb, 0, 0);
            for (int s = 0; s < p.ksteps; ++s) {
                const int buf = s & 1;
                const bool live = s + 1 < p.ksteps;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int slot = q * 4 + t;
#pragma unroll
                        for (int a = 0; a < TM; ++a)
#pragma unroll
                            for (int b = 0; b < TN; ++b)
                                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(cb[b][t], ca[a][t], acc[a][b], 0, 0, 0);
                        if (t == 1 && q < 3) read_frags(na, nb, buf, q + 1);
                        if (slot < NP) issue_piece(slot, buf ^ 1, live);
                        if (slot == 12) read_frags(na, nb, buf ^ 1, 0);
                        // interleave: MFMA, a few VALU/SALU, MFMA, ... then the DMA and the LDS reads
#pragma unroll
                        for (int i = 0; i < TM * TN; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                      // 1 MFMA
                            __builtin_amdgcn_sched_group_barrier(0x006, (16 + TM * TN - 1) / (TM * TN), 0);  // VALU|SALU
                            if (i == TM * TN - 2 || TM * TN == 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // VMEM read
                        }
                        __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);                    // DS reads
                        if (slot == 11) {
                            // the DMA of step s+1 must have landed before anyone reads it (do not rely on the
                            // compiler to drain an LDS-DMA queue at a barrier)
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            __syncthreads();
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int a = 0; a < TM; ++a) ca[a] = na[a];
#pragma unroll
                    for (int b = 0; b < TN; ++b) cb[b] = nb[b];
                }
                advance();
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }

        // ---- epilogue, phase 1: accumulators -> LDS C tile [m][n], chunk index swizzled with m & 7.
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

        // ---- phase 2: stream rows out as float4 (scale/shift, LeakyReLU, residual, store)
        if (p.out_mode != 2) {
            const int n4 = tid % CH, r0 = tid / CH;
            const int n = n0 + n4 * 4;
            const f32x4 sc = *reinterpret_cast<const f32x4*>(p.scale + n);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(p.shift + n);
            const int nvalid = p.cout - n;       // >= 4: whole chunk valid
#pragma unroll 4
            for (int ps = 0; ps < BM / RP; ++ps) {
                const int ml = ps * RP + r0;
                const int m = m0 + ml;
                if (m >= p.M || nvalid <= 0) continue;
                f32x4 v = sC[ml * CH + (n4 ^ (ml & 7))];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float t = fmaf(v[k], sc[k], sh[k]);
                    v[k] = p.leaky ? (t > 0.f ? t : t * 0.1f) : t;
                }
                const bool vec = p.vec_io && nvalid >= 4;
                if (p.out_mode == 0) {
                    float* o = p.out + (size_t)m * p.out_pix_stride + n;
                    if (p.res) {
                        const float* rp = p.res + (size_t)m * p.res_pix_stride + n;
                        if (vec) {
                            const f32x4 rv = *reinterpret_cast<const f32x4*>(rp);
                            v += rv;
                        } else {
                            for (int k = 0; k < 4 && k < nvalid; ++k) v[k] += rp[k];
                        }
                    }
                    if (vec) *reinterpret_cast<f32x4*>(o) = v;
                    else
                        for (int k = 0; k < 4 && k < nvalid; ++k) o[k] = v[k];
                } else {
                    const int bi = m / p.HoWo;
                    const int rr = m - bi * p.HoWo;
                    const int oy = rr / p.Wo, ox = rr - oy * p.Wo;
                    const int Wu = p.Wo * p.up;
                    const size_t base = ((size_t)bi * p.Ho * p.up + (size_t)oy * p.up) * Wu + (size_t)ox * p.up;
                    for (int dy = 0; dy < p.up; ++dy)
                        for (int dx = 0; dx < p.up; ++dx) {
                            float* o = p.out + (base + (size_t)dy * Wu + dx) * p.out_pix_stride + n;
                            if (vec) *reinterpret_cast<f32x4*>(o) = v;
                            else
                                for (int k = 0; k < 4 && k < nvalid; ++k) o[k] = v[k];
                        }
                }
            }
        } else {
            // NCHW output (orientation head): consecutive threads walk pixels of one channel
            const float* sCf = reinterpret_cast<const float*>(smem);
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
                p.out[((size_t)bi * p.cout + n) * p.HoWo + rr] = t;
            }
        }
        if (!p.ticket) break;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the C tile is dead before the next tile's operands land in
        __builtin_amdgcn_s_barrier();                            // LDS; the stores keep flying
    }
}

struct TileChoice { int bm, bn; };

// Pick the tile with the lowest estimated time: ceil(tiles / CUs) tile-times (the ticket queue keeps
// CUs balanced to within one tile), weighted by how well each shape feeds the matrix pipe.
static TileChoice choose_tile(int M, int cout_pad) {
    struct Cand { int bm, bn; double eff; };
    const Cand cands[] = {{128, 128, 1.00}, {64, 128, 0.93}, {128, 64, 0.90}, {64, 64, 0.84}, {128, 32, 0.70}};
    const int num_cu = 256;
    TileChoice best{0, 0};
    double best_cost = 1e300;
    for (const Cand& c : cands) {
        if (cout_pad % c.bn) continue;
        const long long tiles = (long long)((M + c.bm - 1) / c.bm) * (cout_pad / c.bn);
        const long long rounds = (tiles + num_cu - 1) / num_cu;
        const double cost = (double)rounds * c.bm * c.bn / c.eff;
        if (cost < best_cost) { best_cost = cost; best = TileChoice{c.bm, c.bn}; }
    }
    return best;
}

template <int BM, int BN, int WM, int WN>
static int launch_tile(IgemmParams p, int cout_pad, int blocks_per_cu, hipStream_t stream) {
    const int m_tiles = (p.M + BM - 1) / BM;
    p.n_tiles = cout_pad / BN;
    const long long total = (long long)m_tiles * p.n_tiles;
    OM_REQUIRE(total > 0 && total < (1ll << 31), OM_EINVAL, "conv: %lld tiles out of range", total);
    p.total_tiles = (int)total;
    long long grid = total;
    if (p.ticket) grid = total < 256ll * blocks_per_cu ? total : 256ll * blocks_per_cu;
    hipLaunchKernelGGL((conv_igemm_f32_kernel<BM, BN, WM, WN>), dim3((unsigned)grid), dim3(256), 0, stream, p);
    OM_CHECK_HIP(hipGetLastError());
    return OM_OK;
}

void conv_tile_for(int M, int cout_pad, int* bm, int* bn) {
    const TileChoice t = choose_tile(M, cout_pad);
    *bm = t.bm;
    *bn = t.bn;
}

int launch_conv_igemm(const ConvArgs& a, hipStream_t stream) {
    OM_REQUIRE(a.in && a.w && a.scale && a.shift && a.out, OM_EINVAL, "conv: null pointer");
    OM_REQUIRE(a.cin % 32 == 0 && a.cin >= 32, OM_EINVAL, "conv: cin=%d must be a multiple of 32", a.cin);
    OM_REQUIRE(a.ks == 1 || a.ks == 3, OM_EINVAL, "conv: ksize=%d not supported", a.ks);
    OM_REQUIRE(a.stride == 1 || a.stride == 2, OM_EINVAL, "conv: stride=%d not supported", a.stride);
    OM_REQUIRE(a.in_pix_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(a.in) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(a.w) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.scale) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(a.shift) & 15) == 0,
               OM_EINVAL, "conv: input view / weights / scale / shift must be 16-byte aligned");
    OM_REQUIRE(a.cout_pad % 32 == 0 && a.cout <= a.cout_pad, OM_EINVAL, "conv: cout_pad=%d", a.cout_pad);
    OM_REQUIRE((long long)a.B * a.H * a.W < (1ll << 31) && (long long)a.B * a.Ho * a.Wo < (1ll << 31), OM_EINVAL,
               "conv: more than 2^31 pixels");
    OM_REQUIRE(!(a.res && a.out_mode != 0), OM_EINVAL, "conv: residual only with plain NHWC output");
    IgemmParams p;
    p.in = a.in; p.w = a.w; p.scale = a.scale; p.shift = a.shift; p.res = a.res; p.out = a.out;
    p.ticket = a.ticket;
    p.H = a.H; p.W = a.W; p.cin = a.cin; p.in_pix_stride = a.in_pix_stride;
    p.Ho = a.Ho; p.Wo = a.Wo; p.HoWo = a.Ho * a.Wo; p.cout = a.cout;
    p.ks = a.ks; p.stride = a.stride; p.pad = a.ks / 2;
    p.M = a.B * a.Ho * a.Wo; p.kc = a.cin / 32; p.taps = a.ks * a.ks; p.ksteps = p.taps * p.kc;
    p.leaky = a.leaky; p.res_pix_stride = a.res_pix_stride; p.out_pix_stride = a.out_pix_stride;
    p.out_mode = a.out_mode; p.up = a.up;
    p.n_tiles = 0; p.total_tiles = 0;
    p.total_in_pixels = a.B * a.H * a.W;
    p.w_bytes = a.cout_pad * p.taps * a.cin * 4;
    p.vec_io = (a.out_mode != 2 && a.out_pix_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0 &&
                (!a.res || (a.res_pix_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(a.res) & 15) == 0)))
                   ? 1 : 0;
    const TileChoice t = choose_tile(p.M, a.cout_pad);
    if (t.bm == 128 && t.bn == 128) return launch_tile<128, 128, 64, 64>(p, a.cout_pad, 2, stream);
    if (t.bm == 64 && t.bn == 128) return launch_tile<64, 128, 32, 64>(p, a.cout_pad, 3, stream);
    if (t.bm == 128 && t.bn == 64) return launch_tile<128, 64, 64, 32>(p, a.cout_pad, 3, stream);
    if (t.bm == 64 && t.bn == 64) return launch_tile<64, 64, 32, 32>(p, a.cout_pad, 4, stream);
    return launch_tile<128, 32, 32, 32>(p, a.cout_pad, 4, stream);
}

}  // namespace om
