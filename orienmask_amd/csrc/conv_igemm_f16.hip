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
//   * the f16 matrix pipe is 16x faster (2.5 PFLOP/s dense) while every byte rate and every per-tile fixed cost stays
//     where it was.  Measured on the first version (one 8-wave workgroup per CU, 144 KiB ring): with the DMA, the LDS
//     reads AND the barrier removed from the k-loop the big layers still ran at 35 % of peak -- the tile prologue
//     (ticket, address set-up, first operands from HBM) and the LDS-staged epilogue ran with nothing to overlap them.
//     So: k-step = 32 halfs = 64-byte LDS rows, which makes a 3-deep operand ring small enough (72 KiB for the
//     256x128 tile, 48 KiB for 128x128) for TWO or THREE workgroups per CU; one workgroup's prologue/epilogue hides
//     behind the others' MFMAs, as in the f32 kernels.
//   * the LDS-DMA runs two k-steps ahead (counted vmcnt, raw s_barrier); a wave instruction lands 16 rows x 64 B.
//     DMA addressing is 3 VALU instructions per piece: a per-row byte offset and a per-row "tap is padding" bit mask
//     are computed once per tile; a k-step adds one scalar (tap, channel chunk) offset and moves the tap's mask bit
//     to bit 31 of the offset (>= num_records -> the DMA writes zeros).
//   * LDS rows are 4 x 16 B; the chunk index is XOR-swizzled with (row>>2)&3 (on the DMA source and on the read), so
//     the 16 rows of a ds_read_b128 lane group hit 16 distinct (row&3, chunk) bank quads.
//   * one ds_read_b128 (8 halfs of a row) is exactly one MFMA operand: lanes 0-31 supply k 0..7, lanes 32-63 k 8..15;
//     a k-step is two such k-slices.  256x128 tile: 128x64 per wave, 0.75 KiB of LDS reads per MFMA.
//   * tile boundaries: measured on the 136^2 128->256 layer (0.43 ms) the k-loop alone took 0.24 ms, the LDS-staged
//     epilogue 0.12 ms and the prologue 0.07 ms (each measured by removing the others, so they overlap less than the
//     sum suggests).  The epilogue goes through LDS in fp32 one wave-row (WM pixels) at a time, then 8 channels per
//     thread: one 16-byte fp16 store (and residual load).  Two variants were measured and dropped: storing straight
//     from the accumulators (8-byte pieces, 32 lines per wave instruction: 45 % slower end to end), and requesting the
//     next tile's first k-steps before a 32-row-at-a-time epilogue (no gain on the big layers, 15 % loss on 128x128).
//     The four head convolutions write fp32 (what postprocess reads).
#include <cstdlib>

#include "conv_f16_common.h"

namespace om {

// workgroups per CU each tile shape is built for (LDS: 72 / 48 / 36 / 24 / 36 KiB): caps the register allocation
template <int BM, int BN>
constexpr int f16_blocks_per_cu() { return BM * BN >= 256 * 128 ? 2 : (BM * BN >= 128 * 128 ? 3 : 4); }

// GATHER: the A rows come from up to four tensors at their own resolutions (IgemmHParams::nseg; 1x1 layers only): the row offsets
// are recomputed when the k loop crosses into the next segment, everything else is the same instruction stream and the products
// are summed in the same order as over the materialised concat (bit-identical; the port of conv_igemm_split.hip's form).
template <int BM, int BN, int WM, int WN, int FAST = 0, bool GATHER = false>
__global__ __launch_bounds__(256, (f16_blocks_per_cu<BM, BN>())) void conv_igemm_f16_kernel(const IgemmHParams p) {
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int NWN = BN / WN;
    constexpr int A_CH = BM / 64, B_CH = (BN + 63) / 64, NP = A_CH + B_CH;   // 64 rows x 64 B per workgroup-wide piece
    constexpr int NP0 = NP / 2;                   // pieces issued before the step's barrier
    constexpr int NBUF = 3;
    constexpr int STAGE = (BM + (BN < 64 ? 64 : BN)) * 4;   // f32x4 (16-byte) units per ring stage
    static_assert((BM / WM) * (BN / WN) == 4, "four waves per workgroup");
    static_assert(BM % 64 == 0, "A tile = whole 64-row pieces");
    static_assert(WM * BN / 4 <= NBUF * STAGE, "one wave-row of the fp32 C tile must fit in the operand ring");
    __shared__ f32x4 smem[NBUF * STAGE + 1];      // ONE LDS object (see conv_igemm.hip)
    int* const s_ticket = reinterpret_cast<int*>(smem + NBUF * STAGE);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave / NWN, wn = wave % NWN;
    const int lrow = tid >> 2, lcol = tid & 3;     // loader: row within a 64-row piece, 16-byte position in the row
    const int scol = lcol ^ ((lrow >> 2) & 3);     // logical chunk this lane fetches (the LDS image stays lane-linear)
    const int fi = lane & 31, fk = lane >> 5;
    const int fsw = (fi >> 2) & 3;

    for (;;) {
        // raw barrier: only lane 0's wave pays the ticket's round trip, nobody drains the previous tile's stores
        if (tid == 0) *s_ticket = atomicAdd(p.ticket, 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        int tile = *s_ticket;
        if (tile >= p.total_tiles) break;
        tile = __builtin_amdgcn_readfirstlane(tile);
        const int tile_n = tile % p.n_tiles;
        const int tile_m = tile / p.n_tiles;
        const int m0 = tile_m * BM, n0 = tile_n * BN;

        // ---- loader role: thread -> (row lrow + 64*j, position lcol).  Per A row: the byte offset of tap (0,0)'s chunk
        // relative to the tile's first image, and a mask whose bit t says "tap t of this row is padding / beyond M".
        int rowoff[A_CH];
        unsigned invmask[A_CH];
        [[maybe_unused]] int g_img[A_CH], g_y[A_CH], g_x[A_CH];      // GATHER: the row's pixel, for the segments' own resolutions
        const int b_first = (m0 < p.M ? m0 : p.M - 1) / p.HoWo;
#pragma unroll
        for (int j = 0; j < A_CH; ++j) {
            int m = m0 + lrow + 64 * j;
            const bool mok = m < p.M;
            if (!mok) m = p.M - 1;
            const int b = m / p.HoWo;
            const int rr = m - b * p.HoWo;
            const int oy = rr / p.Wo;
            const int ox = rr - oy * p.Wo;
            const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
            rowoff[j] = (((b - b_first) * p.H * p.W + iy0 * p.W + ix0) * p.in_pix_stride + scol * 8) * 2;
            if constexpr (GATHER) { g_img[j] = b - b_first; g_y[j] = oy; g_x[j] = ox; }
            unsigned badrow = 0, badcol = 0;        // bit k: input row iy0 + k / column ix0 + k is outside the image
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                badrow |= ((unsigned)(iy0 + k) < (unsigned)p.H ? 0u : 1u) << k;
                badcol |= ((unsigned)(ix0 + k) < (unsigned)p.W ? 0u : 1u) << k;
            }
            unsigned inv;
            if (p.ks == 3) {
                inv = ((badrow & 1u) ? 0x007u : 0u) | ((badrow & 2u) ? 0x038u : 0u) | ((badrow & 4u) ? 0x1C0u : 0u) | badcol * 0x49u;
            } else {
                inv = (badrow | badcol) & 1u;
            }
            invmask[j] = mok ? inv : 0xFFFFFFFFu;
        }
        int rowoffB[B_CH];
        const int row_halfs = p.taps * p.cin;
#pragma unroll
        for (int j = 0; j < B_CH; ++j) rowoffB[j] = ((n0 + lrow + 64 * j) * row_halfs + scol * 8) * 2;
        const _Float16* in_base = p.in + (size_t)b_first * p.H * p.W * p.in_pix_stride;
        const size_t in_left = ((size_t)p.total_in_pixels - (size_t)b_first * p.H * p.W) * p.in_pix_stride * 2;
        int in_bytes = in_left < 0x7FFFFFFFull ? (int)in_left : 0x7FFFFFFF;
        [[maybe_unused]] int seg = 0, seg_c0 = 0, seg_c1 = 0x7FFFFFFF;      // GATHER: current segment, its 32-channel chunk range
        [[maybe_unused]] auto set_segment = [&](int g) {
            // (selects, not p.seg_x[g]: a run-time index into the by-value parameter block would copy the arrays to scratch)
            auto pick = [g](const auto (&v)[4]) { return g == 0 ? v[0] : g == 1 ? v[1] : g == 2 ? v[2] : v[3]; };
            const int sh = pick(p.seg_shift), Hs = p.Ho >> sh, Ws = p.Wo >> sh, st = pick(p.seg_stride);
            in_base = pick(p.seg_ptr) + (size_t)b_first * Hs * Ws * st;
            const size_t left = (size_t)(p.nimg - b_first) * Hs * Ws * st * 2;
            in_bytes = left < 0x7FFFFFFFull ? (int)left : 0x7FFFFFFF;
#pragma unroll
            for (int j = 0; j < A_CH; ++j)
                rowoff[j] = (((g_img[j] * Hs + (g_y[j] >> sh)) * Ws + (g_x[j] >> sh)) * st + scol * 8) * 2;
            seg = g;
            seg_c0 = g == 0 ? 0 : g == 1 ? p.seg_end[0] : g == 2 ? p.seg_end[1] : p.seg_end[2];
            seg_c1 = pick(p.seg_end);
        };
        if constexpr (GATHER) set_segment(0);

        int n_kh = 0, n_kw = 0, n_cc = 0;          // step being fetched: (tap row, tap col, 32-channel chunk)
        auto advance = [&]() {
            if (++n_cc == p.kc) {
                n_cc = 0;
                if (++n_kw == p.ks) { n_kw = 0; ++n_kh; }
            }
        };
        // one piece = 64 tile rows x 64 B (16 rows per wave).  `live` = false (a prefetch past the last step) zeroes
        // num_records -> every lane reads zeros
        auto issue_piece = [&](int piece, int buf, bool live) {
            f32x4* dst = smem + buf * STAGE + wave_u * 64;
            if (piece < A_CH) {
                const int j = piece;
                if constexpr (GATHER) {
                    if (piece == 0 && live && n_cc >= seg_c1) set_segment(seg + 1);      // uniform; the steps walk the channels in order
                }
                const auto rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(in_base), 0, live ? in_bytes : 0, 0x00020000);
                const int tap = n_kh * p.ks + n_kw;
                const int tap_off = GATHER ? (n_cc - seg_c0) * 64      // 1x1: one tap; the chunk within its segment
                                           : ((n_kh * p.W + n_kw) * p.in_pix_stride + n_cc * 32) * 2;      // scalar
                const int voff = (rowoff[j] + tap_off) | ((invmask[j] << (31 - tap)) & 0x80000000u);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr_t)(dst + j * 256), 16, voff, 0, 0, 0);
            } else {
                const int j = piece - A_CH;
                // BN = 32: rows 32..63 of the piece belong to the next N tile (or lie past the end: zeros); never read back
                const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.w), 0, live ? p.w_bytes : 0, 0x00020000);
                const int koff = ((n_kh * p.ks + n_kw) * p.cin + n_cc * 32) * 2;                  // scalar
                // (a plain local: passing an expression with a captured array element straight to the builtin makes
                // hipcc drop the kernel's host stub)
                const int vo = rowoffB[j] + koff;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(dst + BM * 4 + j * 256), 16, vo, 0, 0, 0);
            }
        };

        f32x16 acc[TM][TN];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

        const f32x4* fragA = smem + (wm * WM + fi) * 4;
        const f32x4* fragB = smem + BM * 4 + (wn * WN + fi) * 4;
        f32x4 ca[TM], cb[TN], na[TM], nb[TN];
        auto read_frags = [&](f32x4(&fa)[TM], f32x4(&fb)[TN], int buf, int q) {
            const int ch = (2 * q + fk) ^ fsw;
            const int bo = buf * STAGE;
#pragma unroll
            for (int a = 0; a < TM; ++a) fa[a] = fragA[bo + a * 32 * 4 + ch];
#pragma unroll
            for (int b = 0; b < TN; ++b) fb[b] = fragB[bo + b * 32 * 4 + ch];
        };
        auto mfmas = [&]() {
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    // weights first: D[i = channel][j = pixel]
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, cb[b]),
                                                                       __builtin_bit_cast(f16x8, ca[a]), acc[a][b], 0, 0, 0);
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
            const int buf2 = buf1 == NBUF - 1 ? 0 : buf1 + 1;    // step s+2: last read in step s-1, before its barrier
            const bool live2 = s + 2 < p.ksteps;
            // ---- k-slice 0 (the LDS reads of the next slice are issued first: they return under the MFMAs)
            read_frags(na, nb, buf, 1);
            mfmas();
#pragma unroll
            for (int i = 0; i < NP0; ++i) issue_piece(i, buf2, live2);
            // step s+1 has landed (everything older than the NP0 pieces just issued); my reads of `buf` are done
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"i"(NP0) : "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < TM; ++a) ca[a] = na[a];
#pragma unroll
            for (int b = 0; b < TN; ++b) cb[b] = nb[b];
            // ---- k-slice 1
            read_frags(na, nb, buf1, 0);
            mfmas();
#pragma unroll
            for (int i = NP0; i < NP; ++i) issue_piece(i, buf2, live2);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < TM; ++a) ca[a] = na[a];
#pragma unroll
            for (int b = 0; b < TN; ++b) cb[b] = nb[b];
            buf = buf1;
            advance();
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();

        f16_epilogue<BM, BN, WM, WN, FAST>(p, smem, acc, m0, n0, tid, wm, wn, fi, fk);
    }
}

struct TileChoiceH { int bm, bn; };

// Matrix-pipe time over all tiles, weighted by how well the shape feeds the pipe; partial rounds are not charged (the other
// batch in flight fills them: conv3x3_f16.hip, conv3x3_tile_for_f16).
static TileChoiceH choose_tile_f16(int M, int cout_pad) {
    struct Cand { int bm, bn; double eff; };
    const Cand cands[] = {{256, 128, 1.00}, {128, 128, 0.85}, {128, 64, 0.70}, {64, 64, 0.55}, {128, 32, 0.45}};
    TileChoiceH best{0, 0};
    double best_cost = 1e300;
    for (const Cand& c : cands) {
        if (cout_pad % c.bn) continue;
        const long long tiles = (long long)((M + c.bm - 1) / c.bm) * (cout_pad / c.bn);
        const double cost = (double)tiles * c.bm * c.bn / c.eff;
        if (cost < best_cost) { best_cost = cost; best = TileChoiceH{c.bm, c.bn}; }
    }
    return best;
}

void conv_tile_for_f16(int M, int cout_pad, int cin, int* bm, int* bn) {
    (void)cin;
    const TileChoiceH t = choose_tile_f16(M, cout_pad);
    *bm = t.bm; *bn = t.bn;
}

template <int BM, int BN, int WM, int WN, bool GATHER = false>
static int launch_tile_f16(IgemmHParams p, int cout_pad, int blocks_per_cu, hipStream_t stream) {
    const int m_tiles = (p.M + BM - 1) / BM;
    p.n_tiles = cout_pad / BN;
    const long long total = (long long)m_tiles * p.n_tiles;
    OM_REQUIRE(total > 0 && total < (1ll << 31), OM_EINVAL, "conv f16: %lld tiles out of range", total);
    p.total_tiles = (int)total;
    long long grid = total;
    grid = total < 256ll * blocks_per_cu ? total : 256ll * blocks_per_cu;
    // the epilogue without loads in its row sweeps (f16_epilogue: FAST) wherever the layer allows it
    const bool fast = p.out_mode == 0 && !p.out_f32 && p.vec_io && p.cout == cout_pad;
    if constexpr (GATHER) {
        OM_REQUIRE(fast && !p.res, OM_EINVAL, "conv f16: a gathered input needs a plain fp16 NHWC output without residual");
        hipLaunchKernelGGL((conv_igemm_f16_kernel<BM, BN, WM, WN, 1, true>), dim3((unsigned)grid), dim3(256), 0, stream, p);
        OM_CHECK_HIP(hipGetLastError());
        return OM_OK;
    }
    if (fast && !p.res) hipLaunchKernelGGL((conv_igemm_f16_kernel<BM, BN, WM, WN, 1>), dim3((unsigned)grid), dim3(256), 0, stream, p);
    else if (fast) hipLaunchKernelGGL((conv_igemm_f16_kernel<BM, BN, WM, WN, 2>), dim3((unsigned)grid), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((conv_igemm_f16_kernel<BM, BN, WM, WN, 0>), dim3((unsigned)grid), dim3(256), 0, stream, p);
    OM_CHECK_HIP(hipGetLastError());
    return OM_OK;
}

int launch_conv_igemm_f16(const ConvArgsH& a, hipStream_t stream) {
    OM_REQUIRE((a.in || a.nseg > 0) && a.w && a.scale && a.shift && a.out, OM_EINVAL, "conv f16: null pointer");
    if (a.nseg == 0 && conv3x3_f16_supported(a)) return launch_conv3x3_f16(a, stream);
    OM_REQUIRE(a.cin % 32 == 0 && a.cin >= 32, OM_EINVAL, "conv f16: cin=%d must be a multiple of 32", a.cin);
    OM_REQUIRE(a.ks == 1 || a.ks == 3, OM_EINVAL, "conv f16: ksize=%d not supported", a.ks);
    OM_REQUIRE(a.stride == 1 || a.stride == 2, OM_EINVAL, "conv f16: stride=%d not supported", a.stride);
    OM_REQUIRE((a.nseg > 0 || (a.in_pix_stride % 8 == 0 && (reinterpret_cast<uintptr_t>(a.in) & 15) == 0)) &&
                   (reinterpret_cast<uintptr_t>(a.w) & 15) == 0,
               OM_EINVAL, "conv f16: input view / weights must be 16-byte aligned");
    OM_REQUIRE(a.cout_pad % 32 == 0 && a.cout <= a.cout_pad, OM_EINVAL, "conv f16: cout_pad=%d", a.cout_pad);
    OM_REQUIRE((long long)a.B * a.H * a.W < (1ll << 31) && (long long)a.B * a.Ho * a.Wo < (1ll << 31), OM_EINVAL,
               "conv f16: more than 2^31 pixels");
    OM_REQUIRE(!(a.res && (a.out_mode != 0 || a.out_f32)), OM_EINVAL, "conv f16: residual only with plain NHWC fp16 output");
    OM_REQUIRE(!(a.out_mode == 2 && !a.out_f32) && !(a.out_mode == 1 && a.out_f32), OM_EINVAL,
               "conv f16: NCHW output is fp32, up-sampled output is fp16");
    IgemmHParams p;
    p.in = static_cast<const _Float16*>(a.in); p.w = static_cast<const _Float16*>(a.w);
    p.scale = a.scale; p.shift = a.shift; p.res = static_cast<const _Float16*>(a.res); p.out = a.out;
    p.ticket = a.ticket;
    p.H = a.H; p.W = a.W; p.cin = a.cin; p.in_pix_stride = a.in_pix_stride;
    p.Ho = a.Ho; p.Wo = a.Wo; p.HoWo = a.Ho * a.Wo; p.cout = a.cout;
    p.ks = a.ks; p.stride = a.stride; p.pad = a.ks / 2;
    p.M = a.B * a.Ho * a.Wo; p.taps = a.ks * a.ks;
    p.kc = a.cin / 32;
    p.ksteps = p.taps * p.kc;
    p.leaky = a.leaky; p.res_pix_stride = a.res_pix_stride; p.out_pix_stride = a.out_pix_stride;
    p.out_mode = a.out_mode; p.up = a.up; p.out_f32 = a.out_f32;
    p.n_tiles = 0; p.total_tiles = 0;
    p.total_in_pixels = a.B * a.H * a.W;
    p.w_bytes = (int)(conv_f16_weight_halfs(a.cout_pad, a.ks, a.cin) * 2);
    const int esz = a.out_f32 ? 4 : 2;
    p.vec_io = (a.out_mode != 2 && (a.out_pix_stride * esz) % 16 == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0 &&
                (!a.res || (a.res_pix_stride % 8 == 0 && (reinterpret_cast<uintptr_t>(a.res) & 15) == 0)))
                   ? 1 : 0;
    OM_REQUIRE(a.ticket, OM_EINVAL, "conv f16: the tile queue needs a zeroed ticket word");
    p.nseg = 0; p.nimg = a.B;
    for (int g = 0; g < 4; ++g) { p.seg_ptr[g] = p.in; p.seg_stride[g] = 0; p.seg_shift[g] = 0; p.seg_end[g] = 0x7FFFFFFF; }
    int bm, bn;
    conv_tile_for_f16(p.M, a.cout_pad, a.cin, &bm, &bn);
    if (a.nseg > 0) {
        // gathered input: 1x1, whole 32-channel chunks per segment, every segment's resolution a power-of-two fraction of this one
        OM_REQUIRE(a.nseg <= 4 && a.ks == 1 && a.stride == 1 && a.cout_pad % 128 == 0, OM_EINVAL,
                   "conv f16: a gathered input needs a 1x1 stride-1 layer with cout_pad %% 128 == 0 and at most 4 segments (nseg=%d ks=%d "
                   "stride=%d cout_pad=%d)", a.nseg, a.ks, a.stride, a.cout_pad);
        int end = 0;
        for (int g = 0; g < a.nseg; ++g) {
            const int up = a.seg_up[g];
            int sh = 0;
            while ((1 << sh) < up) ++sh;
            OM_REQUIRE(a.seg_ptr[g] && up >= 1 && (1 << sh) == up && a.H % up == 0 && a.W % up == 0 && a.seg_channels[g] > 0 &&
                           a.seg_channels[g] % 32 == 0 && a.seg_pix_stride[g] % 8 == 0 && a.seg_pix_stride[g] >= a.seg_channels[g] &&
                           (reinterpret_cast<uintptr_t>(a.seg_ptr[g]) & 15) == 0,
                       OM_EINVAL, "conv f16: segment %d (channels=%d pix_stride=%d up=%d) of a gathered input", g, a.seg_channels[g],
                       a.seg_pix_stride[g], up);
            end += a.seg_channels[g] / 32;
            p.seg_ptr[g] = static_cast<const _Float16*>(a.seg_ptr[g]);
            p.seg_stride[g] = a.seg_pix_stride[g]; p.seg_shift[g] = sh; p.seg_end[g] = end;
        }
        OM_REQUIRE(end * 32 == a.cin, OM_EINVAL, "conv f16: the segments hold %d channels, the layer reads %d", end * 32, a.cin);
        p.nseg = a.nseg;
        p.in = p.seg_ptr[0];
        if (bm == 256) return launch_tile_f16<256, 128, 128, 64, true>(p, a.cout_pad, 2, stream);
        return launch_tile_f16<128, 128, 64, 64, true>(p, a.cout_pad, 3, stream);
    }
    if (bm == 256 && bn == 128) return launch_tile_f16<256, 128, 128, 64>(p, a.cout_pad, 2, stream);
    if (bm == 128 && bn == 128) return launch_tile_f16<128, 128, 64, 64>(p, a.cout_pad, 3, stream);
    if (bm == 128 && bn == 64) return launch_tile_f16<128, 64, 64, 32>(p, a.cout_pad, 4, stream);
    if (bm == 64 && bn == 64) return launch_tile_f16<64, 64, 32, 32>(p, a.cout_pad, 4, stream);
    return launch_tile_f16<128, 32, 32, 32>(p, a.cout_pad, 4, stream);
}

}  // namespace om
