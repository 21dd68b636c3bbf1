// 3x3 stride-1 convolution, fp16 activations / fp32 accumulate, with the three horizontal taps of a kernel row sharing
// ONE copy of the input rows in LDS (the stride-1 3x3 layers are 88 % of the network's FLOPs,
// /root/reference/model/backbone/darknet.py:9-13, model/orienmask_yolo_fpnplus.py:33-71; arithmetic as
// conv_igemm_f16.hip: fp16 operands, fp32 sums, one rounding at the store).
//
// Why (MI355X): in the generic implicit-GEMM kernel every tap re-fetches the A operand, and measured on the 136^2
// 128->256 layer the k-loop was bound by the LDS-DMA path (instruction issue and ~9 TB/s of L2->LDS traffic), not by the
// matrix pipe: with the operand fetches answered with zeros it ran 1.7x faster.  For an M tile of BM consecutive raster
// pixels the A rows of taps (kh, 0..2) are the same BM+2 input pixels shifted by one, so:
//   * per (channel chunk, kernel row) ONE "patch" of BM+2 input pixels x 32 channels is DMA'd (BM/64 + 1 pieces instead
//     of 3 * BM/64) and the three taps read their fragments at row offsets 0, 1, 2;
//   * zero padding can no longer come from the DMA (a patch row is a valid neighbour for one tap and padding for
//     another): every lane carries a 9-bit "tap is padding" mask for each of its pixels and ANDs its A fragment with it
//     (4 v_and per fragment, hidden under the MFMAs);
//   * the weights stream through a 3-deep ring of per-tap slices (BN rows x 64 B), two steps ahead, as before; the patch
//     is double buffered and requested during the first two taps of the previous patch; counted vmcnt per tap.
// LDS: 2 x (BM + 64) + 3 x BN rows of 64 B = 64 KiB for the 256x128 tile (2 workgroups/CU), exactly the fp32 C tile of
// one wave-row for the epilogue (conv_f16_common.h).
#include <cstdlib>

#include "conv_f16_common.h"

namespace om {

template <int BM, int BN>
constexpr int c3_blocks_per_cu() { return BM * BN >= 256 * 128 ? 2 : (BM * BN >= 128 * 128 ? 3 : 4); }

template <int BM, int BN, int WM, int WN, int FAST = 0>
__global__ __launch_bounds__(256, (c3_blocks_per_cu<BM, BN>())) void conv3x3_f16_kernel(const IgemmHParams p) {
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int NWN = BN / WN;
    constexpr int A_CH = BM / 64 + 1;             // patch pieces (64 rows x 64 B each); BM + 2 rows are needed
    constexpr int P0 = (A_CH + 1) / 2;            // patch pieces requested during tap 0 (the rest during tap 1)
    constexpr int B_CH = BN / 64;
    constexpr int PATCH = (BM + 64) * 4;          // f32x4 units per patch buffer
    constexpr int BSL = BN * 4;                   // f32x4 units per weight slice
    static_assert((BM / WM) * (BN / WN) == 4, "four waves per workgroup");
    static_assert(BM % 64 == 0 && BN % 64 == 0, "whole 64-row pieces");
    static_assert(WM * BN / 4 <= 2 * PATCH + 3 * BSL, "one wave-row of the fp32 C tile must fit in LDS");
    __shared__ f32x4 smem[2 * PATCH + 3 * BSL + 1];       // ONE LDS object; the last 16 B hold the ticket
    int* const s_ticket = reinterpret_cast<int*>(smem + 2 * PATCH + 3 * BSL);
    f32x4* const s_patch = smem;
    f32x4* const s_w = smem + 2 * PATCH;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave / NWN, wn = wave % NWN;
    const int lrow = tid >> 2, lcol = tid & 3;     // loader: row within a 64-row piece, 16-byte position in the row
    const int scol = lcol ^ ((lrow >> 2) & 3);     // logical chunk this lane fetches (the LDS image stays lane-linear)
    const int fi = lane & 31, fk = lane >> 5;
    const int fswB = (fi >> 2) & 3;
    const int npatch = 3 * p.kc;                   // (channel chunk, kernel row) pairs
    const int in_bytes = ((p.total_in_pixels - 1) * p.in_pix_stride + p.cin) * 2;     // < 2^31, checked by the host
    const int row_halfs = 9 * p.cin;

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

        // loader role: patch row r = lrow + 64 j is input pixel m0 - 1 + (kh - 1) W + r (raster index over the batch);
        // a negative or past-the-end offset is out of the descriptor's range -> zeros
        int rowbase[A_CH], rowoffB[B_CH];
#pragma unroll
        for (int j = 0; j < A_CH; ++j) rowbase[j] = ((m0 - 1 + lrow + 64 * j) * p.in_pix_stride + scol * 8) * 2;
#pragma unroll
        for (int j = 0; j < B_CH; ++j) rowoffB[j] = ((n0 + lrow + 64 * j) * row_halfs + scol * 8) * 2;
        // consumer role: "tap t is padding" bits of this lane's pixels (one per MFMA row block)
        unsigned inv[TM];
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            int m = m0 + wm * WM + a * 32 + fi;
            if (m >= p.M) m = p.M - 1;              // rows beyond M are never stored
            const int rr = m % p.HoWo;
            const int y = rr / p.W, x = rr - y * p.W;
            const unsigned badrow = (y == 0 ? 1u : 0u) | (y == p.H - 1 ? 4u : 0u);
            const unsigned badcol = (x == 0 ? 1u : 0u) | (x == p.W - 1 ? 4u : 0u);
            inv[a] = ((badrow & 1u) ? 0x007u : 0u) | ((badrow & 4u) ? 0x1C0u : 0u) | badcol * 0x49u;
        }

        // ---- DMA issue helpers (plain locals go to the builtin: see conv_igemm_f16.hip)
        auto issue_patch_piece = [&](int j, int pi, bool live) {
            const int cc = pi / 3, kh = pi - 3 * cc;
            const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.in), 0, live ? in_bytes : 0, 0x00020000);
            const int soff = ((kh - 1) * p.W * p.in_pix_stride + cc * 32) * 2;            // scalar
            const int vo = rowbase[j] + soff;
            f32x4* dst = s_patch + (pi & 1) * PATCH + wave_u * 64 + j * 256;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)dst, 16, vo, 0, 0, 0);
        };
        auto issue_w_piece = [&](int j, int s, bool live) {
            const int pi = s / 3, kw = s - 3 * pi;
            const int cc = pi / 3, kh = pi - 3 * cc;
            const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.w), 0, live ? p.w_bytes : 0, 0x00020000);
            const int koff = ((kh * 3 + kw) * p.cin + cc * 32) * 2;                       // scalar
            const int vo = rowoffB[j] + koff;
            f32x4* dst = s_w + (s % 3) * BSL + wave_u * 64 + j * 256;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)dst, 16, vo, 0, 0, 0);
        };

        f32x16 acc[TM][TN];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

        f32x4 ca[TM], cb[TN], na[TM], nb[TN];
        // fragments of (patch buffer pb, tap column kw, weight slot ws), k-slice q
        auto read_frags = [&](f32x4(&fa)[TM], f32x4(&fb)[TN], int pb, int kw, int ws, int q) {
            const int rowA = wm * WM + fi + kw;
            const int chA = (2 * q + fk) ^ ((rowA >> 2) & 3);
            const f32x4* pa = s_patch + pb * PATCH + rowA * 4 + chA;
#pragma unroll
            for (int a = 0; a < TM; ++a) fa[a] = pa[a * 32 * 4];          // + 32 rows keeps (row >> 2) & 3
            const f32x4* pw = s_w + ws * BSL + (wn * WN + fi) * 4 + ((2 * q + fk) ^ fswB);
#pragma unroll
            for (int b = 0; b < TN; ++b) fb[b] = pw[b * 32 * 4];
        };
        auto take = [&](int tap) {       // next -> current, A masked with this tap's padding bits
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                const unsigned keep = ((inv[a] >> tap) & 1u) ? 0u : 0xFFFFFFFFu;
                const unsigned* src = reinterpret_cast<const unsigned*>(&na[a]);
                unsigned* dst = reinterpret_cast<unsigned*>(&ca[a]);
#pragma unroll
                for (int k = 0; k < 4; ++k) dst[k] = src[k] & keep;
            }
#pragma unroll
            for (int b = 0; b < TN; ++b) cb[b] = nb[b];
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

        // prologue: patch 0 and weight slices 0, 1 in flight; everything but slice 1 waited for
#pragma unroll
        for (int j = 0; j < A_CH; ++j) issue_patch_piece(j, 0, true);
#pragma unroll
        for (int j = 0; j < B_CH; ++j) issue_w_piece(j, 0, true);
#pragma unroll
        for (int j = 0; j < B_CH; ++j) issue_w_piece(j, 1, true);
        asm volatile("s_waitcnt vmcnt(%0)" ::"i"(B_CH) : "memory");
        __builtin_amdgcn_s_barrier();
        read_frags(na, nb, 0, 0, 0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        take(0);

        int s = 0;
        for (int pi = 0; pi < npatch; ++pi) {
            const int pb = pi & 1;
            const int kh = pi % 3;
            const bool more_patch = pi + 1 < npatch;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw, ++s) {
                const int tap = kh * 3 + kw;
                const int ws = s % 3;
                const bool live2 = s + 2 < 3 * npatch;
                // ---- k-slice 0 (the LDS reads of the next slice are issued first: they return under the MFMAs)
                read_frags(na, nb, pb, kw, ws, 1);
                mfmas();
#pragma unroll
                for (int j = 0; j < B_CH; ++j) issue_w_piece(j, s + 2, live2);
                if (kw == 0) {
#pragma unroll
                    for (int j = 0; j < P0; ++j) issue_patch_piece(j, pi + 1, more_patch);
                } else if (kw == 1) {
#pragma unroll
                    for (int j = P0; j < A_CH; ++j) issue_patch_piece(j, pi + 1, more_patch);
                }
                // everything requested before this tap has landed (weight slice s+1, and before tap 2 the whole next
                // patch); my reads of this tap's buffers are done
                if (kw == 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"i"(B_CH + P0) : "memory");
                else if (kw == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"i"(B_CH + A_CH - P0) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"i"(B_CH) : "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                take(tap);
                // ---- k-slice 1; the next tap's first fragments come from the next weight slot (and, after tap 2, from
                // the other patch buffer, tap column 0)
                if (kw < 2) read_frags(na, nb, pb, kw + 1, (s + 1) % 3, 0);
                else read_frags(na, nb, pb ^ 1, 0, (s + 1) % 3, 0);
                mfmas();
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                take(kw < 2 ? tap + 1 : ((kh + 1) % 3) * 3);
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();

        f16_epilogue<BM, BN, WM, WN, FAST>(p, smem, acc, m0, n0, tid, wm, wn, fi, fk);
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// The TALL-PATCH form (round 5): 512 raster pixels x 128 output channels per workgroup of EIGHT waves, ONE patch per 32-channel
// chunk for all nine taps.
//
// conv3x3_f16_kernel keeps ~72 KB of operands in flight per CU against ~2 us of memory latency, about 15 B/clk, and its 256 x 128
// tile needs 123 KB of requests per chunk of 32 channels (three patches of 258 pixels + nine weight slices of 128 rows) for 18.9
// MFLOP: the matrix pipe waits for operands 57 % of the time (profiles/r04_experiments.md section 12).  What raises the rate is
// fewer requested bytes per matrix instruction:
//   * the input rows of the three KERNEL ROWS overlap too: for BM consecutive raster pixels all nine taps read the SAME
//     BM + 2 W + 2 pixels (patch row r = input pixel m0 - 1 - W + r; tap (kh, kw) of tile pixel i is patch row i + kh W + kw),
//     so one patch per chunk replaces three: 786 instead of 3 x 514 pixels at W = 136 for 512 pixels;
//   * 512 pixels share every weight slice (twice the pixels per weight byte).
// Per chunk: 50 + 74 KB for 37.7 MFLOP -- half the requests per matrix instruction, with the same 128 x 64 wave tile, fragment
// reads, padding masks (a patch row is a neighbour for one tap and padding for another: 9-bit mask per pixel, as above) and
// epilogue.  LDS: two patch buffers of NP x 128 rows + four weight slices of 128 rows, 64 B each = 114.7 + 32.8 KB at NP = 7.
// Per tap every thread requests at most one patch piece (the NEXT chunk's, taps 0 .. NP-1) and one weight piece (three taps
// ahead, four-slot ring), retired in order: a counted `vmcnt` before the tap's hand-over means "the next tap's weights have landed".
constexpr int C3T_BM = 512, C3T_BN = 128, C3T_WM = 128, C3T_WN = 64, C3T_NT = 512;

template <int NP, int FAST>
__global__ __launch_bounds__(C3T_NT, 1) void conv3x3_f16_tall_kernel(const IgemmHParams p) {
    constexpr int BM = C3T_BM, BN = C3T_BN, WM = C3T_WM, WN = C3T_WN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int NS = 4;                         // weight slices in the ring (requested three taps ahead)
    constexpr int PATCH = NP * 128 * 4;           // f32x4 units per patch buffer (rows of 64 B)
    constexpr int BSL = BN * 4;                   // f32x4 units per weight slice
    static_assert(WM * BN / 4 <= 2 * PATCH, "one wave-row of the fp32 C tile must fit in the patch buffers");
    static_assert(NP >= 5 && NP <= 7, "patch pieces are requested during taps 0 .. NP-1, before the last two taps");
    __shared__ f32x4 smem[2 * PATCH + NS * BSL + 1];
    int* const s_ticket = reinterpret_cast<int*>(smem + 2 * PATCH + NS * BSL);
    f32x4* const s_patch = smem;
    f32x4* const s_w = smem + 2 * PATCH;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int grp = wave_u >> 2;                   // waves w and w + 4 share a SIMD: one of each group per SIMD
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = tid >> 2, lcol = tid & 3;     // loader: row within a 128-row piece, 16-byte position in the row
    const int scol = lcol ^ ((lrow >> 2) & 3);     // logical chunk this lane fetches (the LDS image stays lane-linear)
    const int fi = lane & 31, fk = lane >> 5;
    const int fswB = (fi >> 2) & 3;
    const int in_bytes = ((p.total_in_pixels - 1) * p.in_pix_stride + p.cin) * 2;     // < 2^31, checked by the host
    const int row_halfs = 9 * p.cin;
    const int W = p.W;

    for (;;) {
        if (tid == 0) *s_ticket = atomicAdd(p.ticket, 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        int tile = *s_ticket;
        if (tile >= p.total_tiles) break;
        tile = __builtin_amdgcn_readfirstlane(tile);
        const int tile_n = tile % p.n_tiles;
        const int tile_m = tile / p.n_tiles;
        const int m0 = tile_m * BM, n0 = tile_n * BN;

        // loader role: patch row r = lrow + 128 j is input pixel m0 - 1 - W + r (raster index over the batch); a negative or
        // past-the-end offset is out of the descriptor's range -> zeros
        const int rowbase0 = ((m0 - 1 - W + lrow) * p.in_pix_stride + scol * 8) * 2;
        const int piece_bytes = 128 * p.in_pix_stride * 2;
        const int rowoffB = ((n0 + lrow) * row_halfs + scol * 8) * 2;
        // consumer role: "tap t is padding" bits of this lane's pixels (one per MFMA row block)
        unsigned inv[TM];
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            int m = m0 + wm * WM + a * 32 + fi;
            if (m >= p.M) m = p.M - 1;              // rows beyond M are never stored
            const int rr = m % p.HoWo;
            const int y = rr / p.W, x = rr - y * p.W;
            const unsigned badrow = (y == 0 ? 1u : 0u) | (y == p.H - 1 ? 4u : 0u);
            const unsigned badcol = (x == 0 ? 1u : 0u) | (x == p.W - 1 ? 4u : 0u);
            inv[a] = ((badrow & 1u) ? 0x007u : 0u) | ((badrow & 4u) ? 0x1C0u : 0u) | badcol * 0x49u;
        }

        auto issue_patch_piece = [&](int j, int cc, bool live) {
            const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.in), 0, live ? in_bytes : 0, 0x00020000);
            const int vo = rowbase0 + j * piece_bytes + cc * 64;
            f32x4* dst = s_patch + (cc & 1) * PATCH + j * 512 + wave_u * 64;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)dst, 16, vo, 0, 0, 0);
        };
        auto issue_w_piece = [&](int cc, int tap, bool live) {       // slice 9 cc + tap -> slot (cc + tap) & 3
            const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.w), 0, live ? p.w_bytes : 0, 0x00020000);
            const int vo = rowoffB + (tap * p.cin + cc * 32) * 2;
            f32x4* dst = s_w + ((cc + tap) & (NS - 1)) * BSL + wave_u * 64;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)dst, 16, vo, 0, 0, 0);
        };

        f32x16 acc[TM][TN];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

        f32x4 fa[2][TM], fb[2][TN];
        const int rowA0 = wm * WM + fi;

        // prologue: patch 0 and weight slices 0, 1, 2 requested; patch and slice 0 waited for by everybody
#pragma unroll
        for (int j = 0; j < NP; ++j) issue_patch_piece(j, 0, true);
        issue_w_piece(0, 0, true);
        issue_w_piece(0, 1, true);
        issue_w_piece(0, 2, true);
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // PING-PONG: the waves of group 1 run one phase behind group 0 -- while one wave of a SIMD issues its tap's sixteen
        // matrix instructions the other one reads its fragments, masks them and requests operands, and the barrier between two
        // phases is the hand-over.  (With every wave in the same phase -- the shared-patch kernel's loop -- both waves of a SIMD
        // stand at the barrier, then both mask, then both want the matrix pipe: it is idle half of the time.)
        if (grp) __builtin_amdgcn_s_barrier();

        for (int cc = 0; cc < p.kc; ++cc) {
            const int pb = cc & 1;
            const bool more = cc + 1 < p.kc;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int kh = tap / 3, kw = tap - 3 * kh;
                // ---- memory phase of this tap: requests (the next chunk's patch piece, the weight slice three taps ahead) ...
                if (tap < NP) issue_patch_piece(tap, cc + 1, more);
                if (tap < 6) issue_w_piece(cc, tap + 3, true);
                else issue_w_piece(cc + 1, tap - 6, more);
                // ... the tap's twelve fragments, A masked with the tap's padding bits
                {
                    const int rowA = rowA0 + kh * W + kw;
                    const int sw = (rowA >> 2) & 3;
                    const f32x4* pa = s_patch + pb * PATCH + rowA * 4;
                    const f32x4* pw = s_w + ((cc + tap) & (NS - 1)) * BSL + (wn * WN + fi) * 4;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
#pragma unroll
                        for (int a = 0; a < TM; ++a) fa[q][a] = pa[a * 32 * 4 + ((2 * q + fk) ^ sw)];      // + 32 rows keeps (row >> 2) & 3
#pragma unroll
                        for (int b = 0; b < TN; ++b) fb[q][b] = pw[b * 32 * 4 + ((2 * q + fk) ^ fswB)];
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int a = 0; a < TM; ++a) {
                    const unsigned keep = ((inv[a] >> tap) & 1u) ? 0u : 0xFFFFFFFFu;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        unsigned* d = reinterpret_cast<unsigned*>(&fa[q][a]);
#pragma unroll
                        for (int k = 0; k < 4; ++k) d[k] &= keep;
                    }
                }
                // the NEXT tap's weight slice (and, before tap 0, the next patch) has landed: requested two taps ago; younger
                // are the previous and this tap's requests
                constexpr int dummy = 0; (void)dummy;
                if (tap == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 + 0 + 1) : "memory");
                else if (tap < NP) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 + 1 + 1) : "memory");
                else if (tap == NP) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 + 1 + 0) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2) : "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // ---- matrix phase
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int a = 0; a < TM; ++a)
#pragma unroll
                        for (int b = 0; b < TN; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fb[q][b]),
                                                                               __builtin_bit_cast(f16x8, fa[q][a]), acc[a][b], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (!grp) __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();

        f16_epilogue<BM, BN, WM, WN, FAST, C3T_NT>(p, smem, acc, m0, n0, tid, wm, wn, fi, fk);
    }
}

template <int BM, int BN, int WM, int WN>
static int launch_c3(IgemmHParams p, int cout_pad, hipStream_t stream) {
    const int m_tiles = (p.M + BM - 1) / BM;
    p.n_tiles = cout_pad / BN;
    const long long total = (long long)m_tiles * p.n_tiles;
    OM_REQUIRE(total > 0 && total < (1ll << 31), OM_EINVAL, "conv3x3 f16: %lld tiles out of range", total);
    p.total_tiles = (int)total;
    const long long slots = 256ll * c3_blocks_per_cu<BM, BN>();
    const long long grid = total < slots ? total : slots;
    // the epilogue without loads in its row sweeps (f16_epilogue: FAST) wherever the layer allows it
    const bool fast = p.out_mode == 0 && !p.out_f32 && p.vec_io && p.cout == cout_pad;
    if (fast && !p.res) hipLaunchKernelGGL((conv3x3_f16_kernel<BM, BN, WM, WN, 1>), dim3((unsigned)grid), dim3(256), 0, stream, p);
    else if (fast) hipLaunchKernelGGL((conv3x3_f16_kernel<BM, BN, WM, WN, 2>), dim3((unsigned)grid), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((conv3x3_f16_kernel<BM, BN, WM, WN, 0>), dim3((unsigned)grid), dim3(256), 0, stream, p);
    OM_CHECK_HIP(hipGetLastError());
    return OM_OK;
}

// true if the layer can run here: stride-1 3x3 whose input view fits a 2^31-byte buffer descriptor and whose rows are
// short enough for a patch (any W works: the patch is a run of raster pixels, not an image rectangle)
bool conv3x3_f16_supported(const ConvArgsH& a) {
    if (a.ks != 3 || a.stride != 1 || a.cin % 32 || a.cout_pad % 64) return false;
    const long long bytes = ((long long)a.B * a.H * a.W - 1) * a.in_pix_stride * 2 + a.cin * 2;
    return bytes < 0x70000000ll && a.out_mode == 0;      // headroom: patch rows run up to 320 + W pixels past the end
}

template <int NP>
static int launch_c3_tall(IgemmHParams p, int cout_pad, hipStream_t stream) {
    const int m_tiles = (p.M + C3T_BM - 1) / C3T_BM;
    p.n_tiles = cout_pad / C3T_BN;
    const long long total = (long long)m_tiles * p.n_tiles;
    OM_REQUIRE(total > 0 && total < (1ll << 31), OM_EINVAL, "conv3x3 f16: %lld tiles out of range", total);
    p.total_tiles = (int)total;
    const long long grid = total < 256 ? total : 256;
    const bool fast = p.out_mode == 0 && !p.out_f32 && p.vec_io && p.cout == cout_pad;
    if (fast && !p.res) hipLaunchKernelGGL((conv3x3_f16_tall_kernel<NP, 1>), dim3((unsigned)grid), dim3(C3T_NT), 0, stream, p);
    else if (fast) hipLaunchKernelGGL((conv3x3_f16_tall_kernel<NP, 2>), dim3((unsigned)grid), dim3(C3T_NT), 0, stream, p);
    else hipLaunchKernelGGL((conv3x3_f16_tall_kernel<NP, 0>), dim3((unsigned)grid), dim3(C3T_NT), 0, stream, p);
    OM_CHECK_HIP(hipGetLastError());
    return OM_OK;
}

// 0: never the tall-patch form; 1 (default): where conv3x3_tile_for_f16 picks it; 2: wherever it can run.  OM_C3_TALL in the
// environment, or om_set_conv3x3_f16_variant (A/B runs, tools/conv16_bench.py, the unit tests of both forms).
static int g_c3_tall = -1;
static int c3_tall_mode() {
    if (g_c3_tall < 0) {
        const char* e = std::getenv("OM_C3_TALL");
        g_c3_tall = e ? std::atoi(e) : 1;
    }
    return g_c3_tall;
}
void conv3x3_f16_set_tall(int mode) { g_c3_tall = mode; }
int conv3x3_f16_get_tall() { return c3_tall_mode(); }

void conv3x3_tile_for_f16(int M, int cout_pad, int W, int kc, int* bm, int* bn) {
    if (cout_pad % 128) { *bm = 128; *bn = 64; return; }
    *bn = 128;
    // the tall-patch form: 512 + 2 W + 2 patch rows must fit its seven 128-row pieces.  It is ONE workgroup per CU whose tile
    // ends overlap with nothing, against two co-resident workgroups of the 256-pixel tile: per unit of work it is 8 % faster
    // from 128 input channels on, 17 % from 512 on (longer k loops), and what decides is whole rounds -- ceil(tiles / 256) of its
    // tiles against half as many rounds of twice as many (same-box single-layer A/B at bs = 8 ... 64, profiles/r05_experiments.md
    // section 4: every measured shape falls on the side this rule puts it, but for two within 5 %).
    const bool tall_fits = C3T_BM + 2 * W + 2 <= 7 * 128;
    const long long t512 = (long long)((M + 511) / 512) * (cout_pad / 128);
    if (tall_fits && c3_tall_mode() == 2) { *bm = 512; return; }
    if (tall_fits && c3_tall_mode() == 1 && kc >= 4) {
        const long long t256r = (long long)((M + 255) / 256) * (cout_pad / 128);
        const double rounds_tall = (double)((t512 + 255) / 256) * (kc >= 16 ? 0.83 : 0.92);
        const double rounds_pair = (double)((t256r + 255) / 256) / 2.0;
        if (rounds_tall <= rounds_pair) { *bm = 512; return; }
    }
    // the tile with the least matrix-pipe time over all its tiles: 256x128 unless its M padding costs more than the finer
    // tile's 15 % lower efficiency.  Whole rounds of tiles are NOT part of the cost any more: with two batches in flight
    // (pipeline.py) the other batch's kernels use the compute units a partial round leaves idle -- same-box A/B +4 % with two in
    // flight, -1.7 % one batch at a time (profiles/r02_experiments.md).
    const long long t256 = (long long)((M + 255) / 256) * (cout_pad / 128);
    const long long t128 = (long long)((M + 127) / 128) * (cout_pad / 128);
    const double c256 = (double)t256 * 256 * 128 / 1.0;
    const double c128 = (double)t128 * 128 * 128 / 0.85;
    *bm = c256 <= c128 ? 256 : 128;
}

int launch_conv3x3_f16(const ConvArgsH& a, hipStream_t stream) {
    OM_REQUIRE(a.in && a.w && a.scale && a.shift && a.out && a.ticket, OM_EINVAL, "conv3x3 f16: null pointer");
    OM_REQUIRE(conv3x3_f16_supported(a), OM_EINVAL, "conv3x3 f16: layer not supported by the shared-patch kernel");
    OM_REQUIRE(a.in_pix_stride % 8 == 0 && (reinterpret_cast<uintptr_t>(a.in) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(a.w) & 15) == 0,
               OM_EINVAL, "conv3x3 f16: input view / weights must be 16-byte aligned");
    OM_REQUIRE(!(a.res && a.out_f32), OM_EINVAL, "conv3x3 f16: residual only with fp16 output");
    IgemmHParams p;
    p.in = static_cast<const _Float16*>(a.in); p.w = static_cast<const _Float16*>(a.w);
    p.scale = a.scale; p.shift = a.shift; p.res = static_cast<const _Float16*>(a.res); p.out = a.out;
    p.ticket = a.ticket;
    p.H = a.H; p.W = a.W; p.cin = a.cin; p.in_pix_stride = a.in_pix_stride;
    p.Ho = a.H; p.Wo = a.W; p.HoWo = a.H * a.W; p.cout = a.cout;
    p.ks = 3; p.stride = 1; p.pad = 1;
    p.M = a.B * a.H * a.W; p.taps = 9; p.kc = a.cin / 32; p.ksteps = 9 * p.kc;
    p.leaky = a.leaky; p.res_pix_stride = a.res_pix_stride; p.out_pix_stride = a.out_pix_stride;
    p.out_mode = 0; p.up = 1; p.out_f32 = a.out_f32;
    p.n_tiles = 0; p.total_tiles = 0;
    p.total_in_pixels = a.B * a.H * a.W;
    p.w_bytes = (int)(conv_f16_weight_halfs(a.cout_pad, 3, a.cin) * 2);
    const int esz = a.out_f32 ? 4 : 2;
    p.vec_io = ((a.out_pix_stride * esz) % 16 == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0 &&
                (!a.res || (a.res_pix_stride % 8 == 0 && (reinterpret_cast<uintptr_t>(a.res) & 15) == 0)))
                   ? 1 : 0;
    int bm, bn;
    conv3x3_tile_for_f16(p.M, a.cout_pad, a.W, a.cin / 32, &bm, &bn);
    if (bm == 512) {
        const int np = (C3T_BM + 2 * a.W + 2 + 127) / 128;
        if (np <= 5) return launch_c3_tall<5>(p, a.cout_pad, stream);
        if (np == 6) return launch_c3_tall<6>(p, a.cout_pad, stream);
        return launch_c3_tall<7>(p, a.cout_pad, stream);
    }
    if (bn == 64) return launch_c3<128, 64, 64, 32>(p, a.cout_pad, stream);
    if (bm == 256) return launch_c3<256, 128, 128, 64>(p, a.cout_pad, stream);
    return launch_c3<128, 128, 64, 64>(p, a.cout_pad, stream);
}

}  // namespace om
