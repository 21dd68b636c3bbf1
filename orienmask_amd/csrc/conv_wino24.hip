// Winograd F(2x4, 3x3): 2 output rows x 4 output columns per tile (F(2,3) down the rows, F(4,3) along them).
//
//   Y(2x4) = A_y^T [ sum_c (G_y g G_x^T) (.) (B_y^T d B_x) ] A_x        d = 4 x 6 input patch
//
// 24 multiplies per 8 outputs = 3 per output, against 4 for F(2x2,3x3) and 9 for direct convolution: 25 % fewer
// matrix-core flops than conv_wino.hip on the same layers, and a transformed input of 3x (not 4x) the activation.
// Same fused layer (Conv2d 3x3 s1 p1 -> BatchNorm2d(eval) -> LeakyReLU(0.1) (+ residual),
// /root/reference/model/base.py:104-137, model/backbone/darknet.py:14-15) and the same two kernels as conv_wino.hip:
//
//   wino24_input_kernel   V[xi][tile][c], xi = 6 i + j (i: row transform index 0..3, j: column transform index 0..5)
//   wino24_gemm_kernel    24 GEMMs on v_mfma_f32_32x32x2_f32 (exact fp32) with the inverse transform folded into the
//                         per-xi accumulator flush into EIGHT output accumulators (2 x 4 positions), then eight LDS-staged
//                         epilogues.  64 x 64 tile only (8 x 16 output registers per lane: 2 workgroups per CU; measured:
//                         the F(2x2) kernel loses only 4 % going from 3 to 2 workgroups per CU).
//
// Numerics (fp32, measured against float64 direct convolution on a 128-channel layer): max error 1.5e-6 of the tensor's
// scale, against 3.6e-7 for F(2x2) and 2.4e-7 for MKLDNN's direct fp32 -- the F(4,3) matrices carry the constants
// 4, 5, 2 (input), 1/4, 1/6, 1/12, 1/24 (weights, applied on the host in float64) and 2, 4, 8 (output; exact scalings).
// Far inside the 1e-4 parity budget; the full-size F(4x4,3x3) would be at 4.8e-6 and needs 16 output accumulators.
#if (defined(OM_EXP_V_RESIDENT) || defined(OM_EXP_U_RESIDENT)) && !defined(OM_MEASUREMENT_BUILD)
#error "OM_EXP_*_RESIDENT read stale operands (upper-bound experiments): only for ab/ variants (tools/build_variant.sh defines OM_MEASUREMENT_BUILD and never writes orienmask_amd/lib/)"
#endif
#include <cstdlib>

#include "om_common.h"

#ifndef OM_W24_NBUF
#define OM_W24_NBUF 4          // operand ring depth of the split-operand GEMM: 3 steps of LDS-DMA in flight (+2.7 % end to end
                              // over 3 deep, same-box A/B); the fp32 form is built around 3
#endif

namespace om {


typedef float f32x16 __attribute__((ext_vector_type(16)));

// o += a * c, one scalar fused multiply-add per element.  NOT `o += a * c` on the vector types: hipcc packs that into v_pk_fma_f32
// with a source-half selection (op_sel_hi) on the register that holds c, and on gfx950 a packed fp32 instruction with a register
// half-selection returns wrong lanes now and then while another wave on the same SIMD issues wide-K matrix instructions
// (tools/hazard_probe/pk_opsel_repro.hip, profiles/r05_experiments.md 2).  This file is built with -fno-slp-vectorize for the same reason.
__device__ __forceinline__ void axpy16(f32x16& o, const f32x16& a, float c) {
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = __builtin_fmaf(a[r], c, o[r]);
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// SPLIT operands (precision mode "f32_split"): every fp32 operand x of the 24 GEMMs is carried as two fp16
// numbers hi = fp16(x), lo = fp16(x - hi) (x = hi + lo to ~2^-22 |x|, absolute floor 2^-25), and a product is
// hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 with fp32 accumulation: three matrix instructions of 32 cycles for
// 16 channels instead of eight v_mfma_f32_32x32x2_f32 of 64 cycles -- 5.3x the matrix rate of the f32-input pipe at the
// same bytes per operand element (4).  A row of 32 channels (one k-step, 128 bytes) is laid out as two groups of
// [16 hi | 16 lo] halfs, so that 16-byte chunk 2 q + fk of the row is exactly the 8-half MFMA operand of lane half fk for
// q = 0: hi of channels 0-15, 1: lo of 0-15, 2: hi of 16-31, 3: lo of 16-31 -- the LDS image, the LDS-DMA pieces and the
// swizzle are those of the fp32 kernel.
template <bool SPLIT>
__device__ __forceinline__ void wino24_store_v(float* row, int c4, f32x4 v) {
    if constexpr (!SPLIT) {
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(row + c4 * 4));
    } else {
        const f16x4 h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
        const f16x4 l = {(_Float16)(v[0] - (float)h[0]), (_Float16)(v[1] - (float)h[1]), (_Float16)(v[2] - (float)h[2]),
                         (_Float16)(v[3] - (float)h[3])};
        const u32x2 hw = __builtin_bit_cast(u32x2, h), lw = __builtin_bit_cast(u32x2, l);
        // lanes 2k / 2k+1 hold channels 8k'..8k'+3 / +4..+7 of the same tile: the even lane stores the 8 hi halfs, the odd
        // lane the 8 lo halfs, 16 bytes each
        const bool odd = c4 & 1;
        const u32x2 send = odd ? hw : lw;
        const u32x2 recv = {(unsigned)__shfl_xor((int)send[0], 1), (unsigned)__shfl_xor((int)send[1], 1)};
        const u32x4 st = odd ? u32x4{recv[0], recv[1], lw[0], lw[1]} : u32x4{hw[0], hw[1], recv[0], recv[1]};
        float* dst = row + (c4 >> 2) * 16 + (odd ? 8 + ((c4 - 1) & 3) * 2 : (c4 & 3) * 2);
        __builtin_nontemporal_store(st, reinterpret_cast<u32x4*>(dst));
    }
}

// ------------------------------------------------------------------------------------------------
// input transform: rows B_y^T = F(2,3), columns B_x^T = F(4,3) with points {0, +-1, +-2}
// ------------------------------------------------------------------------------------------------
template <bool SPLIT>
__global__ __launch_bounds__(256) void wino24_input_kernel(const float* __restrict__ in, float* __restrict__ V, int H,
                                                           int W, int C, int pix_stride, int TH, int TW, int T) {
    const int c4n = C >> 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const int c4 = (int)(idx % c4n);
    const long long tile = idx / c4n;
    if (tile >= T) return;
    const int b = (int)(tile / (TH * TW));
    const int r = (int)(tile - (long long)b * TH * TW);
    const int ty = r / TW, tx = r - ty * TW;
    const float* base = in + (size_t)b * H * W * pix_stride + c4 * 4;
    const size_t plane = (size_t)T * C;
    float* o = V + (size_t)tile * C;        // this tile's row of C four-byte words in plane 0
    f32x4 d[4][6];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int y = 2 * ty - 1 + i;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int x = 4 * tx - 1 + j;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W)
                v = *reinterpret_cast<const f32x4*>(base + ((size_t)y * W + x) * pix_stride);
            d[i][j] = v;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        // B_y^T d : rows (d0 - d2, d1 + d2, d2 - d1, d1 - d3)
        f32x4 t[6];
#pragma unroll
        for (int j = 0; j < 6; ++j)
            t[j] = i == 0 ? d[0][j] - d[2][j] : i == 1 ? d[1][j] + d[2][j] : i == 2 ? d[2][j] - d[1][j] : d[1][j] - d[3][j];
        // ... B_x : (4 t0 - 5 t2 + t4, -4 t1 - 4 t2 + t3 + t4, 4 t1 - 4 t2 - t3 + t4, -2 t1 - t2 + 2 t3 + t4,
        //            2 t1 - t2 - 2 t3 + t4, 4 t1 - 5 t3 + t5)
        const f32x4 a12 = t[1] + t[2], s12 = t[1] - t[2];           // shared sub-expressions
        const f32x4 a34 = t[3] + t[4], s34 = t[4] - t[3];
        float* op = o + (size_t)(i * 6) * plane;
        wino24_store_v<SPLIT>(op + 0 * plane, c4, 4.f * t[0] - 5.f * t[2] + t[4]);
        wino24_store_v<SPLIT>(op + 1 * plane, c4, a34 - 4.f * a12);
        wino24_store_v<SPLIT>(op + 2 * plane, c4, 4.f * s12 + s34);
        wino24_store_v<SPLIT>(op + 3 * plane, c4, (t[4] - t[2]) + 2.f * (t[3] - t[1]));
        wino24_store_v<SPLIT>(op + 4 * plane, c4, (t[4] - t[2]) - 2.f * (t[3] - t[1]));
        wino24_store_v<SPLIT>(op + 5 * plane, c4, 4.f * t[1] - 5.f * t[3] + t[5]);
    }
}

// ------------------------------------------------------------------------------------------------
// GEMM + inverse transform + epilogue (the slot-pipelined LDS-DMA loop of conv_wino.hip, 24 planes, 8 outputs)
// ------------------------------------------------------------------------------------------------
struct Wino24Params {
    const float* V;       // [24][T][C]
    const float* U;       // [24][cout_pad][C]
    const float* scale;
    const float* shift;
    const float* res;
    float* out;
    int* ticket;
    int T, TH, TW, C, kc;
    int H, W, cout, cout_pad;
    int n_tiles, total_tiles;
    int leaky, res_pix_stride, out_pix_stride, vec_io;
    // stream-K form only
    float* partial;       // [slots][32][256] f32x4: the eight raw accumulators a slot publishes for the tile it shares
    int* flags;           // [slots], zeroed before the launch: slot s has published
    int slots;
    int* status;          // the forward's status word (OM_STATUS_*), nullptr = not reported
};

// SK = false: whole tiles from a dynamic ticket queue.
// SK = true : STREAM-K.  The (tile, plane) units of the launch are dealt out EVENLY to the resident workgroups: with whole
// tiles a launch runs ceil(tiles / 512) rounds although it has work for tiles / 512 of them -- the 34 x 34 layers (616 tiles)
// kept their waves resident for 74 % of the kernel's duration, the 68 x 68 layers for 85 % (rocprofv3 PMC,
// profiles/r02_experiments.md).  Slot s owns units [s U / S, (s + 1) U / S), U = 24 tiles: whole tiles in the middle, possibly
// the HEAD planes of one tile at the end of its range and the TAIL planes of another at its start.  A slot
//   1. computes its head part FIRST and publishes the eight raw output accumulators (128 KiB: write-through stores, every
//      wave drains, one lane raises a flag -- Guideline 16 R1 of cdna_hip_programming.md),
//   2. runs its whole tiles,
//   3. finishes the tile whose head the PREVIOUS slot published long ago: it STARTS from those accumulators and continues
//      with the remaining planes, so every output is the same sequence of fp32 operations as in an unsplit tile -- results do
//      not depend on where a tile was cut (bit-identical to SK = false, batch-size invariant).
// Slot numbers are drawn at start from eight per-XCD counters (slot = xcd + 8 k): slots s and s + slots / n_tiles walk the same
// transformed-input panels at the same time and sit behind the same L2 (slots = 512, n_tiles a power of two; rocprofv3 PMC showed
// 7 reads of V per launch from HBM with one chip-wide counter; +2 % end to end, same-box A/B).  All slots workgroups are resident
// at once (grid = slots = 2 per CU), so a finisher's partner has started or starts without waiting for anyone; the wait is
// bounded.  Needs tiles >= slots (a tile is cut at most once).
template <bool SK, bool SPLIT>
__global__ __launch_bounds__(256, 2) void wino24_gemm_kernel(const Wino24Params p) {
    constexpr int BM = 64, BN = 64, WM = 32, WN = 32;
    constexpr int NWN = BN / WN;
    constexpr int A_CH = BM / 32, B_CH = BN / 32, NP = A_CH + B_CH;
    constexpr int CH = BN / 4, RP = 256 / CH;
    constexpr int NBUF = SPLIT ? OM_W24_NBUF : 3;
    constexpr int PF = NBUF - 1;               // k-steps the LDS-DMA runs ahead
    __shared__ f32x4 smem[NBUF * (BM + BN) * 8 + 1];     // one LDS object (see conv_igemm.hip)
    int* const s_ticket = reinterpret_cast<int*>(smem + NBUF * (BM + BN) * 8);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave / NWN, wn = wave % NWN;
    const int lrow = tid >> 3, lcol = tid & 7;
    const int scol = lcol ^ ((lrow >> 1) & 7);
    const int fi = lane & 31, fk = lane >> 5;
    const int fsw = (fi >> 1) & 7;
    const size_t v_plane = (size_t)p.T * p.C, u_plane = (size_t)p.cout_pad * p.C;

    // ---- stream-K: my slot and its unit range
    int slot = 0, t_lead = 0, x_lead = 0, t_trail = 0, x_trail = 0, t_full0 = 0, n_full = 0, nseg = 0;
    if constexpr (SK) {
        // slot numbers congruent to the XCD id mod 8: slots s and s + slots / n_tiles (which walk the same panels at the same
        // time) then sit behind the same L2 whenever slots / n_tiles is a multiple of 8
        if (tid == 0) {
            int x;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
            x &= 7;
            // every XCD's counter hands out slots / 8 numbers; a workgroup beyond its XCD's share (placement is not promised to
            // be even) takes one from another XCD's counter: slots numbers for slots workgroups, each drawn exactly once
            int sl = 0;
            for (int h = 0; h < 8; ++h) {
                const int y = (x + h) & 7;
                const int k = atomicAdd(p.ticket + y, 1);
                if (k < (p.slots >> 3)) { sl = y + 8 * k; break; }
            }
            *s_ticket = sl;
        }
        __syncthreads();
        slot = __builtin_amdgcn_readfirstlane(*s_ticket);
        const long long U = (long long)p.total_tiles * 24;
        const long long u0 = U * slot / p.slots, u1 = U * (slot + 1) / p.slots;
        t_lead = (int)(u0 / 24); x_lead = (int)(u0 - (long long)t_lead * 24);      // finish tile t_lead from plane x_lead on
        t_trail = (int)(u1 / 24); x_trail = (int)(u1 - (long long)t_trail * 24);   // head planes [0, x_trail) of tile t_trail
        t_full0 = t_lead + (x_lead != 0);
        n_full = t_trail - t_full0;
        nseg = (x_trail != 0) + n_full + (x_lead != 0);
    }

    int xcd = 0, q_hops = 0;        // whole-tile form: my XCD's queue first, then the others' in ring order
    if constexpr (!SK) {
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcd));
        xcd &= 7;
    }
    for (int seg = 0;; ++seg) {
        int tile, xb = 0, xe = 24, mode = 0;     // mode 1: produce the head planes of a tile, 2: finish from the partner's
        if constexpr (SK) {
            // order: head part (published early) -> whole tiles -> tail part (its partner published long ago)
            if (seg >= nseg) break;
            const int has_trail = x_trail != 0;
            if (has_trail && seg == 0) { tile = t_trail; xe = x_trail; mode = 1; }
            else if (seg - has_trail < n_full) { tile = t_full0 + seg - has_trail; }
            else { tile = t_lead; xb = x_lead; mode = 2; }
        } else {
            // the ticket's round trip is paid by lane 0's wave only: the others wait at a raw barrier, which (unlike
            // __syncthreads) does not make them drain the previous tile's output stores first.  (Drawing the NEXT ticket at the
            // start of a tile hides the round trip too, but pins the last partial round of tiles to the workgroups that started
            // first: the 34 x 34 layers went from 0.33 to 0.44 ms.)
            // EIGHT queues, one per XCD (ticket words 0..7): XCD x owns the M panels [m_tiles x / 8, m_tiles (x + 1) / 8) and its
            // workgroups draw (panel, N tile) pairs N fastest, so the n_tiles workgroups that share a transformed-input panel run
            // at the same time behind the SAME L2 -- with one queue for the chip they sat on different XCDs and every one of them
            // fetched the panel from HBM (rocprofv3 PMC: 4.5 reads of V per launch).  A workgroup whose queue is empty moves
            // on to the next XCD's (never back), so the last round still balances over the whole chip.  Placement only:
            // results do not depend on which workgroup computes a tile.
            if (tid == 0) {
                const int m_tiles_all = p.total_tiles / p.n_tiles;
                int t = -1;
                while (q_hops < 8) {
                    const int q = (xcd + q_hops) & 7;
                    const int pm0 = (int)((long long)m_tiles_all * q >> 3), pm1 = (int)((long long)m_tiles_all * (q + 1) >> 3);
                    const int v = atomicAdd(p.ticket + q, 1);
                    if (v < (pm1 - pm0) * p.n_tiles) { t = pm0 * p.n_tiles + v; break; }
                    ++q_hops;
                }
                *s_ticket = t;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            tile = *s_ticket;
            if (tile < 0) break;
            tile = __builtin_amdgcn_readfirstlane(tile);
        }
        const int ksteps = (xe - xb) * p.kc;
        // ticket queue: N fastest (the n_tiles workgroups that draw consecutive tickets share a transformed-input panel while it
        // is hot).  Stream-K region: M fastest -- a slot walks consecutive M tiles of one N tile, and the slots working on the
        // other N tiles of the same panels run at the same time, slots / n_tiles further on (N fastest there re-read every panel
        // n_tiles times a whole tile-time apart: +4 % on the whole set of layers).
        const int m_tiles = p.total_tiles / p.n_tiles;
        const int tile_n = SK ? tile / m_tiles : tile % p.n_tiles;
        const int tile_m = SK ? tile - tile_n * m_tiles : tile / p.n_tiles;
        const int m0 = tile_m * BM, n0 = tile_n * BN;
        const int rows_valid = min(BM, p.T - m0);

        const int voff0 = (lrow * p.C + scol * 4) * 4;
        const int voff_rows32 = 32 * p.C * 4;

        int n_xi = xb, n_cc = 0;     // (transform index, channel chunk) of the step being fetched
        auto advance = [&]() {
            if (++n_cc == p.kc) { n_cc = 0; ++n_xi; }
        };
        auto issue_piece = [&](int piece, int buf, bool live) {
#ifdef OM_EXP_V_RESIDENT        // upper-bound experiment (wrong numerics): every workgroup reads the SAME M panel of V -> V stays in L2
            const float* abase = p.V + (size_t)n_xi * v_plane;
#else
            const float* abase = p.V + (size_t)n_xi * v_plane + (size_t)m0 * p.C;
#endif
#ifdef OM_EXP_U_RESIDENT        // ... and / or the same N tile of U
            const float* bbase = p.U + (size_t)n_xi * u_plane;
#else
            const float* bbase = p.U + (size_t)n_xi * u_plane + (size_t)n0 * p.C;
#endif
            const auto rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(abase), 0,
                                                               live ? rows_valid * p.C * 4 : 0, 0x00020000);
            const auto rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bbase), 0, live ? BN * p.C * 4 : 0,
                                                               0x00020000);
            f32x4* dst = smem + buf * (BM + BN) * 8 + wave_u * 64;
            const int soff = n_cc * 128;
            if (piece < A_CH) {
                const int vo = voff0 + piece * voff_rows32;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(dst + piece * 256), 16, vo, soff, 0, 0);
            } else {
                const int vo = voff0 + (piece - A_CH) * voff_rows32;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(dst + BM * 8 + (piece - A_CH) * 256), 16, vo, soff,
                                                         0, 0);
            }
        };

        f32x16 acc, outa[2][4];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[r] = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) outa[e >> 2][e & 3][r] = 0.f;
        }
        if constexpr (SK) {
            if (mode == 2) {
                // the partner (slot - 1) published this tile's head planes as its first action
                if (tid == 0) {
                    // The partner drew its slot number BEFORE this workgroup drew its own and publishes as its first action, within
                    // one tile's time; with other streams' kernels on the chip (pipeline.InFlightPipeline) it may be queued behind
                    // them, so the wait is long -- 2^24 polls, several seconds -- but bounded: a fault must not hang the device.  Giving
                    // up is REPORTED (OM_STATUS_SK_TIMEOUT in the forward's status word; the host raises), never silent.
                    int seen = 0;
                    for (int spins = 0; spins < (1 << 24); ++spins) {
                        seen = __hip_atomic_load(p.flags + slot - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (seen) break;
                        __builtin_amdgcn_s_sleep(16);
                    }
                    if (!seen && p.status) atomicOr(p.status, OM_STATUS_SK_TIMEOUT);
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                }
                __syncthreads();
                const auto rs_part = __builtin_amdgcn_make_buffer_rsrc(p.partial, 0, 0x7FFFFFFF, 0x00020000);
                const int pbase = ((slot - 1) * 32 * 256 + tid) * 16;
#pragma unroll
                for (int e = 0; e < 8; ++e)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_part, pbase, (e * 4 + g) * 256 * 16, 16));
                        outa[e >> 2][e & 3][4 * g] = v[0]; outa[e >> 2][e & 3][4 * g + 1] = v[1];
                        outa[e >> 2][e & 3][4 * g + 2] = v[2]; outa[e >> 2][e & 3][4 * g + 3] = v[3];
                    }
            }
        }

        const f32x4* fragA = smem + (wm * WM + fi) * 8;
        const f32x4* fragB = smem + BM * 8 + (wn * WN + fi) * 8;
        f32x4 ca, cb, na, nb;
        auto read_frags = [&](f32x4& fa, f32x4& fb, int buf, int q) {
            const int ch = (2 * q + fk) ^ fsw;
            const int bo = buf * (BM + BN) * 8;
            fa = fragA[bo + ch];
            fb = fragB[bo + ch];
        };

#pragma unroll
        for (int piece = 0; piece < NP; ++piece) issue_piece(piece, 0, true);
        advance();
#pragma unroll
        for (int piece = 0; piece < NP; ++piece) issue_piece(piece, 1, 1 < ksteps);
        advance();                                          // fetch state = step 2
        if constexpr (PF > 2) {
#pragma unroll
            for (int piece = 0; piece < NP; ++piece) issue_piece(piece, 2, 2 < ksteps);
            advance();
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"i"((PF - 1) * NP) : "memory");   // step 0 landed, the later ones may still fly
        __builtin_amdgcn_s_barrier();
        read_frags(ca, cb, 0, 0);
        int xi = xb, cc = 0;
        int buf = 0;
        f32x4 ca1, cb1;                                      // SPLIT: lo halves of channels 0-15 of the current step
        if constexpr (SPLIT) read_frags(ca1, cb1, 0, 1);
        for (int s = 0; s < ksteps; ++s) {
            const int buf1 = buf == NBUF - 1 ? 0 : buf + 1;      // step s+1
            const int buf2 = buf1 == NBUF - 1 ? 0 : buf1 + 1;    // step s+2 (last read during step s-1)
            const bool live2 = s + 2 < ksteps;
            if constexpr (SPLIT) {
                // six matrix instructions per k-step: per group of 16 channels hi*lo + lo*hi + hi*hi (fp32 accumulate)
                f32x4 a2, b2, a3, b3;
                read_frags(a2, b2, buf, 2);
                read_frags(a3, b3, buf, 3);
                const int bufp = buf + PF >= NBUF ? buf + PF - NBUF : buf + PF;      // step s+PF: last read during step s-1
                const bool livep = s + PF < ksteps;
#pragma unroll
                for (int piece = 0; piece < NP; ++piece) issue_piece(piece, bufp, livep);
                const f16x8 ah = __builtin_bit_cast(f16x8, ca), al = __builtin_bit_cast(f16x8, ca1);
                const f16x8 bh = __builtin_bit_cast(f16x8, cb), bl = __builtin_bit_cast(f16x8, cb1);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, al, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, ah, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ah, acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                const f16x8 ah2 = __builtin_bit_cast(f16x8, a2), al2 = __builtin_bit_cast(f16x8, a3);
                const f16x8 bh2 = __builtin_bit_cast(f16x8, b2), bl2 = __builtin_bit_cast(f16x8, b3);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh2, al2, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl2, ah2, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh2, ah2, acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                // step s+1's operands have landed (only this step's NP pieces may still fly); my reads of `buf` are done
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"i"((PF - 1) * NP) : "memory");
                __builtin_amdgcn_s_barrier();
                read_frags(ca, cb, buf1, 0);
                read_frags(ca1, cb1, buf1, 1);
                __builtin_amdgcn_sched_barrier(0);
            } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int sl = q * 4 + t;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cb[t], ca[t], acc, 0, 0, 0);
                    if (t == 1 && q < 3) read_frags(na, nb, buf, q + 1);
                    if (sl < NP) issue_piece(sl, buf2, live2);
                    if (sl == 12) read_frags(na, nb, buf1, 0);
                    if (sl == 11) {
                        // everything older than this step's NP pieces has landed = the operands of step s+1;
                        // all my reads of the current buffer are done (lgkmcnt) -> raw barrier, no compiler fence
                        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"i"(NP) : "memory");
                        __builtin_amdgcn_s_barrier();
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                ca = na;
                cb = nb;
            }
            }
            buf = buf1;
            advance();
            if (++cc == p.kc) {
                // flush M_xi into the eight outputs: Y[py][px] += A_y^T[py][i] * A_x^T[px][j] * M,  xi = 6 i + j
                //   A_y^T = [[1, 1, 1, 0], [0, 1, -1, -1]]
                //   A_x^T = [[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]]
                const int i = xi / 6, j = xi - 6 * i;
                const float cy0 = i < 3 ? 1.f : 0.f, cy1 = i == 0 ? 0.f : (i == 1 ? 1.f : -1.f);
                const float sg = (j & 1) ? 1.f : -1.f;                   // columns 2 and 4 alternate in sign
                float cx[4];
                cx[0] = j < 5 ? 1.f : 0.f;
                cx[1] = (j == 0 || j == 5) ? 0.f : (j < 3 ? sg : 2.f * sg);
                cx[2] = (j == 0 || j == 5) ? 0.f : (j < 3 ? 1.f : 4.f);
                cx[3] = j == 0 ? 0.f : (j == 5 ? 1.f : (j < 3 ? sg : 8.f * sg));
#pragma unroll
                for (int px = 0; px < 4; ++px) {
                    const float c0 = cy0 * cx[px], c1 = cy1 * cx[px];
                    if (c0 != 0.f) axpy16(outa[0][px], acc, c0);
                    if (c1 != 0.f) axpy16(outa[1][px], acc, c1);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                cc = 0;
                ++xi;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        if constexpr (SK) {
            if (mode == 1) {
                // publish the raw accumulators: write-through (sc1) 16-byte stores, every wave drains, one lane raises the flag
                const auto rs_part = __builtin_amdgcn_make_buffer_rsrc(p.partial, 0, 0x7FFFFFFF, 0x00020000);
                const int pbase = (slot * 32 * 256 + tid) * 16;
#pragma unroll
                for (int e = 0; e < 8; ++e)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x16& o = outa[e >> 2][e & 3];
                        const f32x4 v = {o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v),
                                                               rs_part, pbase, (e * 4 + g) * 256 * 16, 16);
                    }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) {
                    // agent-scope release before the flag (MI355X_MICROARCH.md, inter-workgroup visibility): the sc1 stores above are
                    // write-through and drained, the fence also writes back anything this L2 still holds dirty; the inline-asm wait
                    // keeps the compiler from dropping the fence's vmcnt(0) (its scoreboard is provably empty here)
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __hip_atomic_store(p.flags + slot, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                continue;
            }
        }

        // ---- epilogue: row geometry once per tile; per pass of two output positions the residual loads are issued BEFORE the
        // LDS staging (they fly during it), C tiles [position][row][channel chunk ^ (row & 7)] go through LDS and leave as 16-byte
        // rows; the barriers wait for LDS only (a __syncthreads() would also drain the stores: ~2 us of HBM write latency per
        // pass).  Measured in-network: 15.0 -> 14.3 ms over the 30 layers (profiles/r02_experiments.md).
        f32x4* sC = smem;
        const int n4 = tid % CH, r0 = tid / CH;
        const int n = n0 + n4 * 4;
        const f32x4 sc = *reinterpret_cast<const f32x4*>(p.scale + n);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(p.shift + n);
        const int nvalid = p.cout - n;
        const bool vec = p.vec_io && nvalid >= 4;
        int pixb[BM / RP];
        unsigned okb[BM / RP];           // bit py * 4 + px: that output position of the row's tile exists
#pragma unroll
        for (int ps = 0; ps < BM / RP; ++ps) {
            const int m = m0 + ps * RP + r0;
            const bool mok = m < p.T && nvalid > 0;
            const int mm = mok ? m : 0;
            const int bi = mm / (p.TH * p.TW);
            const int rr = mm - bi * p.TH * p.TW;
            const int ty = rr / p.TW, tx = rr - ty * p.TW;
            pixb[ps] = (bi * p.H + 2 * ty) * p.W + 4 * tx;
            unsigned ok = 0;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                ok |= ((mok && 2 * ty + (q >> 2) < p.H && 4 * tx + (q & 3) < p.W) ? 1u : 0u) << q;
            okb[ps] = ok;
        }
        constexpr int PPP = SK ? 1 : 2;       // output positions per pass (the stream-K form has fewer registers to spare)
        // SPLIT: range guard of the hi/lo representation (conv_igemm_split.hip: split_epilogue) -- a transformed input beyond fp16's
        // range makes every output of its tile row NaN; v * 0 is NaN exactly for non-finite v
        float nonfinite = 0.f;
#pragma unroll
        for (int pass = 0; pass < (8 + PPP - 1) / PPP; ++pass) {
            f32x4 rres[PPP][BM / RP];
            if (p.res && vec) {
#pragma unroll
                for (int e = 0; e < PPP; ++e) {
                    const int pq = pass * PPP + e;
                    if (pq >= 8) continue;
#pragma unroll
                    for (int ps = 0; ps < BM / RP; ++ps) {
                        const size_t pix = (size_t)(pixb[ps] + (pq >> 2) * p.W + (pq & 3));
                        const float* rp = p.res + (((okb[ps] >> pq) & 1u) ? pix * p.res_pix_stride + n : 0);
                        rres[e][ps] = *reinterpret_cast<const f32x4*>(rp);
                    }
                }
            }
            const int ml = wm * WM + fi;
#pragma unroll
            for (int e = 0; e < PPP; ++e) {
                const int pq = pass * PPP + e;
                if (pq >= 8) continue;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c4 = (wn * WN) / 4 + 2 * g + fk;
                    const f32x16& o = outa[pq >> 2][pq & 3];
                    f32x4 v = {o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]};
                    sC[e * BM * CH + ml * CH + (c4 ^ (ml & 7))] = v;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int e = 0; e < PPP; ++e) {
                const int pq = pass * PPP + e;
                if (pq >= 8) continue;
#pragma unroll
                for (int ps = 0; ps < BM / RP; ++ps) {
                    if (!((okb[ps] >> pq) & 1u)) continue;
                    const int mr = ps * RP + r0;
                    const size_t pix = (size_t)(pixb[ps] + (pq >> 2) * p.W + (pq & 3));
                    f32x4 v = sC[e * BM * CH + mr * CH + (n4 ^ (mr & 7))];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float tv = fmaf(v[k], sc[k], sh[k]);
                        v[k] = p.leaky ? (tv > 0.f ? tv : tv * 0.1f) : tv;
                        if constexpr (SPLIT) nonfinite = fmaf(tv, 0.f, nonfinite);
                    }
                    float* o = p.out + pix * p.out_pix_stride + n;
                    if (vec) {
                        if (p.res) v += rres[e][ps];
                        *reinterpret_cast<f32x4*>(o) = v;
                    } else {
                        const float* rp = p.res ? p.res + pix * p.res_pix_stride + n : nullptr;
                        for (int k = 0; k < 4 && k < nvalid; ++k) o[k] = rp ? v[k] + rp[k] : v[k];
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        if constexpr (SPLIT) {
            if (p.status && nonfinite != nonfinite) atomicOr(p.status, OM_STATUS_SPLIT_RANGE);
        }
    }
}

size_t wino24_scratch_floats(int B, int H, int W, int C) {
    return (size_t)24 * B * ((H + 1) / 2) * ((W + 3) / 4) * C;
}

// a.w must point at the transformed weights U [24][cout_pad][cin]; scratch holds V (wino24_scratch_floats).
// a.split: U and V are hi/lo fp16 pairs in the [16 hi | 16 lo] row layout described at the top (same sizes); a.scale
// then already carries the per-output-channel power of two that the packer multiplied U's rows with.
int launch_conv_winograd24(const ConvArgs& a, float* scratch, hipStream_t stream) {
    OM_REQUIRE(a.in && a.w && a.scale && a.shift && a.out && scratch && a.ticket, OM_EINVAL, "winograd24: null pointer");
    OM_REQUIRE(a.ks == 3 && a.stride == 1 && a.out_mode == 0, OM_EINVAL, "winograd24: 3x3 stride-1 NHWC layers only");
    OM_REQUIRE(a.cin % 32 == 0 && a.cin >= 32 && a.cout_pad % 64 == 0, OM_EINVAL, "winograd24: cin=%d cout_pad=%d", a.cin,
               a.cout_pad);
    OM_REQUIRE(a.in_pix_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(a.in) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(a.w) & 15) == 0 && (reinterpret_cast<uintptr_t>(scratch) & 15) == 0,
               OM_EINVAL, "winograd24: operands must be 16-byte aligned");
    const int TH = (a.H + 1) / 2, TW = (a.W + 3) / 4;
    const long long T = (long long)a.B * TH * TW;
    OM_REQUIRE(T < (1ll << 31) && 24 * T * a.cin < (1ll << 40), OM_EINVAL, "winograd24: problem too large");
    const long long threads = T * (a.cin / 4);
    if (a.split)
        hipLaunchKernelGGL(wino24_input_kernel<true>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, a.in,
                           scratch, a.H, a.W, a.cin, a.in_pix_stride, TH, TW, (int)T);
    else
        hipLaunchKernelGGL(wino24_input_kernel<false>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, a.in,
                           scratch, a.H, a.W, a.cin, a.in_pix_stride, TH, TW, (int)T);
    OM_CHECK_HIP(hipGetLastError());
    if (a.mid_event) OM_CHECK_HIP(hipEventRecord(a.mid_event, stream));
    Wino24Params p;
    p.V = scratch; p.U = a.w; p.scale = a.scale; p.shift = a.shift; p.res = a.res; p.out = a.out; p.ticket = a.ticket;
    p.T = (int)T; p.TH = TH; p.TW = TW; p.C = a.cin; p.kc = a.cin / 32;
    p.H = a.H; p.W = a.W; p.cout = a.cout; p.cout_pad = a.cout_pad;
    p.leaky = a.leaky; p.res_pix_stride = a.res_pix_stride; p.out_pix_stride = a.out_pix_stride;
    p.vec_io = (a.out_pix_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0 &&
                (!a.res || (a.res_pix_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(a.res) & 15) == 0)))
                   ? 1 : 0;
    const int m_tiles = (int)((T + 63) / 64);
    p.n_tiles = a.cout_pad / 64;
    const long long total = (long long)m_tiles * p.n_tiles;
    OM_REQUIRE(total > 0 && total < (1ll << 31), OM_EINVAL, "winograd24: %lld tiles out of range", total);
    p.total_tiles = (int)total;
    const long long grid = total < 512 ? total : 512;        // 2 workgroups per CU (register-bound)
    p.partial = a.sk_partial; p.flags = a.ticket + SK_FLAG_OFF; p.slots = (int)grid; p.status = a.status;
    // Stream-K only where the last partial round hurts most (fewer than two tiles per slot: the 34 x 34 layers, 616 tiles,
    // -9 %).  A static split is only as fast as the slowest workgroup -- dealt out statically, whole layers ran 2-4 % slower
    // than from the queue -- so from two tiles per slot on the two effects cancel (68 x 68: +-0, 136 x 136: +1-2 %), and a
    // hybrid (bulk from the queue, only the last round cut evenly) lost to both (profiles/r02_experiments.md).
    const bool sk = a.sk_partial && total >= 512 && total < 2 * 512;
    if (a.split) {
        if (sk) hipLaunchKernelGGL((wino24_gemm_kernel<true, true>), dim3((unsigned)grid), dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((wino24_gemm_kernel<false, true>), dim3((unsigned)grid), dim3(256), 0, stream, p);
    } else {
        if (sk) hipLaunchKernelGGL((wino24_gemm_kernel<true, false>), dim3((unsigned)grid), dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((wino24_gemm_kernel<false, false>), dim3((unsigned)grid), dim3(256), 0, stream, p);
    }
    OM_CHECK_HIP(hipGetLastError());
    return OM_OK;
}

}  // namespace om
