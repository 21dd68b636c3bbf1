// Winograd F(2x2, 3x3) convolution for gfx950 (stride-1 3x3 layers, 89 % of the forward's FLOPs).
//
//   Y = A^T [ sum_c (G g G^T) (.) (B^T d B) ] A        (Lavin & Gray 2015; 2.25x fewer multiplies)
//
// Same fused layer as conv_igemm.hip -- Conv2d(3x3, s1, p1) -> BatchNorm2d(eval) -> LeakyReLU(0.1)
// (+ residual), /root/reference/model/base.py:104-137, /root/reference/model/backbone/darknet.py:14-15 --
// evaluated in the transform domain:
//
//   wino_input_kernel   V[xi][tile][c] = (B^T d B)[xi] for every 4x4 input patch d (one patch per 2x2
//                       output tile, zero padded); HBM-bound (reads X once, writes 4x its size).
//   wino_gemm_kernel    16 independent GEMMs M_xi = V_xi (tiles x C) . U_xi (C x cout) on the f32 matrix
//                       cores (v_mfma_f32_32x32x2_f32, exact fp32), with the inverse transform A^T M A
//                       folded into the accumulator flush after each xi, so M never touches memory;
//                       then the usual scale/shift/LeakyReLU/residual epilogue through LDS, four times
//                       (one per output position of the 2x2 tile).
//   U = G g G^T is computed once on the host in float64 (orienmask_amd/pack.py).
//
// The GEMM kernel is the slot-pipelined LDS-DMA loop of conv_igemm.hip with k = (xi, c): operands are
// dense [rows][C] planes, so a DMA piece needs no address arithmetic at all (constant voffset per lane,
// the channel chunk goes in the scalar offset, rows past the end are cut by the descriptor's size).
// Numerics: F(2x2,3x3) in fp32 adds a few 1e-7 of relative error over direct convolution (the only
// non-trivial constants are the 1/2 in G, applied in float64) -- far inside the 1e-4 parity budget.
#include <cstdlib>

#include "om_common.h"

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
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// ------------------------------------------------------------------------------------------------
// input transform
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wino_input_kernel(const float* __restrict__ in, float* __restrict__ V, int H,
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
    f32x4 d[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int y = 2 * ty - 1 + i;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = 2 * tx - 1 + j;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W)
                v = *reinterpret_cast<const f32x4*>(base + ((size_t)y * W + x) * pix_stride);
            d[i][j] = v;
        }
    }
    // B^T d : rows (d0 - d2, d1 + d2, d2 - d1, d1 - d3)
    f32x4 t[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        t[0][j] = d[0][j] - d[2][j];
        t[1][j] = d[1][j] + d[2][j];
        t[2][j] = d[2][j] - d[1][j];
        t[3][j] = d[1][j] - d[3][j];
    }
    const size_t plane = (size_t)T * C;
    float* o = V + (size_t)tile * C + c4 * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        *reinterpret_cast<f32x4*>(o + (size_t)(i * 4 + 0) * plane) = t[i][0] - t[i][2];
        *reinterpret_cast<f32x4*>(o + (size_t)(i * 4 + 1) * plane) = t[i][1] + t[i][2];
        *reinterpret_cast<f32x4*>(o + (size_t)(i * 4 + 2) * plane) = t[i][2] - t[i][1];
        *reinterpret_cast<f32x4*>(o + (size_t)(i * 4 + 3) * plane) = t[i][1] - t[i][3];
    }
}

// ------------------------------------------------------------------------------------------------
// GEMM + inverse transform + epilogue
// ------------------------------------------------------------------------------------------------
struct WinoParams {
    const float* X;       // fused variant: NHWC input view [B,H,W,C] (pixel stride in_pix_stride)
    int in_pix_stride, total_in_pixels;
    const float* V;       // [16][T][C]
    const float* U;       // [16][cout_pad][C]
    const float* scale;
    const float* shift;
    const float* res;
    float* out;
    int* ticket;
    int T, TH, TW, C, kc;
    int H, W, cout, cout_pad;
    int n_tiles, total_tiles;
    int leaky, res_pix_stride, out_pix_stride, vec_io;
    // stream-K form only (see conv_wino24.hip)
    float* partial;       // [slots][4 * BM * BN] floats: the four raw output accumulators a slot publishes
    int* flags;           // [slots], zeroed before the launch
    int slots;
};

// SK = true: the stream-K form of conv_wino24.hip with units = (tile, plane), 16 planes per tile (bit-identical results).
template <int BM, int BN, int WM, int WN, bool SK>
__global__ __launch_bounds__(256) void wino_gemm_kernel(const WinoParams p) {
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int NWN = BN / WN;
    constexpr int A_CH = BM / 32, B_CH = BN / 32, NP = A_CH + B_CH;
    constexpr int CH = BN / 4, RP = 256 / CH;
    static_assert((BM / WM) * (BN / WN) == 4, "four waves per workgroup");
    constexpr int NBUF = 3;         // operand ring: the DMA runs TWO k-steps ahead (a step is only 16*TM*TN MFMAs, far
                                    // shorter than an HBM round trip), waited for with a counted vmcnt(NP)
    static_assert(BM * BN <= NBUF * (BM + BN) * 32, "C tile must fit in the operand buffers");
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

    int slot = 0, t_lead = 0, x_lead = 0, t_trail = 0, x_trail = 0, t_full0 = 0, n_full = 0, nseg = 0;
    if constexpr (SK) {
        if (tid == 0) *s_ticket = atomicAdd(p.ticket, 1);
        __syncthreads();
        slot = __builtin_amdgcn_readfirstlane(*s_ticket);
        const long long U = (long long)p.total_tiles * 16;
        const long long u0 = U * slot / p.slots, u1 = U * (slot + 1) / p.slots;
        t_lead = (int)(u0 / 16); x_lead = (int)(u0 - (long long)t_lead * 16);
        t_trail = (int)(u1 / 16); x_trail = (int)(u1 - (long long)t_trail * 16);
        t_full0 = t_lead + (x_lead != 0);
        n_full = t_trail - t_full0;
        nseg = (x_trail != 0) + n_full + (x_lead != 0);
    }

    for (int seg = 0;; ++seg) {
        int tile, xb = 0, xe = 16, mode = 0;      // mode 1: produce the head planes of a tile, 2: finish from the partner's
        if constexpr (SK) {
            if (seg >= nseg) break;
            const int has_trail = x_trail != 0;
            if (has_trail && seg == 0) { tile = t_trail; xe = x_trail; mode = 1; }
            else if (seg - has_trail < n_full) { tile = t_full0 + seg - has_trail; }
            else { tile = t_lead; xb = x_lead; mode = 2; }
        } else if (p.ticket) {
            // raw barrier: only lane 0's wave pays the ticket's round trip, nobody drains the previous tile's stores
            if (tid == 0) *s_ticket = atomicAdd(p.ticket, 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            tile = *s_ticket;
            if (tile >= p.total_tiles) break;
        } else {
            tile = blockIdx.x;
            if (tile >= p.total_tiles) break;
        }
        tile = __builtin_amdgcn_readfirstlane(tile);
        const int ksteps = (xe - xb) * p.kc;
        const int m_tiles = p.total_tiles / p.n_tiles;
        const int tile_n = SK ? tile / m_tiles : tile % p.n_tiles;
        const int tile_m = SK ? tile - tile_n * m_tiles : tile / p.n_tiles;
        const int m0 = tile_m * BM, n0 = tile_n * BN;
        const int rows_valid = min(BM, p.T - m0);

        // per-lane byte offset of row 0's chunk; piece j adds 32 * j rows (kept as plain ints: passing an
        // element of a captured array to the LDS-DMA builtin makes hipcc drop the kernel's host stub)
        const int voff0 = (lrow * p.C + scol * 4) * 4;
        const int voff_rows32 = 32 * p.C * 4;

        int n_xi = xb, n_cc = 0;     // (transform index, channel chunk) of the step being fetched
        auto advance = [&]() {
            if (++n_cc == p.kc) { n_cc = 0; ++n_xi; }
        };
        auto issue_piece = [&](int piece, int buf, bool live) {
            // rows beyond the valid range (M tail, or a dead prefetch) are cut by num_records -> zeros
            const float* abase = p.V + (size_t)n_xi * v_plane + (size_t)m0 * p.C;
            const float* bbase = p.U + (size_t)n_xi * u_plane + (size_t)n0 * p.C;
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

        f32x16 acc[TM][TN], outa[2][2][TM][TN];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    acc[a][b][r] = 0.f;
                    outa[0][0][a][b][r] = 0.f; outa[0][1][a][b][r] = 0.f;
                    outa[1][0][a][b][r] = 0.f; outa[1][1][a][b][r] = 0.f;
                }
        if constexpr (SK) {
            if (mode == 2) {
                if (tid == 0) {
                    for (int spins = 0; spins < (1 << 22) &&
                                        __hip_atomic_load(p.flags + slot - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0; ++spins)
                        __builtin_amdgcn_s_sleep(16);
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                }
                __syncthreads();
                const auto rs_part = __builtin_amdgcn_make_buffer_rsrc(p.partial, 0, 0x7FFFFFFF, 0x00020000);
                const int pbase = ((slot - 1) * (16 * TM * TN) * 256 + tid) * 16;
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int a = 0; a < TM; ++a)
#pragma unroll
                        for (int b = 0; b < TN; ++b)
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                    rs_part, pbase, (((e * TM + a) * TN + b) * 4 + g) * 256 * 16, 16));
                                f32x16& o = outa[e >> 1][e & 1][a][b];
                                o[4 * g] = v[0]; o[4 * g + 1] = v[1]; o[4 * g + 2] = v[2]; o[4 * g + 3] = v[3];
                            }
            }
        }

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
#pragma unroll
        for (int piece = 0; piece < NP; ++piece) issue_piece(piece, 1, 1 < ksteps);
        advance();                                          // fetch state = step 2
        asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NP) : "memory");   // step 0 landed, step 1 may still fly
        __builtin_amdgcn_s_barrier();
        read_frags(ca, cb, 0, 0);
        int xi = xb, cc = 0;
        int buf = 0;
        for (int s = 0; s < ksteps; ++s) {
            const int buf1 = buf == NBUF - 1 ? 0 : buf + 1;      // step s+1
            const int buf2 = buf1 == NBUF - 1 ? 0 : buf1 + 1;    // step s+2 (last read during step s-1)
            const bool live2 = s + 2 < ksteps;
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
                    if (slot < NP) issue_piece(slot, buf2, live2);
                    if (slot == 12) read_frags(na, nb, buf1, 0);
                    if (slot == 11) {
                        // everything older than this step's NP pieces has landed = the operands of step s+1;
                        // all my reads of the current buffer are done (lgkmcnt) -> raw barrier, no compiler fence
                        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"i"(NP) : "memory");
                        __builtin_amdgcn_s_barrier();
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int a = 0; a < TM; ++a) ca[a] = na[a];
#pragma unroll
                for (int b = 0; b < TN; ++b) cb[b] = nb[b];
            }
            buf = buf1;
            advance();
            if (++cc == p.kc) {
                // flush M_xi into the four outputs: Y[p][q] += A^T[p][i] * A^T[q][j] * M,  xi = 4 i + j,
                // A^T = [[1, 1, 1, 0], [0, 1, -1, -1]]
                const int i = xi >> 2, j = xi & 3;
                const float ci0 = i < 3 ? 1.f : 0.f, ci1 = i == 0 ? 0.f : (i == 1 ? 1.f : -1.f);
                const float cj0 = j < 3 ? 1.f : 0.f, cj1 = j == 0 ? 0.f : (j == 1 ? 1.f : -1.f);
                const float c00 = ci0 * cj0, c01 = ci0 * cj1, c10 = ci1 * cj0, c11 = ci1 * cj1;
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b) {
                        if (c00 != 0.f) axpy16(outa[0][0][a][b], acc[a][b], c00);
                        if (c01 != 0.f) axpy16(outa[0][1][a][b], acc[a][b], c01);
                        if (c10 != 0.f) axpy16(outa[1][0][a][b], acc[a][b], c10);
                        if (c11 != 0.f) axpy16(outa[1][1][a][b], acc[a][b], c11);
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
                    }
                cc = 0;
                ++xi;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        if constexpr (SK) {
            if (mode == 1) {
                const auto rs_part = __builtin_amdgcn_make_buffer_rsrc(p.partial, 0, 0x7FFFFFFF, 0x00020000);
                const int pbase = (slot * (16 * TM * TN) * 256 + tid) * 16;
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int a = 0; a < TM; ++a)
#pragma unroll
                        for (int b = 0; b < TN; ++b)
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                const f32x16& o = outa[e >> 1][e & 1][a][b];
                                const f32x4 v = {o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]};
                                __builtin_amdgcn_raw_buffer_store_b128(
                                    __builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v), rs_part, pbase,
                                    (((e * TM + a) * TN + b) * 4 + g) * 256 * 16, 16);
                            }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) __hip_atomic_store(p.flags + slot, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                continue;
            }
        }

        // ---- epilogue: four output positions, each through the LDS C tile
        f32x4* sC = smem;
        const int n4 = tid % CH, r0 = tid / CH;
        const int n = n0 + n4 * 4;
        const f32x4 sc = *reinterpret_cast<const f32x4*>(p.scale + n);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(p.shift + n);
        const int nvalid = p.cout - n;
        const bool vec = p.vec_io && nvalid >= 4;
#pragma unroll
        for (int pq = 0; pq < 4; ++pq) {
            const int py = pq >> 1, px = pq & 1;
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                const int ml = wm * WM + a * 32 + fi;
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int c4 = (wn * WN + b * 32) / 4 + 2 * g + fk;
                        const f32x16& o = outa[py][px][a][b];
                        f32x4 v = {o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]};
                        sC[ml * CH + (c4 ^ (ml & 7))] = v;
                    }
            }
            __syncthreads();
#pragma unroll 2
            for (int ps = 0; ps < BM / RP; ++ps) {
                const int ml = ps * RP + r0;
                const int m = m0 + ml;
                if (m >= p.T || nvalid <= 0) continue;
                const int bi = m / (p.TH * p.TW);
                const int rr = m - bi * p.TH * p.TW;
                const int ty = rr / p.TW, tx = rr - ty * p.TW;
                const int y = 2 * ty + py, x = 2 * tx + px;
                if (y >= p.H || x >= p.W) continue;
                const size_t pix = ((size_t)bi * p.H + y) * p.W + x;
                f32x4 v = sC[ml * CH + (n4 ^ (ml & 7))];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float tv = fmaf(v[k], sc[k], sh[k]);
                    v[k] = p.leaky ? (tv > 0.f ? tv : tv * 0.1f) : tv;
                }
                float* o = p.out + pix * p.out_pix_stride + n;
                if (p.res) {
                    const float* rp = p.res + pix * p.res_pix_stride + n;
                    if (vec) v += *reinterpret_cast<const f32x4*>(rp);
                    else
                        for (int k = 0; k < 4 && k < nvalid; ++k) v[k] += rp[k];
                }
                if (vec) *reinterpret_cast<f32x4*>(o) = v;
                else
                    for (int k = 0; k < 4 && k < nvalid; ++k) o[k] = v[k];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // LDS reads done; the stores keep flying
            __builtin_amdgcn_s_barrier();
        }
        if (!SK && !p.ticket) break;
    }
}


// ------------------------------------------------------------------------------------------------
// Fused variant: the input transform happens in the GEMM's A-operand loader, V never exists in memory.
//   V_xi[tile][c] = sum over the 2 x 2 patch pixels B^T row i / column j select, with +-1 signs
// Each thread owns two tile rows x one 16-byte channel chunk: 4 buffer loads (zero padded by the descriptor)
// -> 3 vector add/subs -> one ds_write_b128, software-pipelined TWO steps ahead through a 3-deep ring of A
// buffers so that no load is waited for less than ~4 slots after its issue; the weights keep the LDS-DMA path.
// Removes wino_input_kernel (9 % of the step, pure HBM traffic) at the price of 4x more L2 reads of X.
// ------------------------------------------------------------------------------------------------
template <int BN, int WN>
__global__ __launch_bounds__(256, 2) void wino_fused_kernel(const WinoParams p) {
    constexpr int BM = 64, WM = 32;
    constexpr int TN = WN / 32;
    constexpr int NWN = BN / WN;
    constexpr int B_CH = BN / 32;
    constexpr int CH = BN / 4, RP = 256 / CH;
    static_assert((BM / WM) * (BN / WN) == 4, "four waves per workgroup");
    constexpr int A_BUF = BM * 8, B_BUF = BN * 8;                 // f32x4 units
    static_assert(BM * BN / 4 <= 3 * A_BUF + 2 * B_BUF, "C tile must fit in the operand buffers");
    __shared__ f32x4 smem[3 * A_BUF + 2 * B_BUF + 1];
    int* const s_ticket = reinterpret_cast<int*>(smem + 3 * A_BUF + 2 * B_BUF);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave / NWN, wn = wave % NWN;
    const int lrow = tid >> 3, lcol = tid & 7;
    const int lsw = lcol ^ ((lrow >> 1) & 7);
    const int scol = lsw;                                         // DMA source chunk (weights)
    const int fi = lane & 31, fk = lane >> 5;
    const int fsw = (fi >> 1) & 7;
    const int ksteps = 16 * p.kc;
    const size_t u_plane = (size_t)p.cout_pad * p.C;

    for (;;) {
        int tile;
        if (p.ticket) {
            // raw barrier: only lane 0's wave pays the ticket's round trip, nobody drains the previous tile's stores
            if (tid == 0) *s_ticket = atomicAdd(p.ticket, 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            tile = *s_ticket;
        } else {
            tile = blockIdx.x;
        }
        if (tile >= p.total_tiles) break;
        tile = __builtin_amdgcn_readfirstlane(tile);
        const int tile_n = tile % p.n_tiles;
        const int tile_m = tile / p.n_tiles;
        const int m0 = tile_m * BM, n0 = tile_n * BN;

        // ---- per-thread patch geometry of its two tile rows
        const int tiles_per_img = p.TH * p.TW;
        const int b_first = (m0 < p.T ? m0 : p.T - 1) / tiles_per_img;
        int pixrel[2];
        unsigned vmask[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int m = m0 + lrow + 32 * j;
            const bool mok = m < p.T;
            if (!mok) m = p.T - 1;
            const int b = m / tiles_per_img;
            const int rr = m - b * tiles_per_img;
            const int ty = rr / p.TW, tx = rr - ty * p.TW;
            pixrel[j] = ((b - b_first) * p.H + 2 * ty - 1) * p.W + 2 * tx - 1;
            unsigned vm = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const bool ok = mok & ((unsigned)(2 * ty - 1 + r) < (unsigned)p.H) & ((unsigned)(2 * tx - 1 + c) < (unsigned)p.W);
                    vm |= (ok ? 1u : 0u) << (r * 4 + c);
                }
            vmask[j] = vm;
        }
        const float* in_base = p.X + (size_t)b_first * p.H * p.W * p.in_pix_stride;
        const size_t in_left = ((size_t)p.total_in_pixels - (size_t)b_first * p.H * p.W) * p.in_pix_stride * 4;
        const auto rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in_base), 0,
                                                             in_left < 0x7FFFFFFFull ? (int)in_left : 0x7FFFFFFF, 0x00020000);
        const int lcol16 = lcol * 16;
        const int stride4 = p.in_pix_stride * 4;

        // B^T row i (and column j) of the input transform: +-d[ra] +- d[rb]
        //   i : 0 -> d0 - d2, 1 -> d1 + d2, 2 -> d2 - d1, 3 -> d1 - d3
        auto sel_a = [](int i) { return i == 0 ? 0 : 1; };
        auto sel_b = [](int i) { return i == 3 ? 3 : 2; };
        auto sgn_a = [](int i) { return i == 2 ? -1.f : 1.f; };
        auto sgn_b = [](int i) { return (i == 0 || i == 3) ? -1.f : 1.f; };

        f32x4 L[4];                       // the four patch pixels of the row being fetched
        int a_xi = 0, a_cc = 0;           // (transform index, channel chunk) of the A step being fetched
        int n_xi = 0, n_cc = 0;           // ... of the B (weights) step being fetched
        auto adv = [&](int& xi, int& cc) {
            if (++cc == p.kc) { cc = 0; ++xi; }
        };
        auto load_row = [&](int j, int k, bool live) {      // k-th of the 4 patch pixels of tile row j
            const int i = a_xi >> 2, jj = a_xi & 3;
            const int r = (k & 2) ? sel_b(i) : sel_a(i);
            const int c = (k & 1) ? sel_b(jj) : sel_a(jj);
            const bool ok = live & (((vmask[j] >> (r * 4 + c)) & 1u) != 0);
            const int voff = ((pixrel[j] + r * p.W + c) * stride4 + lcol16) | (ok ? 0 : (int)0x80000000);
            L[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, voff, a_cc * 128, 0));
        };
        auto write_row = [&](int j, int abuf) {
            const int i = a_xi >> 2, jj = a_xi & 3;
            const float sa = sgn_a(i), sb = sgn_b(i), ca_ = sgn_a(jj), cb_ = sgn_b(jj);
            // (element by element: a vector times a scalar in a register becomes a packed instruction with a source-half selection,
            // see axpy16)
            f32x4 out;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float top = __builtin_fmaf(L[0][e], ca_, L[1][e] * cb_);
                const float bot = __builtin_fmaf(L[2][e], ca_, L[3][e] * cb_);
                out[e] = __builtin_fmaf(top, sa, bot * sb);
            }
            smem[abuf * A_BUF + (lrow + 32 * j) * 8 + lsw] = out;
        };
        auto issue_b = [&](int piece, int bbuf, bool live) {
            const float* bbase = p.U + (size_t)n_xi * u_plane + (size_t)n0 * p.C;
            const auto rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bbase), 0, live ? BN * p.C * 4 : 0, 0x00020000);
            f32x4* dst = smem + 3 * A_BUF + bbuf * B_BUF + wave_u * 64 + piece * 256;
            const int vo = ((lrow + 32 * piece) * p.C + scol * 4) * 4;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)dst, 16, vo, n_cc * 128, 0, 0);
        };

        f32x16 acc[TN], outa[2][2][TN];
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[b][r] = 0.f;
                outa[0][0][b][r] = 0.f; outa[0][1][b][r] = 0.f; outa[1][0][b][r] = 0.f; outa[1][1][b][r] = 0.f;
            }

        const f32x4* fragA = smem + (wm * WM + fi) * 8;
        const f32x4* fragB = smem + 3 * A_BUF + (wn * WN + fi) * 8;
        f32x4 ca, cb[TN], na, nb[TN];
        auto read_frags = [&](f32x4& fa, f32x4(&fb)[TN], int abuf, int bbuf, int q) {
            const int ch = (2 * q + fk) ^ fsw;
            fa = fragA[abuf * A_BUF + ch];
#pragma unroll
            for (int b = 0; b < TN; ++b) fb[b] = fragB[bbuf * B_BUF + b * 32 * 8 + ch];
        };

        // ---- prologue: A of step 0 (both rows) and step 1 (row 0), row 1 of step 1 left in flight; B of step 0
#pragma unroll
        for (int piece = 0; piece < B_CH; ++piece) issue_b(piece, 0, true);
        adv(n_xi, n_cc);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int k = 0; k < 4; ++k) load_row(j, k, true);
            write_row(j, 0);
        }
        adv(a_xi, a_cc);                                    // A fetch state = step 1
        {
            const bool live1 = 1 < ksteps;
#pragma unroll
            for (int k = 0; k < 4; ++k) load_row(0, k, live1);
            write_row(0, 1);
#pragma unroll
            for (int k = 0; k < 4; ++k) load_row(1, k, live1);   // consumed at slot 0 of step 0
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // the B DMA of step 0 is older than the 4 loads still in flight
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        read_frags(ca, cb, 0, 0, 0);

        int abuf = 0;          // A ring slot of the step being computed
        int xi = 0, cc = 0;
        for (int s = 0; s < ksteps; ++s) {
            const int bbuf = s & 1;
            const int abuf1 = abuf == 2 ? 0 : abuf + 1;     // step s+1
            const int abuf2 = abuf1 == 2 ? 0 : abuf1 + 1;   // step s+2
            const bool live1 = s + 1 < ksteps, live2 = s + 2 < ksteps;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int slot = q * 4 + t;
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(cb[b][t], ca[t], acc[b], 0, 0, 0);
                    if (slot == 0) {
                        write_row(1, abuf1);                // row 1 of step s+1 (loaded during the previous step)
                        adv(a_xi, a_cc);                    // A fetch state = step s+2
                    }
                    if (slot >= 1 && slot <= 4) load_row(0, slot - 1, live2);
                    if (slot < B_CH) issue_b(slot, bbuf ^ 1, live1);
                    if (t == 1 && q < 3) read_frags(na, nb, abuf, bbuf, q + 1);
                    if (slot == 7) write_row(0, abuf2);
                    if (slot >= 8 && slot <= 11) load_row(1, slot - 8, live2);
                    if (slot == 11) {
                        // B DMA of step s+1 and every ds_write so far are done; the 4 row-1 loads stay in flight
                        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                    }
                    if (slot == 12) read_frags(na, nb, abuf1, bbuf ^ 1, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                ca = na;
#pragma unroll
                for (int b = 0; b < TN; ++b) cb[b] = nb[b];
            }
            adv(n_xi, n_cc);
            abuf = abuf1;
            if (++cc == p.kc) {
                const int i = xi >> 2, j = xi & 3;
                const float ci0 = i < 3 ? 1.f : 0.f, ci1 = i == 0 ? 0.f : (i == 1 ? 1.f : -1.f);
                const float cj0 = j < 3 ? 1.f : 0.f, cj1 = j == 0 ? 0.f : (j == 1 ? 1.f : -1.f);
                const float c00 = ci0 * cj0, c01 = ci0 * cj1, c10 = ci1 * cj0, c11 = ci1 * cj1;
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    if (c00 != 0.f) axpy16(outa[0][0][b], acc[b], c00);
                    if (c01 != 0.f) axpy16(outa[0][1][b], acc[b], c01);
                    if (c10 != 0.f) axpy16(outa[1][0][b], acc[b], c10);
                    if (c11 != 0.f) axpy16(outa[1][1][b], acc[b], c11);
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
                }
                cc = 0;
                ++xi;
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();

        // ---- epilogue: four output positions, each through the LDS C tile
        f32x4* sC = smem;
        const int n4 = tid % CH, r0 = tid / CH;
        const int n = n0 + n4 * 4;
        const f32x4 sc = *reinterpret_cast<const f32x4*>(p.scale + n);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(p.shift + n);
        const int nvalid = p.cout - n;
        const bool vec = p.vec_io && nvalid >= 4;
#pragma unroll
        for (int pq = 0; pq < 4; ++pq) {
            const int py = pq >> 1, px = pq & 1;
            const int ml0 = wm * WM + fi;
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c4 = (wn * WN + b * 32) / 4 + 2 * g + fk;
                    const f32x16 o = outa[py][px][b];
                    f32x4 v = {o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]};
                    sC[ml0 * CH + (c4 ^ (ml0 & 7))] = v;
                }
            __syncthreads();
#pragma unroll 2
            for (int ps = 0; ps < BM / RP; ++ps) {
                const int ml = ps * RP + r0;
                const int m = m0 + ml;
                if (m >= p.T || nvalid <= 0) continue;
                const int bi = m / tiles_per_img;
                const int rr = m - bi * tiles_per_img;
                const int ty = rr / p.TW, tx = rr - ty * p.TW;
                const int y = 2 * ty + py, x = 2 * tx + px;
                if (y >= p.H || x >= p.W) continue;
                const size_t pix = ((size_t)bi * p.H + y) * p.W + x;
                f32x4 v = sC[ml * CH + (n4 ^ (ml & 7))];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float tv = fmaf(v[k], sc[k], sh[k]);
                    v[k] = p.leaky ? (tv > 0.f ? tv : tv * 0.1f) : tv;
                }
                float* o = p.out + pix * p.out_pix_stride + n;
                if (p.res) {
                    const float* rp = p.res + pix * p.res_pix_stride + n;
                    if (vec) v += *reinterpret_cast<const f32x4*>(rp);
                    else
                        for (int k = 0; k < 4 && k < nvalid; ++k) v[k] += rp[k];
                }
                if (vec) *reinterpret_cast<f32x4*>(o) = v;
                else
                    for (int k = 0; k < 4 && k < nvalid; ++k) o[k] = v[k];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // LDS reads done; the stores keep flying
            __builtin_amdgcn_s_barrier();
        }
        if (!p.ticket) break;
    }
}

size_t wino_scratch_floats(int B, int H, int W, int C) {
    const size_t T = (size_t)B * ((H + 1) / 2) * ((W + 1) / 2);
    return 16 * T * C;
}

// The fused loader wins where the transform kernel's HBM round trip is large next to the GEMM (cin <= 64:
// conv2.1, conv3.x); for wider layers the separate transform + pure-DMA GEMM is faster (measured: 136^2 128->256
// 1.88 vs 1.79 ms, 34^2 256->512 0.575 vs 0.522 ms).
static bool wino_fused(int cin) { return cin <= 64; }

bool wino_fused_for(int cin) { return wino_fused(cin); }

bool wino_enabled() { return true; }

template <int BM, int BN, int WM, int WN>
static int launch_wino_tile(WinoParams p, int blocks_per_cu, hipStream_t stream) {
    const int m_tiles = (p.T + BM - 1) / BM;
    p.n_tiles = p.cout_pad / BN;
    const long long total = (long long)m_tiles * p.n_tiles;
    OM_REQUIRE(total > 0 && total < (1ll << 31), OM_EINVAL, "winograd: %lld tiles out of range", total);
    p.total_tiles = (int)total;
    long long grid = total;
    if (p.ticket) grid = total < 256ll * blocks_per_cu ? total : 256ll * blocks_per_cu;
    // stream-K between one and two tiles per slot; with fewer tiles than resident workgroups (the 17 x 17 layers: 656 tiles for
    // 768 slots, i.e. two or three per CU) on 512 slots, so that every CU gets the same 2 x 1.28 tiles' worth
    long long sk_slots = grid;
    if (p.partial && p.ticket && total < grid && total >= 512) sk_slots = 512;
    if (p.partial && p.ticket && total >= sk_slots && total < 2 * sk_slots && sk_slots <= SK_SLOTS &&
        (size_t)sk_slots * 4 * BM * BN * sizeof(float) <= SK_PARTIAL_BYTES) {
        p.slots = (int)sk_slots;
        hipLaunchKernelGGL((wino_gemm_kernel<BM, BN, WM, WN, true>), dim3((unsigned)sk_slots), dim3(256), 0, stream, p);
    } else {
        hipLaunchKernelGGL((wino_gemm_kernel<BM, BN, WM, WN, false>), dim3((unsigned)grid), dim3(256), 0, stream, p);
    }
    OM_CHECK_HIP(hipGetLastError());
    return OM_OK;
}

template <int BN, int WN>
static int launch_wino_fused(WinoParams p, int blocks_per_cu, hipStream_t stream) {
    const int m_tiles = (p.T + 63) / 64;
    p.n_tiles = p.cout_pad / BN;
    const long long total = (long long)m_tiles * p.n_tiles;
    OM_REQUIRE(total > 0 && total < (1ll << 31), OM_EINVAL, "winograd: %lld tiles out of range", total);
    p.total_tiles = (int)total;
    long long grid = total;
    if (p.ticket) grid = total < 256ll * blocks_per_cu ? total : 256ll * blocks_per_cu;
    hipLaunchKernelGGL((wino_fused_kernel<BN, WN>), dim3((unsigned)grid), dim3(256), 0, stream, p);
    OM_CHECK_HIP(hipGetLastError());
    return OM_OK;
}

int wino_bn(long long T, int cout_pad) {
    if (cout_pad % 128) return 64;
    return ((T + 63) / 64) * (cout_pad / 128) < 3000 ? 64 : 128;
}

// a.w must point at the transformed weights U [16][cout_pad][cin]; scratch holds V (wino_scratch_floats).
int launch_conv_winograd(const ConvArgs& a, float* scratch, hipStream_t stream) {
    OM_REQUIRE(a.in && a.w && a.scale && a.shift && a.out && scratch, OM_EINVAL, "winograd: null pointer");
    OM_REQUIRE(a.ks == 3 && a.stride == 1 && a.out_mode == 0, OM_EINVAL, "winograd: 3x3 stride-1 NHWC layers only");
    OM_REQUIRE(a.cin % 32 == 0 && a.cin >= 32 && a.cout_pad % 64 == 0, OM_EINVAL, "winograd: cin=%d cout_pad=%d", a.cin,
               a.cout_pad);
    OM_REQUIRE(a.in_pix_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(a.in) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(a.w) & 15) == 0 && (reinterpret_cast<uintptr_t>(scratch) & 15) == 0,
               OM_EINVAL, "winograd: operands must be 16-byte aligned");
    const int TH = (a.H + 1) / 2, TW = (a.W + 1) / 2;
    const long long T = (long long)a.B * TH * TW;
    OM_REQUIRE(T < (1ll << 31) && 16 * T * a.cin < (1ll << 40), OM_EINVAL, "winograd: problem too large");
    const bool fused = wino_fused(a.cin);
    const long long threads = T * (a.cin / 4);
    if (!fused)
        hipLaunchKernelGGL(wino_input_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, a.in, scratch, a.H,
                       a.W, a.cin, a.in_pix_stride, TH, TW, (int)T);
    OM_CHECK_HIP(hipGetLastError());
    if (a.mid_event) OM_CHECK_HIP(hipEventRecord(a.mid_event, stream));
    WinoParams p;
    p.X = a.in; p.in_pix_stride = a.in_pix_stride; p.total_in_pixels = a.B * a.H * a.W;
    p.V = scratch; p.U = a.w; p.scale = a.scale; p.shift = a.shift; p.res = a.res; p.out = a.out; p.ticket = a.ticket;
    p.T = (int)T; p.TH = TH; p.TW = TW; p.C = a.cin; p.kc = a.cin / 32;
    p.H = a.H; p.W = a.W; p.cout = a.cout; p.cout_pad = a.cout_pad;
    p.n_tiles = 0; p.total_tiles = 0;
    p.partial = a.sk_partial; p.flags = a.ticket ? a.ticket + SK_FLAG_OFF : nullptr; p.slots = 0;
    p.leaky = a.leaky; p.res_pix_stride = a.res_pix_stride; p.out_pix_stride = a.out_pix_stride;
    p.vec_io = (a.out_pix_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0 &&
                (!a.res || (a.res_pix_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(a.res) & 15) == 0)))
                   ? 1 : 0;
    if (fused) {
        if (a.cout_pad % 128 == 0) return launch_wino_fused<128, 64>(p, 2, stream);
        return launch_wino_fused<64, 32>(p, 3, stream);
    }
    // 64x128 tiles (2 workgroups/CU) win only when there are many of them; with fewer than ~3000 the 64x64 tiles
    // (3 workgroups/CU, finer load balance) are 5-11 % faster (measured at 68^2, 34^2 and 17^2).
    const bool bn64 = wino_bn(p.T, a.cout_pad) == 64;
    if (a.cout_pad % 128 == 0 && !bn64) return launch_wino_tile<64, 128, 32, 64>(p, 2, stream);
    return launch_wino_tile<64, 64, 32, 32>(p, 3, stream);
}

}  // namespace om
