// Fused stride-1 3x3 convolution with SPLIT fp32 operands (precision mode 1): Winograd F(4,3) along the image rows, the
// three kernel rows as part of the contraction, and the input transform computed INSIDE the workgroup -- the transformed
// input never exists in global memory.
//
//   Y[oy][4 t + px] = sum_j A^T[px][j] * M_j[oy][t],      M_j[oy][t][n] = sum_ky sum_c U[ky][j][n][c] * V_j[oy + ky - 1][t][c]
//   V_j[y][t][c]    = sum_x B^T[j][x] d[y][4 t - 1 + x][c]                (6 transform points per 4 outputs, x = 0..5)
//   U[ky][j][n][c]  = sum_kx G[j][kx] w[n][c][ky][kx]                     (host, float64; orienmask_amd/pack.py)
//
// Same layer as conv_wino24.hip (Conv2d 3x3 s1 p1 -> BatchNorm2d(eval) -> LeakyReLU(0.1) (+ residual),
// /root/reference/model/base.py:104-137, model/backbone/darknet.py:14-15) and the same F(4,3) matrices as its column transform.
//
// Why this form (MI355X, measured in round 3: profiles/r03_experiments.md): with the two-kernel F(2x4) form the
// transformed input (3x the activation) is written and read through HBM -- 53 of the forward's 83 GB per step.  Keeping V on chip needs the accumulators of ALL planes of
// a tile resident while the channel chunks stream by (the transform of a chunk yields every plane at once): 24 planes x 16
// registers for F(2x4) leaves room for one 32x32 tile per SIMD; F(4,3) along the rows only has SIX planes -- 96 registers per
// 32x32 wave tile, eight waves per workgroup, a 128 x 64 tile per CU -- at 4.5 instead of 3 matrix products per output, which the
// idle matrix pipe has room for.  Per 16-channel chunk a workgroup
//   * (four producer waves) loads the (R + 2) x (4 Ct + 2) input pixels of its R x Ct block of 1x4 output tiles straight into
//     registers (one chunk ahead), applies B^T, splits into hi/lo fp16 and writes V_j[row][t] as 64-byte LDS entries
//     [8 hi | 8 hi | 8 lo | 8 lo] -- 2 1/2 planes per group of matrix work (see the schedule at `chunk` below);
//   * (eight consumer waves) runs 18 (ky, j) steps of three v_mfma_f32_32x32x16_f16 per wave: the A fragment of tap ky is the SAME LDS plane read Ct
//     entries further on (row oy + ky of the block: conv3x3_f16.hip's shared patch, one dimension up), so V is stored once for
//     the three kernel rows; zero padding is zero ENTRIES (rows / columns outside the image are written as zeros), no masks;
//   * streams U through a ring of W14_RING (four since round 4; three before) LDS-DMA slots of (j; ky = 0..2) groups, 12 KiB each, contiguous in the packed blob; a
//     group's first weight fragments are read AFTER the barrier that starts the group, so a group requested in group g is not
//     needed before group g + 2 (two groups to land: the landing time of this stream is the consumers' critical path).
// What binds the kernel is the CU's vector-memory request path (L1 pending-request stalls 41 % of the time: 72 KB of weights and
// 61 KB of input per chunk of a 128 x 64 tile against ~20 B/clk of ingest), not HBM, the matrix pipe (27 % busy) or the producers.
// Tiles are blocks of the PADDED row space G = b (H + 2) + y + 1 (one zero row above and below every image), so a block may
// span images (17 x 17 layers) and the row shift ky needs no per-lane case.  The inverse transform runs once per tile, in the
// epilogue, position by position from the six plane accumulators.
//
// Numerics: V_j grows the activation by at most 10x (|4| + |5| + 1), so the split representation's range is |activation| <
// 6550 here (3275 for F(2x4)); one-dimensional F(4,3) is better conditioned than the two-dimensional F(2x4) it replaces.
// Every output is the same sequence of fp32 operations whatever tile it falls in (chunks outer, ky, then the three products),
// so results do not depend on the batch size.
#include "wino14_shared.h"

namespace om {

#ifndef W14_RING
#define W14_RING 4             // weight-ring slots.  4: blocks of (R + 2) * Ct <= 144 entries (two V buffers of 110 592 B leave room for a
#endif                         // fourth 12 KiB slot), a group's first fragments are read BEFORE the barrier that starts it (no LDS round
                               // trip between the barrier and the group's first matrix instruction) and its weights still have two
                               // groups to land (requested three groups ahead)
constexpr int W14_EMAX = W14_RING == 4 ? 144 : 160;      // LDS entries per plane: (R + 2) * Ct <= W14_EMAX
constexpr int W14_VPLANE = W14_EMAX * 4;      // f32x4 units (16 B) per plane
constexpr int W14_VBUF = 6 * W14_VPLANE;      // one transformed chunk: 61440 B
constexpr int W14_UGRP = 3 * W14_BN * 4;      // one (j; ky = 0..2) weight group: 12288 B
#ifndef W14_QUEUE
#define W14_QUEUE 0            // per-XCD tile queues: 0 = M blocks partitioned (N-tile siblings share the input behind one L2),
#endif                         // 1 = N tiles partitioned (an XCD streams one or two N tiles' weights: they stay in its L2)
#ifndef W14_B_ACROSS
#define W14_B_ACROSS (W14_RING == 4)   // 1: the first weight fragments of a group are read before the barrier that starts it (its weights
#endif                         // landed a group earlier); 0: after it -- the requests get two groups to land (three-slot ring)
#ifndef W14_DMA_AFTER
#define W14_DMA_AFTER 0        // the weight request follows the matrix instructions of this kernel row of the group
#endif
#ifndef W14_DMA_WAVES
#define W14_DMA_WAVES 0        // who requests a weight group's twelve 1-KiB pieces: 0 = every consumer wave one, waves 0-3 a second one;
#endif                         // 1 = waves 0-3 three each (they reach the group barrier ~280 cycles before waves 4-7: a request that
                               // stalls on a full vector-memory queue costs them slack instead of matrix-instruction issue)
constexpr int W14_THREADS = 768;              // waves 0-7: consumers (LDS reads + matrix instructions), 8-11: producers


// Roles.  The matrix waves must never wait on global memory: with the input loads, the transform and the weight DMA in their
// own instruction streams (round 3's first version) a 16-channel chunk cost a third more than its matrix instructions -- VMEM
// issue stalls of 60-180 cycles per request and 250 vector instructions per chunk in front of in-order matrix instructions.
// So waves 8-11 are PRODUCERS -- they request the next chunk's input pixels, apply B^T, split and write V -- and waves 0-7 are
// CONSUMERS: the weight ring's LDS-DMA requests, LDS fragment reads, matrix instructions, and the epilogue from registers.
// The two roles are two separate loops over the same sequence of tiles, chunks and groups that meet at one s_barrier per group
// (the barrier counts waves, not program counters), so the six plane accumulators are live only in the consumers' code and the
// input registers only in the producers': 168 registers, three waves per SIMD, one workgroup per CU.
//
// Tiles overlap at their ends (profiles/r03_w14_trace_tile_phases.txt: prologue + epilogue were 21 000 of a cin = 128 tile's
// 80 000 cycles): the ticket of tile i + 1 is taken during the prologue of tile i (two LDS words, published by the prologue
// barrier), so no barrier separates two tiles; the producers request the next tile's first chunk during the LAST chunk of this
// one (their input registers are free then) and transform it while the consumers -- who first request the next tile's first
// two weight groups -- run their epilogue.
// MODE: the epilogue's form -- 0 buffer-descriptor stores, no residual; 1 the same with a residual; 2 any view, any cout
template <int MODE>
__global__ __launch_bounds__(W14_THREADS, 3) void wino14_split_kernel(const Wino14Params p) {
    __shared__ f32x4 smem[2 * W14_VBUF + W14_RING * W14_UGRP + 1];      // ONE LDS object (conv_igemm.hip); last 16 B: two ticket words
    int* const s_ticket = reinterpret_cast<int*>(smem + 2 * W14_VBUF + W14_RING * W14_UGRP);
    f32x4* const s_u = smem + 2 * W14_VBUF;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ngroups = 6 * p.nch;

    // EIGHT tile queues, one per XCD (ticket words 0..7), as in conv_wino24.hip: XCD x owns the row/column blocks
    // [m_tiles x / 8, m_tiles (x + 1) / 8) and its workgroups draw (block, N tile) pairs N fastest, so the n_tiles workgroups
    // that transform the same input block run at the same time behind the SAME L2 -- the first one's requests go to HBM, the
    // others' hit (the producers cannot hide an HBM round trip: their registers hold barely one chunk in flight).  A workgroup
    // whose queue is empty moves on to the next XCD's (never back), so the last round still balances over the whole chip.
    // Placement only: results do not depend on which workgroup computes a tile.  Drawn by ONE thread (the producers' first),
    // which keeps the queue position.
    int q_xcd = 0, q_hops = 0;
    auto draw_tile = [&]() {
        const int m_tiles = p.total_tiles / p.n_tiles;
#if W14_QUEUE == 1
        // queue q: N tiles k, k + classes, ... (k = q % classes) of the M blocks' part q / classes
        const int classes = p.n_tiles >= 8 ? 8 : (p.n_tiles == 4 || p.n_tiles == 2 || p.n_tiles == 1) ? p.n_tiles : 1;
        const int mparts = 8 / classes;
#endif
        while (q_hops < 8) {
            const int q = (q_xcd + q_hops) & 7;
#if W14_QUEUE == 1
            const int k = q % classes, mp = q / classes;
            const int pm0 = (int)((long long)m_tiles * mp / mparts), pm1 = (int)((long long)m_tiles * (mp + 1) / mparts);
            const int nk = (p.n_tiles - k + classes - 1) / classes;
            const int v = atomicAdd(p.ticket + q, 1);
            if (v < (pm1 - pm0) * nk) return (pm0 + v / nk) * p.n_tiles + k + classes * (v % nk);
#else
            const int pm0 = (int)((long long)m_tiles * q >> 3), pm1 = (int)((long long)m_tiles * (q + 1) >> 3);
            const int v = atomicAdd(p.ticket + q, 1);
            if (v < (pm1 - pm0) * p.n_tiles) return pm0 * p.n_tiles + v;
#endif
            ++q_hops;
        }
        return p.total_tiles;
    };
    if (tid == 512) {
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(q_xcd));
        q_xcd &= 7;
        if (OM_W14_ABLATE & 8192) q_xcd = 0;        // measurement: one queue order for the whole chip
        s_ticket[0] = draw_tile();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int tile = __builtin_amdgcn_readfirstlane(s_ticket[0]);
    int tslot = 0;                      // s_ticket[tslot] is this tile's ticket, s_ticket[tslot ^ 1] takes the next one
#if OM_W14_TRACE
    bool first_tile = true;
    int n_traced = 0;
#endif

    if (wave >= 8) {
        // ================================================================ producers
        const int pid = tid - 512;
        const int hp2 = p.H + 2;
        const int ecount = (p.R + 2) * p.Ct;
        // a producer's few instructions per group are on everybody's critical path (the group barrier): issue them first
        if (!(OM_W14_ABLATE & 4096)) __builtin_amdgcn_s_setprio(3);
        const auto rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);

        // Items.  Entry e = rr * Ct + t is padded row g0 - 1 + rr, tile column t0 + t: six pixels x = 4 t - 1 .. 4 t + 4 of 16
        // channels per chunk.  A thread owns channel QUAD pid & 3 of entries pid >> 2 (item 0) and 64 + (pid >> 2) (item 1), and a
        // channel PAIR of entry 128 + (pid >> 3) (item 2: quad (pid >> 1) & 3, pair pid & 1) -- the 32 entries beyond 128 are half
        // an item's work per thread as pairs, a whole item's as quads (every instruction of a partly filled round costs the same).
        // Per item: the byte offset of its first pixel, a 6-bit "pixel exists" mask, the LDS byte address of its hi halfs in
        // plane 0 (lo: the 16-byte chunk two further on; -1: no such entry).
        int xbase[3], xlds[3];
        unsigned xok[3];
        auto setup_items = [&](int tile_id) {
            Wino14Tile tl;
            wino14_decode(p, tile_id, tl);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int e = k < 2 ? 64 * k + (pid >> 2) : 128 + (pid >> 3);
                const int q = k < 2 ? pid & 3 : (pid >> 1) & 3;
                const int ch = k < 2 ? 4 * q : 4 * q + 2 * (pid & 1);
                const int rr = e / p.Ct, t = e - rr * p.Ct;
                const int g = tl.g0 - 1 + rr;
                const int b = g / hp2;
                const int y = g - b * hp2 - 1;
                const bool rowok = e < ecount && g >= 0 && g < p.gtot && y >= 0 && y < p.H;
                const int x0 = 4 * (tl.t0 + t) - 1;
                unsigned ok = 0;
#pragma unroll
                for (int x = 0; x < 6; ++x) ok |= (rowok && (unsigned)(x0 + x) < (unsigned)p.W ? 1u : 0u) << x;
                xok[k] = ok;
                xbase[k] = (((b * p.H + y) * p.W + x0) * p.in_ps + ch) * 4;
                // measurement (262144): the addresses a channel-chunk-major activation [B][cin / 16][H][W][16] would be read at
                if (OM_W14_ABLATE & 262144) xbase[k] = (((b * p.nch * p.H + y) * p.W + x0) * 16 + ch) * 4;
                const int sw = (e >> 2) & 3;
                xlds[k] = e < ecount ? e * 64 + (((q >> 1) ^ sw) * 16) + (ch & 7) * 2 : -1;
            }
        };
        f32x4 xq[2][6];         // the two quad items' pixels
        f32x2 xp[2][6];         // the pair item's, of two chunks in turn (it is in use in every group: see the schedule below)
        auto item_offset = [&](int k, int x, int c) {
            // a pixel outside the image (or a pad row) gets an offset beyond the descriptor's range: the load returns zeros
            int off = ((xok[k] >> x) & 1u) ? xbase[k] + x * p.in_ps * 4 + c * 64 : (int)0x80000000;
            if (OM_W14_ABLATE & 262144) off = ((xok[k] >> x) & 1u) ? xbase[k] + x * 64 + c * (p.H * p.W * 64) : (int)0x80000000;
            if (OM_W14_ABLATE & 512) off = lane * 16 + (k * 6 + x) * 1024;      // measurement: every request hits the same 18 KiB
            if ((OM_W14_ABLATE & 16384) && (c & 1)) off = (int)0x80000000;     // measurement: half the input requests (odd chunks none)
            if ((OM_W14_ABLATE & 32768) && (x == 0 || x == 5)) off = (int)0x80000000;   // measurement: no halo pixels (4 of 6 requests)
            return off;
        };
        // pixels [x0, x1) of an item
        auto load_quad_px = [&](int k, int c, int x0, int x1) {
            if (OM_W14_ABLATE & 16) return;
#pragma unroll
            for (int x = 0; x < 6; ++x)
                if (x >= x0 && x < x1) xq[k][x] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, item_offset(k, x, c), 0, 0));
        };
        auto load_quad = [&](int k, int c) { load_quad_px(k, c, 0, 6); };
        auto load_pair_px = [&](auto buf, int c, int x0, int x1) {
            if (OM_W14_ABLATE & 16) return;
#pragma unroll
            for (int x = 0; x < 6; ++x)
                if (x >= x0 && x < x1)
                    xp[decltype(buf)::value][x] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_in, item_offset(2, x, c), 0, 0));
        };
        auto load_pair = [&](auto buf, int c) { load_pair_px(buf, c, 0, 6); };
        // hi/lo split of one transformed value per channel and its LDS stores (hi: RNE fp16 of x; lo: RNE fp16 of x - hi, the
        // difference exact in one v_fma_mix_f32 per element -- the fp16 operand is widened by the instruction)
        auto split_store4 = [&](const f32x4& v, char* dst, int lo_off) {
            const f16x4 h = __builtin_convertvector(v, f16x4);
            const u32x2 hb = __builtin_bit_cast(u32x2, h);
            f32x4 rem;
            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(rem[0]) : "v"(hb[0]), "v"(v[0]));
            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rem[1]) : "v"(hb[0]), "v"(v[1]));
            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(rem[2]) : "v"(hb[1]), "v"(v[2]));
            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rem[3]) : "v"(hb[1]), "v"(v[3]));
            const f16x4 l = __builtin_convertvector(rem, f16x4);
            if ((OM_W14_ABLATE & 256) && h[0] != (_Float16)123.f) return;
            *reinterpret_cast<u32x2*>(dst) = hb;
            *reinterpret_cast<u32x2*>(dst + lo_off) = __builtin_bit_cast(u32x2, l);
        };
        auto split_store2 = [&](const f32x2& v, char* dst, int lo_off) {
            const f16x2 h = __builtin_convertvector(v, f16x2);
            const unsigned hb = __builtin_bit_cast(unsigned, h);
            f32x2 rem;
            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(rem[0]) : "v"(hb), "v"(v[0]));
            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rem[1]) : "v"(hb), "v"(v[1]));
            const f16x2 l = __builtin_convertvector(rem, f16x2);
            if ((OM_W14_ABLATE & 256) && h[0] != (_Float16)123.f) return;
            *reinterpret_cast<unsigned*>(dst) = hb;
            *reinterpret_cast<unsigned*>(dst + lo_off) = __builtin_bit_cast(unsigned, l);
        };
        // B^T along the row (the column transform of conv_wino24.hip), one transform point (plane) at a time:
        //   v0 = 4 d0 - 5 d2 + d4   v1 = (d3 + d4) - 4 (d1 + d2)   v2 = (d4 - d3) + 4 (d1 - d2)
        //   v5 = 4 d1 - 5 d3 + d5   v3 = (d4 - d2) + 2 (d3 - d1)   v4 = (d4 - d2) - 2 (d3 - d1)
        auto point = [&](const auto* d, int j) {
            using T = std::remove_cv_t<std::remove_reference_t<decltype(d[0])>>;
            const T c4 = 4.f, cm4 = -4.f, c2 = 2.f, cm2 = -2.f, cm5 = -5.f;
            switch (j) {
                case 0: return __builtin_elementwise_fma(d[2], cm5, __builtin_elementwise_fma(d[0], c4, d[4]));
                case 1: return __builtin_elementwise_fma(d[1] + d[2], cm4, d[3] + d[4]);
                case 2: return __builtin_elementwise_fma(d[1] - d[2], c4, d[4] - d[3]);
                case 3: return __builtin_elementwise_fma(d[3] - d[1], c2, d[4] - d[2]);
                case 4: return __builtin_elementwise_fma(d[3] - d[1], cm2, d[4] - d[2]);
                default: return __builtin_elementwise_fma(d[3], cm5, __builtin_elementwise_fma(d[1], c4, d[5]));
            }
        };
        // two planes (positions i and i + 1 of the consumers' plane order) of quad item k into V buffer vb
        auto quad_planes = [&](int k, int vb, int i) {
            if ((OM_W14_ABLATE & (16 | 64)) || xlds[k] < 0) return;
            char* base = reinterpret_cast<char*>(smem + vb * W14_VBUF) + xlds[k];
            // lo lives two 16-byte chunks after hi (chunk index XOR-swizzled: + 2 flips bit 1 of the chunk)
            const int lo_off = ((((xlds[k] >> 4) & 3) ^ 2) - ((xlds[k] >> 4) & 3)) * 16;
            const int ja = w14_plane(i), jb = w14_plane(i + 1);
            split_store4(point(xq[k], ja), base + ja * (W14_VPLANE * 16), lo_off);
            split_store4(point(xq[k], jb), base + jb * (W14_VPLANE * 16), lo_off);
        };
        auto pair_plane = [&](auto buf, int vb, int i) {
            if ((OM_W14_ABLATE & (16 | 64)) || xlds[2] < 0) return;
            char* base = reinterpret_cast<char*>(smem + vb * W14_VBUF) + xlds[2];
            const int lo_off = ((((xlds[2] >> 4) & 3) ^ 2) - ((xlds[2] >> 4) & 3)) * 16;
            const int j = w14_plane(i);
            split_store2(point(xp[decltype(buf)::value], j), base + j * (W14_VPLANE * 16), lo_off);
        };

        // the first tile's first chunk; every later tile's is requested during the tile before it
        if (tile < p.total_tiles) {
            setup_items(tile);
            load_quad(0, 0);
            load_quad(1, 0);
            load_pair(std::integral_constant<int, 0>{}, 0);
        }
        while (tile < p.total_tiles) {
#if OM_W14_TRACE
            unsigned long long pp0, pp1, pp2 = 0, pp3 = 0;
            W14_STAMP(pp0);
#endif
            // prologue: the next ticket requested, chunk 0 transformed, chunk 1 requested
            int next_ticket = 0;
            if (pid == 0) next_ticket = draw_tile();
#pragma unroll
            for (int i = 0; i < 6; i += 2) { quad_planes(0, 0, i); quad_planes(1, 0, i); }
#pragma unroll
            for (int i = 0; i < 6; ++i) pair_plane(std::integral_constant<int, 0>{}, 0, i);
            if (pid == 0) s_ticket[tslot ^ 1] = next_ticket;
            if (p.nch > 1) {
                load_quad(0, 1);
                load_quad(1, 1);
                load_pair(std::integral_constant<int, 1>{}, 1);
            }
#if OM_W14_TRACE
            W14_STAMP(pp1);
#endif
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const int next_tile = __builtin_amdgcn_readfirstlane(s_ticket[tslot ^ 1]);
            // One chunk of matrix work = six groups, one plane each, in the order w14_plane(0..5).  The producers' share of group
            // g while chunk c is being multiplied -- every group the same work, 2 1/2 planes of a quad item:
            //   groups 0-2   planes (2 g, 2 g + 1) of quad item 0 of chunk c + 1;   groups 3-5   the same of quad item 1;
            //   every group  plane g of the pair item of chunk c + 1
            // into the other V buffer.  Position i of chunk c + 1 is read by the consumers from group i - 1 on (position 0: from
            // group 5 of chunk c): item 1's planes (0, 1) are complete at the barrier that ends group 3.  A quad item's registers
            // take the request for chunk c + 2 after its last planes (groups 2 and 5: four groups of matrix work before their
            // first use; the input comes from HBM for the first of the N-tile siblings, ~3 us under load); the pair item, in
            // use in every group, alternates between two register sets (chunk parity `par`), the free one requested in group 0.
            // Three straight-line forms, without a branch around any request -- the compiler's wait for "this item has landed"
            // counts the requests issued after it on the worst path, and with `if (c + 2 < nch)` around each request it waited
            // for all but the three youngest (two groups' worth of HBM latency in every group):
            //   0  (chunks 0 .. nch - 3)  transform chunk c + 1, request chunk c + 2;     1  (chunk nch - 2)  the transform only;
            //   2  (chunk nch - 1)        nothing left to transform, every request of this tile has landed: the registers take
            //                             the first chunk of the NEXT tile.
            auto chunk = [&](int c, auto form, auto par) {
                constexpr int FORM = decltype(form)::value;
                constexpr int PAR = decltype(par)::value;           // c & 1: chunk c + 1's pair item is in xp[PAR ^ 1]
#pragma unroll
                for (int g = 0; g < 6; ++g) {
#if OM_W14_TRACE
                    unsigned long long ta, tb = 0, tc, td = 0;
                    W14_STAMP(ta);
#endif
                    if constexpr (FORM < 2) {
                        if constexpr (FORM == 0 && !W14_SPREAD) {
                            if (g == 0) load_pair(std::integral_constant<int, PAR>{}, c + 2);
                        }
                        quad_planes(g / 3, PAR ^ 1, 2 * (g % 3));
                        pair_plane(std::integral_constant<int, PAR ^ 1>{}, PAR ^ 1, g);
                        if constexpr (FORM == 0 && !W14_SPREAD) {
                            if (g == 2) load_quad(0, c + 2);
                            if (g == 5) load_quad(1, c + 2);
                        }
                        if constexpr (FORM == 0 && W14_SPREAD) {
                            // planes (0, 5) come first: only they read pixels 0 and 5, whose registers are free after the
                            // item's first group (2 requests); pixels 1..4 after its last (4); the pair item's free set in
                            // two threes: 2, 3, 4, 2, 3, 4 requests per group
                            if (g == 0 || g == 3) { load_quad_px(g / 3, c + 2, 0, 1); load_quad_px(g / 3, c + 2, 5, 6); }
                            if (g == 2 || g == 5) load_quad_px(g / 3, c + 2, 1, 5);
                            if (g == 1) load_pair_px(std::integral_constant<int, PAR>{}, c + 2, 0, 3);
                            if (g == 4) load_pair_px(std::integral_constant<int, PAR>{}, c + 2, 3, 6);
                        }
                    } else {
                        if (next_tile < p.total_tiles) {
                            if (g == 0) { setup_items(next_tile); load_quad(0, 0); }
                            if (g == 2) load_quad(1, 0);
                            if (g == 4) load_pair(std::integral_constant<int, 0>{}, 0);
                        }
                    }
#if OM_W14_TRACE
                    W14_STAMP(tc);
#endif
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // my LDS writes are done
                    if (!(OM_W14_ABLATE & 4)) __builtin_amdgcn_s_barrier();
#if OM_W14_TRACE
                    W14_SETTLE2(ta, tc);
                    if (first_tile) w14_trace_put(p, wave, c * 6 + g, ta, tb, tc, td);
#endif
                }
            };
            using I0 = std::integral_constant<int, 0>;
            using I1 = std::integral_constant<int, 1>;
            using I2 = std::integral_constant<int, 2>;
            int c = 0;
            for (; c + 3 < p.nch; c += 2) { chunk(c, I0{}, I0{}); chunk(c + 1, I0{}, I1{}); }
            if (c + 2 < p.nch) {            // c even, nch odd
                chunk(c, I0{}, I0{});
                chunk(c + 1, I1{}, I1{});
                chunk(c + 2, I2{}, I0{});
            } else if (c + 1 < p.nch) {     // c even, nch even
                chunk(c, I1{}, I0{});
                chunk(c + 1, I2{}, I1{});
            } else {                        // nch == 1
                chunk(c, I2{}, I0{});
            }
#if OM_W14_TRACE
            W14_SETTLE(pp0, pp1, pp2, pp3);
            if (wave == 8 && !first_tile && pp3 == 0) w14_trace_put(p, 8, 62, pp0, pp1, pp2, pp3);      // a steady-state prologue
            first_tile = false;
#endif
            tile = next_tile;
            tslot ^= 1;
        }
        return;
    }

    // ==================================================================== consumers
    const int wm = wave >> 1, wn = wave & 1;
    const int fi = lane & 31, fk = lane >> 5;
    // weight-group DMA: a group's 192 rows of 64 bytes are twelve 1-KiB pieces of 16 rows; wave w requests piece w, waves 0-3
    // also piece 8 + w
    const int drow = lane >> 2, dcol = lane & 3;
    constexpr int W14_NPIECE = W14_DMA_WAVES ? 3 : 2, W14_PSTEP = W14_DMA_WAVES ? 4 : 8;
    int dvo[W14_NPIECE];
#pragma unroll
    for (int i = 0; i < W14_NPIECE; ++i) {
        const int row = 16 * (wave + W14_PSTEP * i) + drow;
        dvo[i] = row * 64 + ((dcol ^ ((row >> 2) & 3)) * 16);      // swizzle on the SOURCE chunk: the LDS image stays lane-linear
    }
    const auto rs_u = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.u), 0, p.u_bytes, 0x00020000);
    // B fragments: row 32 wn + fi of a tap's 64 rows
    const int swB = (fi >> 2) & 3;
    const int boff_hi = (32 * wn + fi) * 4 + (fk ^ swB);
    const int boff_lo = (32 * wn + fi) * 4 + ((2 + fk) ^ swB);
    // A entries of this lane for the three kernel rows: entry m + ky Ct of the plane
    int aoff_hi[3], aoff_lo[3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int e = 32 * wm + fi + ky * p.Ct;
        const int sw = (e >> 2) & 3;
        aoff_hi[ky] = e * 4 + (fk ^ sw);
        aoff_lo[ky] = e * 4 + ((2 + fk) ^ sw);
    }
    // weight group g = 6 c + j of N tile tn: 12 KiB at ((tn * nch + c) * 6 + j) * 12288 bytes of the packed blob
    auto issue_group = [&](int ubase, int g) {
        if (!(OM_W14_ABLATE & 8) && g < ngroups) {
            const int slot = g % W14_RING;
            const int soff = ubase + (g - g % 6 + w14_plane(g % 6)) * (W14_UGRP * 16);
            // measurement (65536): every second weight group is not requested (the DMA writes zeros for out-of-range offsets)
            const int oob = ((OM_W14_ABLATE & 65536) && (g & 1)) ? (int)0x80000000 : 0;
            if (W14_DMA_WAVES) {
                if (wave < 4) {
#pragma unroll
                    for (int i = 0; i < W14_NPIECE; ++i)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_u, (lds_ptr_t)(s_u + slot * W14_UGRP + (wave + 4 * i) * 64), 16, dvo[i] | oob, soff, 0, 0);
                }
            } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_u, (lds_ptr_t)(s_u + slot * W14_UGRP + wave * 64), 16, dvo[0] | oob, soff, 0, 0);
            if (wave < 4)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_u, (lds_ptr_t)(s_u + slot * W14_UGRP + (wave + 8) * 64), 16, dvo[1] | oob, soff, 0, 0);
            }
        }
    };
    Wino14Tile tl;
    bool first_of_wg = true;
    if (tile < p.total_tiles) {
        wino14_decode(p, tile, tl);
        const int ubase = tl.tile_n * p.nch * 6 * (W14_UGRP * 16);
        issue_group(ubase, 0);
        issue_group(ubase, 1);
        if (W14_RING == 4) issue_group(ubase, 2);
    }
    while (tile < p.total_tiles) {
        const int ubase = tl.tile_n * p.nch * 6 * (W14_UGRP * 16);
        f32x16 acc[6];
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        // weight groups 0 and 1 have landed: they were requested before the previous tile's epilogue stores, which need not have
        // (the counter retires in order; the very first tile has nothing behind its requests)
        if ((OM_W14_ABLATE & 32) || W14_EPI_OPS<MODE> == 0 || first_of_wg) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W14_EPI_OPS<MODE>) : "memory");
        first_of_wg = false;
        __builtin_amdgcn_s_barrier();           // prologue: chunk 0 (producers), weight groups 0, 1 and the next ticket are in LDS
        const int next_tile = __builtin_amdgcn_readfirstlane(s_ticket[tslot ^ 1]);
#if OM_W14_TRACE
        unsigned long long pt0, pt1, pt2, pt3 = 0;
        W14_STAMP(pt0);
#endif
        // fragments are read one step ahead, across the group barrier too: weight group g + 1 -- requested in group g - 1
        // -- is waited for at the END of group g - 1 (one group of matrix work for 12 KiB from L2), and the next chunk's V is
        // published by the barrier that ends group 4
        f32x4 ca[4], na[4];        // A hi, A lo, B hi, B lo of the current / next step
        auto read_a = [&](f32x4(&f)[4], const f32x4* sV, int j, int ky) {
            f[0] = sV[w14_plane(j) * W14_VPLANE + aoff_hi[ky]];
            f[1] = sV[w14_plane(j) * W14_VPLANE + aoff_lo[ky]];
        };
        auto read_b = [&](f32x4(&f)[4], int ky, int slot) {
            f[2] = s_u[slot * W14_UGRP + ky * (W14_BN * 4) + boff_hi];
            f[3] = s_u[slot * W14_UGRP + ky * (W14_BN * 4) + boff_lo];
        };
        auto read_frags = [&](f32x4(&f)[4], const f32x4* sV, int j, int ky, int slot) {
            if constexpr (OM_W14_ABLATE & 2) {
                f[0] = f[1] = f[2] = f[3] = f32x4{(float)(j + ky), 1.f, 2.f, (float)lane};
            } else {
                read_a(f, sV, j, ky);
                read_b(f, ky, slot);
            }
        };
        if (W14_B_ACROSS) read_frags(ca, smem, 0, 0, 0);
        else read_a(ca, smem, 0, 0);
        // weights first: D[i = channel][j = entry]
        auto mma = [&](const f32x4(&f)[4], auto plc) {
            constexpr int pl = decltype(plc)::value;
            const f16x8 ah = __builtin_bit_cast(f16x8, f[0]), al = __builtin_bit_cast(f16x8, f[1]);
            const f16x8 bh = __builtin_bit_cast(f16x8, f[2]), bl = __builtin_bit_cast(f16x8, f[3]);
            if constexpr (OM_W14_ABLATE & 1024) {
                asm volatile("" ::"v"(ah), "v"(al), "v"(bh), "v"(bl));
            } else {
                acc[pl] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, al, acc[pl], 0, 0, 0);
                acc[pl] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, ah, acc[pl], 0, 0, 0);
                acc[pl] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ah, acc[pl], 0, 0, 0);
            }
        };
        int g = 0;
        for (int c = 0; c < p.nch; ++c) {
            const f32x4* sV = smem + (c & 1) * W14_VBUF;
            [[maybe_unused]] const f32x4* sVn = smem + ((c & 1) ^ 1) * W14_VBUF;
            // (six explicit instances, not a loop the optimizer may decline to unroll: acc[] must stay in registers)
            auto group = [&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const int slot = g % W14_RING, slot1 = slot == W14_RING - 1 ? 0 : slot + 1;
#if OM_W14_TRACE
                unsigned long long ta, tb = 0, tc = 0, td;
                W14_STAMP(ta);
#endif
                // groups g and g + 1 are in LDS; every wave has left group g - 1: its slot takes group g + 2 (requested below),
                // which has to land by the end of this group
                if (W14_DMA_AFTER < 0) issue_group(ubase, g + W14_RING - 1);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    if constexpr (OM_W14_ABLATE & 2048) continue;       // idle consumers: the producers' own speed
                    if (!W14_B_ACROSS && ky == 0) read_b(ca, 0, slot);       // this group's weights: published by the barrier just passed
                    if (ky < 2) read_frags(na, sV, j, ky + 1, slot);
                    else if (W14_B_ACROSS) {
                        if (j < 5) read_frags(na, sV, j + 1, 0, slot1);
                        else if (c + 1 < p.nch) read_frags(na, sVn, 0, 0, slot1);
                    } else {
                        if (j < 5) read_a(na, sV, j + 1, 0);
                        else if (c + 1 < p.nch) read_a(na, sVn, 0, 0);
                    }
                    mma(ca, std::integral_constant<int, w14_plane(j)>{});
#pragma unroll
                    for (int i = 0; i < 4; ++i) ca[i] = na[i];
                    // the request (60-200 cycles of issue) behind the first matrix instructions, not in front of them
                    if (ky == W14_DMA_AFTER) {
                        issue_group(ubase, g + W14_RING - 1);
                    }
                }
                // my pieces of weight group g + 2 have landed; my reads of this group's slot and plane are done
                if (OM_W14_ABLATE & 131072) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // measurement: late weights are not waited for
                else if ((W14_B_ACROSS && W14_RING == 3) || g + W14_RING - 1 >= ngroups) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                else if (W14_DMA_WAVES) {
                    if (wave < 4) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
                    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                         // these waves requested nothing
                }
                else if (wave < 4) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");      // all but the pieces requested in this group
                else asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
#if OM_W14_TRACE
                W14_STAMP(td);
#endif
                if (!(OM_W14_ABLATE & 4)) __builtin_amdgcn_s_barrier();
#if OM_W14_TRACE
                W14_SETTLE2(ta, td);
                if (first_tile) w14_trace_put(p, wave, g, ta, tb, tc, td);
#endif
                ++g;
            };
            group(std::integral_constant<int, 0>{});
            group(std::integral_constant<int, 1>{});
            group(std::integral_constant<int, 2>{});
            group(std::integral_constant<int, 3>{});
            group(std::integral_constant<int, 4>{});
            group(std::integral_constant<int, 5>{});
        }
#if OM_W14_TRACE
        W14_STAMP(pt1);
#endif
        // every operand in LDS is dead (the last group's barrier has passed): the next tile's first two weight groups are
        // requested before the epilogue, its first chunk is being transformed by the producers meanwhile
        Wino14Tile tn = tl;
        if (next_tile < p.total_tiles) wino14_decode(p, next_tile, tn);
        wino14_epilogue<MODE>(p, acc, tl, smem + W14_VBUF + wave * 256, wm, wn, lane, [&]() {
            if (next_tile < p.total_tiles) {
                const int ub = tn.tile_n * p.nch * 6 * (W14_UGRP * 16);
                issue_group(ub, 0);
                issue_group(ub, 1);
                if (W14_RING == 4) issue_group(ub, 2);
            }
        });
#if OM_W14_TRACE
        W14_STAMP(pt2);
        W14_SETTLE(pt0, pt1, pt2, pt3);
        if (wave == 0 && first_tile) w14_trace_put(p, 0, 60, pt0, pt1, pt2, pt3);
        if (wave == 1 && n_traced < 8) w14_trace_put(p, 1, 48 + n_traced, pt0, pt1, pt2, pt3);      // phases of this workgroup's first eight tiles
        ++n_traced;
        first_tile = false;
#endif
        tl = tn;
        tile = next_tile;
        tslot ^= 1;
    }
}

// =================================================================================================================================
// Round 6: the same layer as TWO kernels with a 128 x 128 tile -- for the layers whose input is small next to their weights (the
// 512 -> 1024 layers at 1/32 scale).  What bounds the fused kernel above is its ingest and instruction count per matrix instruction
// (DESIGN.md 3.9); the only tile with more products per ingested byte that HIP source can express is eight waves of 64 entries x 32
// output channels x six planes (192 accumulators, two waves per SIMD), which leaves no registers for producer waves: the transformed
// input V must be made elsewhere and arrive like the weights do, by LDS-DMA.  Measured with V's traffic and the real epilogue
// (profiles/r06_w14_bigtile_probe.txt) that tile is 13-20 % faster than the fused kernel BEFORE V is paid for; a pre-pass that
// writes V costs 2.5 x the layer's input through HBM, which only the 17^2 layers (18.9 MB of input for 87 GFLOP) can afford.
//   wino14_v_kernel        V[c][j][G][t] = the fused kernel's producers' arithmetic (same formulas, same fused multiply-adds, same
//                          hi/lo split), one 64-byte entry [8 hi | 8 hi | 8 lo | 8 lo] per (16-channel chunk c, plane j, padded row
//                          G = b (H + 2) + y + 1, tile column t); rows outside the image are never written (the consumer reads them
//                          as zero entries through its buffer descriptor)
//   wino14_wide_kernel     per chunk: V's six planes x 144 entries (54 KiB, rows of Ct entries gathered: whole lines) into the other
//                          V buffer, one plane per group; weights through a two-slot ring of 24-KiB groups (one plane's three kernel
//                          rows x 128 output channels); 18 matrix instructions per wave and group; the fused kernel's epilogue on
//                          both 32 x 32 sub-tiles of a wave.  Same products in the same order as the fused kernel: BIT-IDENTICAL
//                          outputs (tests/test_hip_parity.py::test_wino14_wide_equals_fused).
constexpr int W14B_UGRP = 2 * W14_UGRP;      // one plane's three kernel rows x 128 output channels: 24 KiB

struct Wino14VParams {
    const float* in;
    _Float16* v;
    int B, H, W, in_ps, in_bytes, nch, TW, gtot;
    long long items;        // B * H * TW * nch * 2 (channel octets)
};

__device__ __forceinline__ f32x4 w14_point(const f32x4 (&d)[6], int j) {
    const f32x4 c4 = 4.f, cm4 = -4.f, c2 = 2.f, cm2 = -2.f, cm5 = -5.f;
    switch (j) {        // the producers' `point` (wino14_split_kernel), operation for operation
        case 0: return __builtin_elementwise_fma(d[2], cm5, __builtin_elementwise_fma(d[0], c4, d[4]));
        case 1: return __builtin_elementwise_fma(d[1] + d[2], cm4, d[3] + d[4]);
        case 2: return __builtin_elementwise_fma(d[1] - d[2], c4, d[4] - d[3]);
        case 3: return __builtin_elementwise_fma(d[3] - d[1], c2, d[4] - d[2]);
        case 4: return __builtin_elementwise_fma(d[3] - d[1], cm2, d[4] - d[2]);
        default: return __builtin_elementwise_fma(d[3], cm5, __builtin_elementwise_fma(d[1], c4, d[5]));
    }
}

__global__ __launch_bounds__(256) void wino14_v_kernel(const Wino14VParams p) {
    const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
    if (id >= p.items) return;
    // id = ((row * nch + c) * TW + t) * 2 + o: the two octets of an entry and neighbouring entries of a row are adjacent lanes
    const int o = (int)(id & 1);
    long long r = id >> 1;
    const int t = (int)(r % p.TW); r /= p.TW;
    const int c = (int)(r % p.nch); r /= p.nch;
    const int b = (int)(r / p.H), y = (int)(r - (long long)b * p.H);
    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
    f32x4 d[2][6];
#pragma unroll
    for (int x = 0; x < 6; ++x) {
        const int px = 4 * t - 1 + x;
        const bool ok = (unsigned)px < (unsigned)p.W;
        const int off = ok ? (((b * p.H + y) * p.W + px) * p.in_ps + 16 * c + 8 * o) * 4 : (int)0x80000000;
        d[0][x] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 0, 0));
        d[1][x] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, ok ? off + 16 : off, 0, 0));
    }
    const long long G = (long long)b * (p.H + 2) + y + 1;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        f16x8 h, l;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const f32x4 v = w14_point(d[q], j);
            const f16x4 hh = __builtin_convertvector(v, f16x4);
            f32x4 rem;
#pragma unroll
            for (int k = 0; k < 4; ++k) rem[k] = __builtin_fmaf((float)hh[k], -1.0f, v[k]);      // exact: what v_fma_mix_f32 computes in the producers
            const f16x4 ll = __builtin_convertvector(rem, f16x4);
#pragma unroll
            for (int k = 0; k < 4; ++k) { h[4 * q + k] = hh[k]; l[4 * q + k] = ll[k]; }
        }
        _Float16* e = p.v + ((((long long)c * 6 + j) * p.gtot + G) * p.TW + t) * 32;
        *reinterpret_cast<f16x8*>(e + 8 * o) = h;
        *reinterpret_cast<f16x8*>(e + 16 + 8 * o) = l;
    }
}

struct Wino14WideParams {
    Wino14Params k;         // the fused kernel's parameters (epilogue, weights, geometry); k.in / k.in_bytes are not used
    const _Float16* v;
    int v_bytes, TW;
};

template <int MODE>
__global__ __launch_bounds__(512, 2) void wino14_wide_kernel(const Wino14WideParams pw) {
    const Wino14Params& p = pw.k;
    // two V buffers (chunk parity) + two weight slots of 24 KiB.  (Measured against it and not kept: ONE V buffer refilled plane by
    // plane behind the group that read it + a four-slot weight ring requested three groups ahead -- bit-identical, 0.238 against
    // 0.224-0.230 ms on 17^2 512 -> 1024: the weights' landing time is not what a group waits for.)
    __shared__ f32x4 smem[2 * W14_VBUF + 2 * W14B_UGRP + 1];
    f32x4* const s_u = smem + 2 * W14_VBUF;
    int* const s_ticket = reinterpret_cast<int*>(smem + 2 * W14_VBUF + 2 * W14B_UGRP);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm2 = wave >> 2, wn4 = wave & 3;
    const int fi = lane & 31, fk = lane >> 5;
    const int ngroups = 6 * p.nch;
    const int n_tiles2 = p.n_tiles >> 1, total2 = (p.total_tiles / p.n_tiles) * n_tiles2;
    const int TW = pw.TW, hp2 = p.H + 2;
    const int ecount = (p.R + 2) * p.Ct;
    const auto rs_u = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.u), 0, p.u_bytes, 0x00020000);
    const auto rs_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(pw.v), 0, pw.v_bytes, 0x00020000);
    // weights: 24 pieces of 1 KiB per group: wave w requests pieces w, w + 8, w + 16 (0-11: the first 64-channel half, 12-23 the second)
    const int drow = lane >> 2, dcol = lane & 3;
    int dvo[3], dhalf[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int piece = wave + 8 * i;
        dhalf[i] = piece >= 12;
        const int row = 16 * (piece - 12 * dhalf[i]) + drow;
        dvo[i] = row * 64 + ((dcol ^ ((row >> 2) & 3)) * 16);      // swizzle on the SOURCE chunk: the LDS image stays lane-linear
    }
    const int swB = (fi >> 2) & 3;
    const int brow = (wn4 >> 1) * (W14_UGRP) + (32 * (wn4 & 1) + fi) * 4;       // half, then row of the tap's 64 rows
    const int boff_hi = brow + (fk ^ swB), boff_lo = brow + ((2 + fk) ^ swB);
    int aoff_hi[3], aoff_lo[3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int e = 64 * wm2 + fi + ky * p.Ct;
        const int sw = (e >> 2) & 3;
        aoff_hi[ky] = e * 4 + (fk ^ sw);
        aoff_lo[ky] = e * 4 + ((2 + fk) ^ sw);
    }
    auto issue_group = [&](int tn2, int g) {
        if (g >= ngroups) return;
        const int slot = g & 1;
        const int c = g / 6, j = g - 6 * c;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int t64 = 2 * tn2 + dhalf[i];
            const int soff = ((t64 * p.nch + c) * 6 + w14_plane(j)) * (W14_UGRP * 16);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_u, (lds_ptr_t)(s_u + slot * W14B_UGRP + (wave + 8 * i) * 64), 16, dvo[i], soff, 0, 0);
        }
    };
    // V: a plane of a chunk is 144 entries x 64 B = nine 1-KiB pieces; wave w requests piece w, wave 0 also piece 8.  Entry e of the
    // block = padded row g0 - 1 + e / Ct, tile column t0 + e % Ct; rows outside an image, columns beyond the row's tiles and entries
    // beyond the block get an out-of-range offset: the DMA writes zeros (the fused kernel's zero entries).
    int vsrc[2];
    auto setup_v = [&](const Wino14Tile& tl) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = 16 * (wave + 8 * i) + drow;
            const int rr = e / p.Ct, t = e - rr * p.Ct;
            const int sw = (e >> 2) & 3;
            const int g = tl.g0 - 1 + rr;
            const int b = g / hp2;
            const int y = g - b * hp2 - 1;
            const bool ok = e < ecount && g >= 0 && g < p.gtot && y >= 0 && y < p.H && tl.t0 + t < TW;
            vsrc[i] = ok ? ((g * TW + tl.t0 + t) * 64 + ((dcol ^ sw) * 16)) : (int)0x80000000;
        }
    };
    auto issue_v_plane = [&](int c, int pl, int vb) {       // plane pl of chunk c into V buffer vb
        if (c >= p.nch) return;
        const int soff = (c * 6 + pl) * (p.gtot * TW * 64);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_v, (lds_ptr_t)(smem + vb * W14_VBUF + pl * W14_VPLANE + wave * 64), 16, vsrc[0], soff, 0, 0);
        if (wave == 0)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_v, (lds_ptr_t)(smem + vb * W14_VBUF + pl * W14_VPLANE + 8 * 64), 16, vsrc[1], soff, 0, 0);
    };
    f32x16 acc[2][6];
    for (;;) {
        if (tid == 0) s_ticket[0] = atomicAdd(p.ticket + 8, 1);
        __syncthreads();
        const int tile = __builtin_amdgcn_readfirstlane(s_ticket[0]);
        __syncthreads();            // (also: every wave is done with the previous tile's epilogue transposes in V buffer 1)
        if (tile >= total2) break;
        const int tn2 = tile % n_tiles2, tm = tile / n_tiles2;      // N fastest: the N-tile siblings read the same V block from L2
        Wino14Tile tl;
        tl.g0 = (tm / p.ncb) * p.R; tl.t0 = (tm % p.ncb) * p.Ct; tl.tile_n = 2 * tn2 + (wn4 >> 1); tl.n0 = tl.tile_n * W14_BN;
        setup_v(tl);
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[b][j][r] = 0.f;
        issue_group(tn2, 0);
#pragma unroll
        for (int pl = 0; pl < 6; ++pl) issue_v_plane(0, pl, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        int g = 0;
        for (int c = 0; c < p.nch; ++c) {
            const f32x4* sV = smem + (c & 1) * W14_VBUF;
            auto group = [&](auto jc) {
                constexpr int j = decltype(jc)::value;
                constexpr int pl = w14_plane(j);
                const int slot = g & 1;
                issue_group(tn2, g + 1);                    // the other slot: every wave left group g - 1 at the barrier just passed
                issue_v_plane(c + 1, pl, (c & 1) ^ 1);      // the other V buffer: chunk c - 1 is done with
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    // all six fragments of the step, ONE wait, six matrix instructions back to back (left to itself the compiler --
                    // out of registers beside 192 accumulators -- reads a fragment, waits, multiplies, reads the next: an LDS round
                    // trip in front of every one or two matrix instructions); the other wave of the SIMD multiplies meanwhile
                    f32x4 fr[6];
                    fr[0] = s_u[slot * W14B_UGRP + ky * (W14_BN * 4) + boff_hi];
                    fr[1] = s_u[slot * W14B_UGRP + ky * (W14_BN * 4) + boff_lo];
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        fr[2 + 2 * b] = sV[pl * W14_VPLANE + aoff_hi[ky] + 128 * b];
                        fr[3 + 2 * b] = sV[pl * W14_VPLANE + aoff_lo[ky] + 128 * b];
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fr[0]), "+v"(fr[1]), "+v"(fr[2]), "+v"(fr[3]), "+v"(fr[4]), "+v"(fr[5])::"memory");
                    const f16x8 bhh = __builtin_bit_cast(f16x8, fr[0]), bll = __builtin_bit_cast(f16x8, fr[1]);
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const f16x8 ahh = __builtin_bit_cast(f16x8, fr[2 + 2 * b]), all = __builtin_bit_cast(f16x8, fr[3 + 2 * b]);
                        // the fused kernel's three products in its order: bit-identical sums
                        acc[b][pl] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bhh, all, acc[b][pl], 0, 0, 0);
                        acc[b][pl] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bll, ahh, acc[b][pl], 0, 0, 0);
                        acc[b][pl] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bhh, ahh, acc[b][pl], 0, 0, 0);
                    }
                }
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // the next group's weights and this group's V plane
                __builtin_amdgcn_s_barrier();
                ++g;
            };
            group(std::integral_constant<int, 0>{});
            group(std::integral_constant<int, 1>{});
            group(std::integral_constant<int, 2>{});
            group(std::integral_constant<int, 3>{});
            group(std::integral_constant<int, 4>{});
            group(std::integral_constant<int, 5>{});
        }
        // the fused kernel's epilogue, once per 32 x 32 sub-tile of the wave (entries 64 wm2 + 32 b ..).  With a residual the first
        // call needs ~70 registers beside the 96 accumulators it reads, and the other sub-tile's 96 are still live: the compiler spilled
        // 12-14 of them to scratch -- loads and stores in the one in-order counter the epilogue is built around (+0.024 ms per layer,
        // profiles/r06_experiments.md section 1).  Two planes of the second sub-tile wait in LDS instead (every operand area is dead
        // since the last group's barrier: V buffer 0 for waves 0-5, the weight slots for waves 6-7; lane-linear, 8 KiB per wave).
        f32x4* const park = (wave < 6 ? smem + wave * 512 : s_u + (wave - 6) * 512) + lane;
        if constexpr (MODE == 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                park[(q) * 64] = f32x4{acc[1][4][4 * q], acc[1][4][4 * q + 1], acc[1][4][4 * q + 2], acc[1][4][4 * q + 3]};
                park[(4 + q) * 64] = f32x4{acc[1][5][4 * q], acc[1][5][4 * q + 1], acc[1][5][4 * q + 2], acc[1][5][4 * q + 3]};
            }
        }
        wino14_epilogue<MODE>(p, acc[0], tl, smem + W14_VBUF + wave * 256, 2 * wm2, wn4 & 1, lane, [] {});
        if constexpr (MODE == 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 a4 = park[(q) * 64], a5 = park[(4 + q) * 64];
#pragma unroll
                for (int k = 0; k < 4; ++k) { acc[1][4][4 * q + k] = a4[k]; acc[1][5][4 * q + k] = a5[k]; }
            }
        }
        wino14_epilogue<MODE>(p, acc[1], tl, smem + W14_VBUF + wave * 256, 2 * wm2 + 1, wn4 & 1, lane, [] {});
    }
}

// Block shape for a layer: Ct tile columns (a divisor-like split of ceil(W / 4)) x R padded rows with R * Ct <= 128 and
// (R + 2) * Ct <= 160, picked for the largest share of useful rows in the 128-row matrix tile.
void wino14_geometry(int B, int H, int W, int* R, int* Ct, int* ncb, int* nrb) {
    const int TW = (W + 3) / 4;
    const long long gtot = (long long)B * (H + 2);
    double best = -1.0;
    for (int split = 1; split <= 8; ++split) {
        const int ct = (TW + split - 1) / split;
        if (ct > 64 || ct < 1) continue;
        int r = W14_BM / ct;
        while (r > 1 && (r + 2) * ct > W14_EMAX) --r;
        if (r < 1 || (r + 2) * ct > W14_EMAX) continue;
        if (r > gtot) r = (int)gtot;
        const int cbs = (TW + ct - 1) / ct;
        const long long rbs = (gtot + r - 1) / r;
        const double useful = (double)B * H * TW;
        const double util = useful / ((double)rbs * cbs * W14_BM);
        if (util > best) { best = util; *R = r; *Ct = ct; *ncb = cbs; *nrb = (int)rbs; }
    }
}

static int g_w14_variant = -1;
int wino14_variant() {
    if (g_w14_variant < 0) {
        const char* e = std::getenv("OM_W14_VARIANT");
        g_w14_variant = e ? std::atoi(e) : 0;
    }
    return g_w14_variant;
}
void wino14_set_variant(int v) { g_w14_variant = v; }

size_t wino14_weight_halfs(int cout_pad, int cin) { return (size_t)18 * cout_pad * cin * 2; }

// the fused kernel's parameters from a layer's arguments (shared by the two-kernel wide form)
static int wino14_fill_params(const ConvArgs& a, Wino14Params& p) {
    OM_REQUIRE(a.in && a.w && a.scale && a.shift && a.out && a.ticket, OM_EINVAL, "wino14: null pointer");
    OM_REQUIRE(a.ks == 3 && a.stride == 1 && a.out_mode == 0, OM_EINVAL, "wino14: 3x3 stride-1 NHWC layers only");
    OM_REQUIRE(a.cin % 16 == 0 && a.cin >= 16 && a.cout_pad % 64 == 0, OM_EINVAL, "wino14: cin=%d cout_pad=%d", a.cin, a.cout_pad);
    OM_REQUIRE(a.in_pix_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(a.in) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(a.w) & 15) == 0,
               OM_EINVAL, "wino14: operands must be 16-byte aligned");
    const long long in_bytes = ((long long)a.B * a.H * a.W - 1) * a.in_pix_stride * 4 + (long long)a.cin * 4;
    OM_REQUIRE(in_bytes < 0x7FFFFFF0ll, OM_EINVAL, "wino14: input view of %lld bytes exceeds a buffer descriptor", in_bytes);
    p.in = a.in; p.u = reinterpret_cast<const _Float16*>(a.w); p.scale = a.scale; p.shift = a.shift; p.res = a.res; p.out = a.out;
    p.ticket = a.ticket; p.status = a.status;
    p.B = a.B; p.H = a.H; p.W = a.W; p.in_ps = a.in_pix_stride; p.in_bytes = (int)in_bytes;
    p.cout = a.cout; p.out_ps = a.out_pix_stride; p.res_ps = a.res_pix_stride; p.leaky = a.leaky;
    // the epilogue's buffer-descriptor form: 16-byte aligned views below 2 GiB, whole channel quads
    const long long npix = (long long)a.B * a.H * a.W;
    const long long out_bytes = ((npix - 1) * a.out_pix_stride + a.cout) * 4;
    const long long res_bytes = a.res ? ((npix - 1) * a.res_pix_stride + a.cout) * 4 : 0;
    p.fast_io = (a.cout % 4 == 0 && a.out_pix_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0 &&
                 (!a.res || (a.res_pix_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(a.res) & 15) == 0)) &&
                 out_bytes < 0x7FFFFFF0ll && res_bytes < 0x7FFFFFF0ll)
                    ? 1 : 0;
    p.out_bytes = p.fast_io ? (int)out_bytes : 0;
    p.res_bytes = p.fast_io ? (int)res_bytes : 0;
    OM_REQUIRE(npix < (1ll << 31), OM_EINVAL, "wino14: %lld pixels out of range", npix);
    int R = 0, Ct = 0, ncb = 0, nrb = 0;
    wino14_geometry(a.B, a.H, a.W, &R, &Ct, &ncb, &nrb);
    OM_REQUIRE(R >= 1 && Ct >= 1, OM_EINVAL, "wino14: no block shape for %d x %d", a.H, a.W);
    p.R = R; p.Ct = Ct; p.ncb = ncb; p.gtot = a.B * (a.H + 2);
    p.n_tiles = a.cout_pad / W14_BN;
    p.nch = a.cin / 16;
    const long long total = (long long)nrb * ncb * p.n_tiles;
    OM_REQUIRE(total > 0 && total < (1ll << 31), OM_EINVAL, "wino14: %lld tiles out of range", total);
    p.total_tiles = (int)total;
    const size_t ub = wino14_weight_halfs(a.cout_pad, a.cin) * 2;
    OM_REQUIRE(ub < 0x7FFFFFF0ull, OM_EINVAL, "wino14: weights exceed a buffer descriptor");
    p.u_bytes = (int)ub;
#if OM_W14_TRACE
    p.trace = g_w14_trace;
    OM_REQUIRE(p.trace, OM_EINVAL, "wino14 trace build: om_debug_w14_trace() first");
#endif
    return OM_OK;
}

// a.w: the packed F(4,3) weights (include/orienmask_hip.h: om_layer_info.wsplit_off for wino layers); a.scale: scale * 2^-e
int launch_conv_wino14_split(const ConvArgs& a, hipStream_t stream) {
    Wino14Params p;
    if (int rc = wino14_fill_params(a, p)) return rc;
    const long long total = p.total_tiles;
    // round 5: the four-dual-role-wave form (conv_wino14d.hip) on request only (om_set_wino14_variant(1) / OM_W14_VARIANT=1): bit-identical,
    // but measured 8-25 % slower than this file's twelve-wave kernel on every layer shape (profiles/r05_experiments.md 1).  Since round 6
    // it is only in libraries built with `make W14D=1` (the default library holds no kernel the forward cannot reach).
#ifdef OM_WITH_W14D
    if (wino14_variant() == 1 && wino14_dual_supported(p)) return launch_wino14_dual(p, a.res != nullptr, stream);
#endif
    const long long grid = total < 256 ? total : 256;        // one 768-thread workgroup per CU (156 KiB of LDS)
    if (!p.fast_io) hipLaunchKernelGGL(wino14_split_kernel<2>, dim3((unsigned)grid), dim3(W14_THREADS), 0, stream, p);
    else if (a.res) hipLaunchKernelGGL(wino14_split_kernel<1>, dim3((unsigned)grid), dim3(W14_THREADS), 0, stream, p);
    else hipLaunchKernelGGL(wino14_split_kernel<0>, dim3((unsigned)grid), dim3(W14_THREADS), 0, stream, p);
    OM_CHECK_HIP(hipGetLastError());
    return OM_OK;
}

// ---- the two-kernel wide form (wino14_v_kernel + wino14_wide_kernel)
// floats of scratch for V: [cin / 16][6][B (H + 2)][ceil(W / 4)] entries of 64 bytes
size_t wino14_wide_scratch_floats(int B, int H, int W, int cin) {
    return (size_t)(cin / 16) * 6 * B * (H + 2) * ((W + 3) / 4) * 16;
}

// can this layer run it?  Whole pairs of 64-channel N tiles, the epilogue's buffer-descriptor form, V below 2 GiB.
bool wino14_wide_supported(const ConvArgs& a) {
    if (a.ks != 3 || a.stride != 1 || a.out_mode != 0 || a.cin % 16 || a.cout_pad % 128 || a.cout % 4 || a.out_pix_stride % 4) return false;
    if ((reinterpret_cast<uintptr_t>(a.out) & 15) || (a.res && ((a.res_pix_stride % 4) || (reinterpret_cast<uintptr_t>(a.res) & 15)))) return false;
    return wino14_wide_scratch_floats(a.B, a.H, a.W, a.cin) * 4 < 0x7FFFFFF0ull;
}

// which layers take it (om_forward's choice; the unit entry om_conv2d_wino14_wide runs it on any supported layer): the pre-pass moves
// 2.5 x the layer's input through HBM, the wide tile saves 13-20 % of the fused kernel's time -- it pays where the input is small next
// to the layer's work, i.e. from 512 input channels on (the 512 -> 1024 layers at 1/32 scale: 47 MB of pre-pass traffic for 87 GFLOP;
// profiles/r06_experiments.md section 1)
bool wino14_wide_pays(const ConvArgs& a) { return a.cin >= 512 && wino14_wide_supported(a); }

int launch_conv_wino14_wide(const ConvArgs& a, float* scratch, hipStream_t stream) {
    OM_REQUIRE(scratch && (reinterpret_cast<uintptr_t>(scratch) & 15) == 0, OM_EINVAL, "wino14 wide: scratch must be 16-byte aligned");
    OM_REQUIRE(wino14_wide_supported(a), OM_EINVAL, "wino14 wide: cout_pad=%d must be a multiple of 128 and the views 16-byte aligned", a.cout_pad);
    Wino14WideParams pw;
    if (int rc = wino14_fill_params(a, pw.k)) return rc;
    OM_REQUIRE(pw.k.fast_io && pw.k.n_tiles % 2 == 0, OM_EINVAL, "wino14 wide: views not in the epilogue's buffer-descriptor form");
    const int TW = (a.W + 3) / 4;
    pw.TW = TW;
    pw.v = reinterpret_cast<const _Float16*>(scratch);
    pw.v_bytes = (int)(wino14_wide_scratch_floats(a.B, a.H, a.W, a.cin) * 4);
    Wino14VParams pv;
    pv.in = a.in; pv.v = reinterpret_cast<_Float16*>(scratch);
    pv.B = a.B; pv.H = a.H; pv.W = a.W; pv.in_ps = a.in_pix_stride; pv.in_bytes = pw.k.in_bytes; pv.nch = pw.k.nch; pv.TW = TW; pv.gtot = pw.k.gtot;
    pv.items = (long long)a.B * a.H * TW * pv.nch * 2;
    hipLaunchKernelGGL(wino14_v_kernel, dim3((unsigned)((pv.items + 255) / 256)), dim3(256), 0, stream, pv);
    OM_CHECK_HIP(hipGetLastError());
    if (a.mid_event) OM_CHECK_HIP(hipEventRecord(a.mid_event, stream));      // profiling: pre-pass | wide kernel
    const long long total2 = (long long)(pw.k.total_tiles / pw.k.n_tiles) * (pw.k.n_tiles / 2);
    const unsigned grid = (unsigned)(total2 < 256 ? total2 : 256);
    if (a.res) hipLaunchKernelGGL(wino14_wide_kernel<1>, dim3(grid), dim3(512), 0, stream, pw);
    else hipLaunchKernelGGL(wino14_wide_kernel<0>, dim3(grid), dim3(512), 0, stream, pw);
    OM_CHECK_HIP(hipGetLastError());
    return OM_OK;
}

}  // namespace om
