// Fused stride-1 3x3 convolution with SPLIT fp32 operands, round 5: FOUR DUAL-ROLE WAVES, one per SIMD.
//
// Same layer, same algorithm (Winograd F(4,3) along the image rows, the three kernel rows part of the contraction, input
// transform on chip), same packed weights, same 128 x 64 tile and block geometry and the same sequence of fp32 operations per
// output as conv_wino14.hip -- the two kernels are bit-identical (tests/test_hip_parity.py::test_wino14_dual_equals_twelve_wave)
// -- in another execution structure.  conv_wino14.hip runs eight consumer waves (two per SIMD, 96 accumulators each) and four
// producer waves that meet at a barrier per group of nine matrix instructions per wave; round 4 measured what that costs
// (profiles/r04_experiments.md 1-4): three parties at every group barrier (1150-1300 cycles per group for 576 cycles of matrix
// work), 30 % for the producers as a whole, 1.33 KB of LDS reads per matrix instruction.  Here (Conv2d 3x3 s1 p1 ->
// BatchNorm2d(eval) -> LeakyReLU(0.1) (+ residual): /root/reference/model/base.py:104-137, model/backbone/darknet.py:6-15):
//
//   * 256 threads = 4 waves, ONE PER SIMD with the whole 512-register file: a wave owns 64 entries x 32 channels x 6 planes
//     (192 accumulators) and does everything for its share in ONE instruction stream -- 18 matrix instructions per group with
//     the producer's work of the group (a quarter of the workgroup's transform: ~45 vector instructions, 5 LDS stores), its
//     three 1-KiB pieces of the weight ring and its fragment reads placed between them.  A B fragment feeds two 32 x 32 blocks:
//     1 KB of LDS reads per matrix instruction, and the only parties at a group's barrier are four waves running the same code.
//   * The 192 accumulators are NOT compiler values: they are the accumulation registers a0 .. a191 by name, written by inline-asm
//     matrix instructions and read by inline-asm copies in the epilogue.  As values of the program, hipcc (ROCm 7.2) copies
//     whole 16-register planes into vector registers wherever an element is used and spills accumulators to scratch (200-300
//     spilled registers, every reload a wait for the whole vector-memory queue); owned by name, nothing of them is ever moved.
//     The statements list them as clobbers, so the compiler keeps out of them: the build must show no spill and no compiler
//     v_accvgpr_* (tools/kernel_resources.sh conv_wino14d; tests/test_host_cpu.py::test_wino14d_has_no_spills).
//   * ONE V buffer (6 planes x 160 entries x 64 B), reused plane by plane: while chunk c is multiplied (plane order 0, 5, 1, 2,
//     3, 4) groups 0-1 write planes (3, 4) of chunk c ITSELF (read from group 4 on), groups 2-3 planes (0, 5) of chunk c + 1 (their
//     slots were read last in groups 0 and 1), groups 4-5 planes (1, 2) of chunk c + 1.  The LDS this frees holds a weight ring of
//     EIGHT 12-KiB groups requested seven groups ahead: in one wave's stream the weight requests and the input requests retire
//     through the same in-order counter, so "my pieces of group g + 2 have landed" also waits for every older input request --
//     with eight slots those are at least six groups (one chunk) old, what the first of the N-tile siblings' trip to HBM takes.
//   * The chunk stream runs ACROSS tiles: the last chunk of a tile writes the first planes of the next tile's first chunk, the
//     weight ring never drains; between two tiles there is only the epilogue (inverse transform, transpose, scale / shift /
//     LeakyReLU / residual, stores: wino14_shared.h's arithmetic) and one barrier.  Input pixels live in two register sets
//     (chunk parity), requested a whole chunk before their first use -- except the next tile's second chunk, requested behind
//     the epilogue (its set is the epilogue's working registers).
//
// Requires an even number of 16-channel chunks >= 2 (every layer of the network: cin = 32 ... 512) and the buffer-descriptor
// epilogue (16-byte aligned views).
//
// MEASURED (profiles/r05_experiments.md 1): 8-25 % SLOWER than conv_wino14.hip on every layer shape (136 x 136 128 -> 256: 0.89 ms
// against 0.73).  A lone wave per SIMD overlaps only about half of its other instructions with its own matrix instructions: a
// group costs 900-1000 cycles of issue for 576 cycles of matrix work, the other instructions alone 760-800.  The kernel stays
// as a tested alternative (om_set_wino14_variant(1) / OM_W14_VARIANT=1); om_forward runs conv_wino14.hip.
#include "wino14_shared.h"

namespace om {

#ifndef WD_RING
#define WD_RING 8
#endif
#ifndef OM_WD_ABLATE
#define OM_WD_ABLATE 0         // measurement builds only (wrong numerics): 1 no epilogue, 2 no transform / V stores, 4 no pixel requests,
#endif                         // 8 no weight requests, 16 no matrix instructions, 32 no fragment reads, 64 no group barrier
#ifndef OM_WD_TRACE
#define OM_WD_TRACE 0          // measurement builds only: s_memtime stamps of every group of the first workgroups (tools/wd_trace.py)
#endif
#if (OM_WD_ABLATE || OM_WD_TRACE) && !defined(OM_MEASUREMENT_BUILD)
#error "measurement switches (wrong numerics) are only for ab/ variants: build them with tools/build_variant.sh"
#endif
constexpr int WD_VENT = 160;                        // LDS entries per plane: every item of every thread has one (128 + 256 / 8), so the
                                                    // producer's stores are unconditional; entries beyond (R + 2) Ct hold zeros
constexpr int WD_VPLANE = WD_VENT * 4;              // f32x4 units (16 B) per plane: 10240 B
constexpr int WD_VBUF = 6 * WD_VPLANE;              // the V buffer: 61440 B
constexpr int WD_UGRP = 3 * W14_BN * 4;             // one (j; ky = 0..2) weight group: 12288 B
constexpr int WD_THREADS = 256;
static_assert((WD_RING & (WD_RING - 1)) == 0 && WD_RING >= 4, "ring slots: a power of two");
static_assert((WD_VBUF + WD_RING * WD_UGRP + 1) * 16 <= 160 * 1024, "LDS");
static_assert(W14_EMAX_ALL <= WD_VENT && 128 + WD_THREADS / 8 <= WD_VENT, "entries");

// Vector-memory operations of one wave per group position q, steady state: three weight pieces at the top of every group, the
// 18 pixel requests of a chunk in groups 2 and 3 (nine each).  A wave waits for "my pieces of group G + 2" at the end of group G:
// they were requested at the top of group G + 3 - RING, so the operations younger than them are the pieces of RING - 3 groups and
// the pixel requests of the RING - 2 groups G + 3 - RING .. G.
__host__ __device__ constexpr int wd_loads_in(int q) { return (q == 2 || q == 3) ? 9 : 0; }
__host__ __device__ constexpr int wd_younger(int q) {
    int n = 3 * (WD_RING - 3);
    for (int k = 0; k < WD_RING - 2; ++k) n += wd_loads_in(((q - k) % 6 + 6) % 6);
    return n;
}

#if OM_WD_TRACE
// [workgroup < 8][wave][group < 512][4]: start of the group (behind the barrier), last matrix instruction issued, end of the
// group's wait (in front of the barrier), and at a tile's last group the end of the epilogue
static unsigned long long* g_wd_trace = nullptr;
extern "C" void om_debug_wd_trace(void* buf) { g_wd_trace = static_cast<unsigned long long*>(buf); }
#define WD_STAMP(x) asm volatile("s_memtime %0" : "=s"(x)::"memory")
#endif

// ---------------------------------------------------------------- the accumulators: a[16 N .. 16 N + 15], N = 6 block + plane
template <int N>
__device__ __forceinline__ void wd_mfma(const f16x8& a, const f16x8& b) {
    if constexpr (OM_WD_ABLATE & 16) { asm volatile("" ::"v"(a), "v"(b)); return; }
    asm volatile("v_mfma_f32_32x32x16_f16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(a), "v"(b), "i"(16 * N), "i"(16 * N + 15) :
#include "wino14d_acc.inc"
    );
}
// ... the first of a step.  Its operands come straight from LDS reads; should the compiler ever put a register copy (a vector-ALU
// write) in front of it, the matrix instruction needs two wait states the compiler does not pad inside an asm:
// tests/test_host_cpu.py::test_wino14d_isa_audit checks the emitted code for that
template <int N>
__device__ __forceinline__ void wd_mfma_first(const f16x8& a, const f16x8& b) {
    if constexpr (OM_WD_ABLATE & 16) { asm volatile("" ::"v"(a), "v"(b)); return; }
    asm volatile("v_mfma_f32_32x32x16_f16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(a), "v"(b), "i"(16 * N), "i"(16 * N + 15) :
#include "wino14d_acc.inc"
    );
}
template <int N>
__device__ __forceinline__ void wd_acc_zero(const f16x8& z) {       // 0 x 0 + 0
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 a[%c1:%c2], %0, %0, 0" ::"v"(z), "i"(16 * N), "i"(16 * N + 15) :
#include "wino14d_acc.inc"
    );
}
template <int R>
__device__ __forceinline__ float wd_acc_read() {
    float x;
    asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(x) : "i"(R));
    return x;
}

// floor(a / d) for 0 <= a < 2^22, 1 <= d: one reciprocal estimate and two corrections (the epilogue's and the items' entry
// coordinates; the compiler's exact 32-bit division is ~35 instructions)
__device__ __forceinline__ int wd_div(int a, int d, float rd) {
    int q = (int)((float)a * rd);
    q -= (q * d > a) ? 1 : 0;
    q += ((q + 1) * d <= a) ? 1 : 0;
    return q;
}

// Epilogue of ONE 32-entry block of a wave (wino14_shared.h: wino14_epilogue's arithmetic, operation for operation): inverse
// transform position by position, 32 x 32 transpose through 4 KiB of the wave's own LDS, scale / shift, LeakyReLU, residual,
// 16-byte stores through buffer descriptors.  Nothing waits for a store; every request is unconditional (out-of-range offsets
// where there is nothing to store), so a wave issues exactly 16 (MODE 0) / 32 (MODE 1) vector-memory operations per block behind
// the two scale / shift loads.  Two passes over the accumulators: positions 0 and 3 (all six planes), then 1 and 2 (planes 1-4).
template <int MODE, int BLK>
__device__ __forceinline__ void wd_epilogue(const Wino14Params& p, const Wino14Tile& tl, f32x4* sT, int mb, int wn, int lane, float& nonfinite) {
    asm volatile("" : "+v"(lane));
    const int fi = lane & 31, fk = lane >> 5;
    const int hp2 = p.H + 2;
    const float r_ct = 1.0f / (float)p.Ct, r_hp2 = 1.0f / (float)hp2;
    const int c8 = lane & 7;
    const int nb = tl.n0 + 32 * wn + 4 * c8;
    const int nvalid = p.cout - nb;
    const f32x4 sc = *reinterpret_cast<const f32x4*>(p.scale + nb);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(p.shift + nb);
    int pix0[4], oxe[4];
#pragma unroll
    for (int rd = 0; rd < 4; ++rd) {
        const int ml = 32 * mb + 8 * rd + (lane >> 3);
        const int r = wd_div(ml, p.Ct, r_ct), t = ml - r * p.Ct;
        const int gg = tl.g0 + r;
        const int b = wd_div(gg, hp2, r_hp2);
        const int y = gg - b * hp2 - 1;
        const bool rowok = r < p.R && gg < p.gtot && y >= 0 && y < p.H && nvalid > 0;
        oxe[rd] = rowok ? 4 * (tl.t0 + t) : p.W;
        pix0[rd] = (b * p.H + y) * p.W + 4 * (tl.t0 + t);
    }
    f32x4 y[2][4];              // [position of the pass][register quad gq]: channels 8 gq + 4 fk .. + 3 of entry fi
    auto form = [&](auto passc) {
        constexpr int PASS = decltype(passc)::value;
        auto quad = [&](auto gqc) {
            constexpr int gq = decltype(gqc)::value;
            auto elem = [&](auto kc) {
                constexpr int k = decltype(kc)::value;
                constexpr int R0 = 16 * (6 * BLK) + 4 * gq + k;     // plane j: R0 + 16 j
                const float a1 = wd_acc_read<R0 + 16>(), a2 = wd_acc_read<R0 + 32>(), a3 = wd_acc_read<R0 + 48>(), a4 = wd_acc_read<R0 + 64>();
                if constexpr (PASS == 0) {
                    const float a0 = wd_acc_read<R0>(), a5 = wd_acc_read<R0 + 80>();
                    y[0][gq][k] = a0 + a1 + a2 + a3 + a4;
                    y[1][gq][k] = (a1 - a2) + 8.f * (a3 - a4) + a5;
                } else {
                    y[0][gq][k] = (a1 - a2) + 2.f * (a3 - a4);
                    y[1][gq][k] = (a1 + a2) + 4.f * (a3 + a4);
                }
            };
            elem(std::integral_constant<int, 0>{}); elem(std::integral_constant<int, 1>{});
            elem(std::integral_constant<int, 2>{}); elem(std::integral_constant<int, 3>{});
        };
        // (scheduling fences: left alone, the scheduler reads every accumulator up front and keeps every intermediate alive -- spills)
        quad(std::integral_constant<int, 0>{}); __builtin_amdgcn_sched_barrier(0);
        quad(std::integral_constant<int, 1>{}); __builtin_amdgcn_sched_barrier(0);
        quad(std::integral_constant<int, 2>{}); __builtin_amdgcn_sched_barrier(0);
        quad(std::integral_constant<int, 3>{}); __builtin_amdgcn_sched_barrier(0);
    };
    auto transpose_in = [&](int slot) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) sT[fi * 8 + ((2 * gq + fk) ^ (fi & 7))] = y[slot][gq];
    };
    auto transpose_out = [&](int rd) {
        const int e = 8 * rd + (lane >> 3);
        return sT[e * 8 + (c8 ^ (e & 7))];
    };
    auto activate = [&](f32x4 v, bool ok) {
        float nf = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float tv = fmaf(v[k], sc[k], sh[k]);
            nf = fmaf(tv, 0.f, nf);
            v[k] = p.leaky ? fmaxf(tv, tv * 0.1f) : tv;
        }
        nonfinite += ok ? nf : 0.f;
        return v;
    };
    const auto rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
    const auto rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res), 0, MODE == 1 ? p.res_bytes : 0, 0x00020000);
    auto offset = [&](int rd, int px, int ps) { return oxe[rd] + px < p.W ? ((pix0[rd] + px) * ps + nb) * 4 : (int)0x80000000; };
    f32x4 rc[4];
    if constexpr (MODE == 1) {
#pragma unroll
        for (int rd = 0; rd < 4; ++rd) rc[rd] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, offset(rd, 0, p.res_ps), 0, 0));
    }
    auto one_position = [&](auto pxc, auto nextc) {
        constexpr int px = decltype(pxc)::value, nx = decltype(nextc)::value;
        transpose_in(px == 0 || px == 1 ? 0 : 1);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 v[4];
#pragma unroll
        for (int rd = 0; rd < 4; ++rd) {
            v[rd] = activate(transpose_out(rd), oxe[rd] + px < p.W);
            if constexpr (MODE == 1) v[rd] += rc[rd];
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (MODE == 1 && nx >= 0) {
#pragma unroll
            for (int rd = 0; rd < 4; ++rd)
                rc[rd] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, offset(rd, nx, p.res_ps), 0, 0));
        }
#pragma unroll
        for (int rd = 0; rd < 4; ++rd)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v[rd]), rs_out, offset(rd, px, p.out_ps), 0, 0);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    using std::integral_constant;
    form(integral_constant<int, 0>{});
    one_position(integral_constant<int, 0>{}, integral_constant<int, 3>{});
    one_position(integral_constant<int, 3>{}, integral_constant<int, 1>{});
    form(integral_constant<int, 1>{});
    one_position(integral_constant<int, 1>{}, integral_constant<int, 2>{});
    one_position(integral_constant<int, 2>{}, integral_constant<int, -1>{});
}

template <int MODE> constexpr int WD_EPI_OPS = MODE == 0 ? 2 * 16 : 2 * 32;       // per wave and tile, not counting the scale / shift loads
static_assert(wd_younger(0) + WD_EPI_OPS<0> >= 63, "the clamped wait of a tile's first groups must not be weaker than the exact one");

// MODE: 0 no residual, 1 residual (both: buffer-descriptor epilogue)
template <int MODE>
__global__ __launch_bounds__(WD_THREADS, 1) void wino14_dual_kernel(const Wino14Params p
#if OM_WD_TRACE
                                                                    , unsigned long long* trace
#endif
) {
    __shared__ f32x4 smem[WD_VBUF + WD_RING * WD_UGRP + 1];       // ONE LDS object; last 16 B: ticket words
    int* const s_ticket = reinterpret_cast<int*>(smem + WD_VBUF + WD_RING * WD_UGRP);
    f32x4* const s_u = smem + WD_VBUF;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hp2 = p.H + 2;
    const float r_ct = 1.0f / (float)p.Ct, r_hp2 = 1.0f / (float)hp2;

    // ---------------------------------------------------------------- tile queue (conv_wino14.hip: eight per-XCD queues, M blocks
    // partitioned, N fastest; a workgroup whose queue is empty moves on to the next XCD's).  Drawn by ONE thread -- the first of
    // wave 3 -- TWO tiles ahead: the request goes out at the top of a tile and its answer is read at the tile's end, so the wave
    // never waits for it (an atomic's answer retires through the same in-order counter as every other request of the wave).
    int q_xcd = 0, q_hops = 0;
    auto queue_range = [&](int q, int& pm0, int& cnt) {
        const int m_tiles = p.total_tiles / p.n_tiles;
        pm0 = (int)((long long)m_tiles * q >> 3);
        const int pm1 = (int)((long long)m_tiles * (q + 1) >> 3);
        cnt = (pm1 - pm0) * p.n_tiles;
    };
    auto draw_request = [&]() { return atomicAdd(p.ticket + ((q_xcd + q_hops) & 7), 1); };
    auto draw_finish = [&](int v) {         // v: the answer of draw_request() on queue q_xcd + q_hops
        while (q_hops < 8) {
            int pm0, cnt;
            queue_range((q_xcd + q_hops) & 7, pm0, cnt);
            if (v < cnt) return pm0 * p.n_tiles + v;
            ++q_hops;
            if (q_hops < 8) v = draw_request();
        }
        return p.total_tiles;
    };
    if (tid == 192) {
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(q_xcd));
        q_xcd &= 7;
        s_ticket[0] = draw_finish(draw_request());
        s_ticket[1] = draw_finish(q_hops < 8 ? draw_request() : 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int tile = __builtin_amdgcn_readfirstlane(s_ticket[0]);
    int tile_next = __builtin_amdgcn_readfirstlane(s_ticket[1]);
    if (tile >= p.total_tiles) return;

    // ---------------------------------------------------------------- producer side: items (conv_wino14.hip's: a thread owns channel
    // quad tid & 3 of entries tid >> 2 and 64 + (tid >> 2), and a channel pair of entry 128 + (tid >> 3))
    const int ecount = (p.R + 2) * p.Ct;
    // Per-lane LDS addresses are recomputed at the top of every tile from an opaque copy of the thread id: as kernel-long values
    // they are live across the epilogue, whose working set then spills (and a reload in front of a matrix step waits for the
    // whole vector-memory queue).
    int xlds[3];               // LDS byte address of the item's hi halfs in plane 0 (lo: the 16-byte chunk two further on, XOR-swizzled)
    int dvo, boff_hi, boff_lo, aoff_hi[2][3], aoff_lo[2][3];
    auto setup_lane = [&]() {
        int t_ = tid;
        asm volatile("" : "+v"(t_));
        const int l_ = t_ & 63, w_ = __builtin_amdgcn_readfirstlane(t_ >> 6);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int e = k < 2 ? 64 * k + (t_ >> 2) : 128 + (t_ >> 3);
            const int q = k < 2 ? t_ & 3 : (t_ >> 1) & 3;
            const int ch = k < 2 ? 4 * q : 4 * q + 2 * (t_ & 1);
            const int sw = (e >> 2) & 3;
            xlds[k] = e * 64 + (((q >> 1) ^ sw) * 16) + (ch & 7) * 2;
        }
        // weight-group DMA: twelve 1-KiB pieces of 16 rows; wave w requests pieces 3 w .. 3 w + 2: one LDS base (M0) per group, the
        // instruction's immediate offset (0 / 1024 / 2048) steps through both the source and the LDS image
        const int drow = l_ >> 2, dcol = l_ & 3;
        const int row = 16 * (3 * w_) + drow;             // rows 16 i + drow of piece i: the swizzle (row >> 2) & 3 is the same in every piece
        dvo = row * 64 + ((dcol ^ ((row >> 2) & 3)) * 16);  // swizzle on the SOURCE chunk: the LDS image stays lane-linear
        const int fi_ = l_ & 31, fk_ = l_ >> 5;
        const int swB = (fi_ >> 2) & 3;
        boff_hi = (32 * (w_ & 1) + fi_) * 4 + (fk_ ^ swB);
        boff_lo = (32 * (w_ & 1) + fi_) * 4 + ((2 + fk_) ^ swB);
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int e = 32 * (2 * (w_ >> 1) + blk) + fi_ + ky * p.Ct;
                const int sw = (e >> 2) & 3;
                aoff_hi[blk][ky] = e * 4 + (fk_ ^ sw);
                aoff_lo[blk][ky] = e * 4 + ((2 + fk_) ^ sw);
            }
    };
    setup_lane();
    struct Items { int base[3]; unsigned nok[3]; };      // nok: bit 26 + x set = pixel x of the item does not exist
    // (everything recomputed from an opaque copy of the thread id: nothing of it is live across the chunk loop)
    auto setup_items = [&](int tile_id, Items& it) {
        int t_ = tid;
        asm volatile("" : "+v"(t_));
        Wino14Tile tt;
        wino14_decode(p, tile_id, tt);
        const bool tile_ok = tile_id < p.total_tiles;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int e = k < 2 ? 64 * k + (t_ >> 2) : 128 + (t_ >> 3);
            const int q = k < 2 ? t_ & 3 : (t_ >> 1) & 3;
            const int ch = k < 2 ? 4 * q : 4 * q + 2 * (t_ & 1);
            const int rr = wd_div(e, p.Ct, r_ct), t = e - rr * p.Ct;
            const int g = tt.g0 - 1 + rr;
            const int b = g >= 0 ? wd_div(g, hp2, r_hp2) : 0;
            const int y = g - b * hp2 - 1;
            const bool rowok = tile_ok && e < ecount && g >= 0 && g < p.gtot && y >= 0 && y < p.H;
            const int x0 = 4 * (tt.t0 + t) - 1;
            unsigned ok = 0;
#pragma unroll
            for (int x = 0; x < 6; ++x) ok |= (rowok && (unsigned)(x0 + x) < (unsigned)p.W ? 1u : 0u) << x;
            it.nok[k] = ~ok << 26;
            it.base[k] = (((b * p.H + y) * p.W + x0) * p.in_ps + ch) * 4;
        }
    };
    Items it;                   // the tile of the NEXT pixel requests: this tile's until chunk nch - 2 (whose requests are the next tile's first chunk)
    f32x4 xq[2][2][6];          // [register set = chunk parity][quad item][pixel]
    f32x2 xp[2][6];             // [register set][pixel] of the pair item
    // Pixel x of item k in chunk c: the lane's part of the offset is the item's first pixel, with bit 31 (beyond the descriptor's
    // range: the load returns zeros and requests nothing) for a pixel outside the image; pixel and chunk go into the scalar
    // offset operand, which the range check does not see.  Requests that go nowhere (c < 0) use a descriptor of size zero.
    // (the descriptor starts one pixel BEFORE the view -- nothing is read there: that pixel is "outside the image" wherever an
    // item's window begins at column -1 -- so the lane's part of the offset is never negative: the range check sees only that part)
    const auto rs_in1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in) - p.in_ps, 0, p.in_bytes + p.in_ps * 4, 0x00020000);
    const auto rs_none = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, 0, 0x00020000);
    auto item_voff = [&](int k, int x) { return (int)(((it.nok[k] << (5 - x)) & 0x80000000u) | (unsigned)(it.base[k] + p.in_ps * 4)); };
    auto load_quad_px = [&](auto setc, int k, int c, int x0, int x1) {
        constexpr int S = decltype(setc)::value;
        const bool live = c >= 0 && !(OM_WD_ABLATE & 4);
#pragma unroll
        for (int x = 0; x < 6; ++x)
            if (x >= x0 && x < x1)
                xq[S][k][x] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(live ? rs_in1 : rs_none, item_voff(k, x), x * p.in_ps * 4 + c * 64, 0));
    };
    auto load_quad = [&](auto setc, int k, int c) { load_quad_px(setc, k, c, 0, 6); };
    auto load_pair = [&](auto setc, int c, int x0, int x1) {
        constexpr int S = decltype(setc)::value;
        const bool live = c >= 0 && !(OM_WD_ABLATE & 4);
#pragma unroll
        for (int x = 0; x < 6; ++x)
            if (x >= x0 && x < x1)
                xp[S][x] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(live ? rs_in1 : rs_none, item_voff(2, x), x * p.in_ps * 4 + c * 64, 0));
    };
    // B^T along the row, one transform point at a time (conv_wino14.hip: point)
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
    // hi / lo split of a transformed value (conv_wino14.hip: split_store4 / split_store2: hi = RNE fp16, lo = RNE fp16 of the exact
    // remainder), in the pieces the group's schedule places between matrix instructions
    struct Split4 { u32x2 hb; f32x4 rem; };
    auto split4_hi = [&](const f32x4& v, Split4& s) {
        if constexpr (OM_WD_ABLATE & 2) { s.rem = v; s.hb = u32x2{0u, 0u}; return; }
        s.hb = __builtin_bit_cast(u32x2, __builtin_convertvector(v, f16x4));
        asm("v_fma_mix_f32 %0, %4, -1.0, %6 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %1, %4, -1.0, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %2, %5, -1.0, %8 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %3, %5, -1.0, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
            : "=&v"(s.rem[0]), "=&v"(s.rem[1]), "=&v"(s.rem[2]), "=&v"(s.rem[3])
            : "v"(s.hb[0]), "v"(s.hb[1]), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
    };
    auto split4_store = [&](const Split4& s, int k, int j) {
        if constexpr (OM_WD_ABLATE & 2) { if (s.rem[0] != 123.f) return; }
        char* dst = reinterpret_cast<char*>(smem) + xlds[k] + j * (WD_VPLANE * 16);
        const int lo_off = ((((xlds[k] >> 4) & 3) ^ 2) - ((xlds[k] >> 4) & 3)) * 16;
        *reinterpret_cast<u32x2*>(dst) = s.hb;
        *reinterpret_cast<u32x2*>(dst + lo_off) = __builtin_bit_cast(u32x2, __builtin_convertvector(s.rem, f16x4));
    };
    auto pair_plane = [&](auto setc, int j) {
        constexpr int S = decltype(setc)::value;
        const f32x2 v = point(xp[S], j);
        if constexpr (OM_WD_ABLATE & 2) { if (v[0] != 123.f) return; }
        const unsigned hb = __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
        f32x2 rem;
        asm("v_fma_mix_f32 %0, %2, -1.0, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %1, %2, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
            : "=&v"(rem[0]), "=&v"(rem[1]) : "v"(hb), "v"(v[0]), "v"(v[1]));
        char* dst = reinterpret_cast<char*>(smem) + xlds[2] + j * (WD_VPLANE * 16);
        const int lo_off = ((((xlds[2] >> 4) & 3) ^ 2) - ((xlds[2] >> 4) & 3)) * 16;
        *reinterpret_cast<unsigned*>(dst) = hb;
        *reinterpret_cast<unsigned*>(dst + lo_off) = __builtin_bit_cast(unsigned, __builtin_convertvector(rem, f16x2));
    };
    auto quad_plane = [&](auto setc, int k, int j) {       // (prologue: one piece)
        constexpr int S = decltype(setc)::value;
        Split4 s;
        split4_hi(point(xq[S][k], j), s);
        split4_store(s, k, j);
    };

    // ---------------------------------------------------------------- consumer side
    const int wm = wave >> 1, wn = wave & 1;
    const auto rs_u = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.u), 0, p.u_bytes, 0x00020000);
    constexpr int GRP_BYTES = WD_UGRP * 16;
    int G = 0;                                  // groups since the kernel's start: group G lives in ring slot G % RING
    Wino14Tile tl, tn;
    wino14_decode(p, tile, tl);
    wino14_decode(p, tile_next, tn);
    int ubase = tl.tile_n * p.nch * 6 * GRP_BYTES;
    int ubase_next = tile_next < p.total_tiles ? tn.tile_n * p.nch * 6 * GRP_BYTES : -1;
    // the weight group at chunk tc (may run into the next tile), plane position qq, into the slot of stream group Gs
    auto issue_group = [&](int tc, int qq, int Gs) {
        const bool here = tc < p.nch;
        const int base = here ? ubase : ubase_next;
        const int soff = (base < 0 ? 0 : base) + ((here ? tc : tc - p.nch) * 6 + w14_plane(qq)) * GRP_BYTES;
        const int oob = (base < 0 || (OM_WD_ABLATE & 8)) ? (int)0x80000000 : 0;      // no next tile: the DMA writes zeros and requests nothing
        const int slot = Gs & (WD_RING - 1);
        const lds_ptr_t dst = (lds_ptr_t)(s_u + slot * WD_UGRP + 3 * wave * 64);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_u, dst, 16, dvo | oob, soff, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_u, dst, 16, dvo | oob, soff, 1024, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_u, dst, 16, dvo | oob, soff, 2048, 0);
    };

    // fragments of a step: A hi / lo of the wave's two blocks, B hi / lo.  Three sets in rotation (step ky multiplies set ky and
    // reads set (ky + 1) % 3): no register copies between the steps.
    struct Frags { f32x4 a[4], b[2]; };
    Frags F0, F1, F2;
    auto read_a = [&](Frags& f, int j, int ky) {
        if constexpr (OM_WD_ABLATE & 32) { f.a[0] = f.a[1] = f.a[2] = f.a[3] = f32x4{(float)j, 1.f, (float)ky, (float)lane}; return; }
        f.a[0] = smem[j * WD_VPLANE + aoff_hi[0][ky]];
        f.a[1] = smem[j * WD_VPLANE + aoff_lo[0][ky]];
        f.a[2] = smem[j * WD_VPLANE + aoff_hi[1][ky]];
        f.a[3] = smem[j * WD_VPLANE + aoff_lo[1][ky]];
    };
    auto read_a1 = [&](Frags& f, int i, int j, int ky) {      // one of the four A fragments: hi / lo of block 0, hi / lo of block 1
        if constexpr (OM_WD_ABLATE & 32) { f.a[i] = f32x4{(float)j, 1.f, (float)ky, (float)lane}; return; }
        f.a[i] = smem[j * WD_VPLANE + ((i & 1) ? aoff_lo[i >> 1][ky] : aoff_hi[i >> 1][ky])];
    };
    // (bh / bl: the group's slot base + this lane's row, one address computation per group; the kernel row is an immediate offset)
    auto read_b = [&](Frags& f, int ky, const f32x4* bh, const f32x4* bl) {
        if constexpr (OM_WD_ABLATE & 32) { f.b[0] = f.b[1] = f32x4{(float)ky, 1.f, (float)ky, (float)lane}; return; }
        f.b[0] = bh[ky * (W14_BN * 4)];
        f.b[1] = bl[ky * (W14_BN * 4)];
    };
    auto zero_acc = [&]() {
        f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        asm volatile("" : "+v"(z));
        wd_acc_zero<0>(z); wd_acc_zero<1>(z); wd_acc_zero<2>(z); wd_acc_zero<3>(z); wd_acc_zero<4>(z); wd_acc_zero<5>(z);
        wd_acc_zero<6>(z); wd_acc_zero<7>(z); wd_acc_zero<8>(z); wd_acc_zero<9>(z); wd_acc_zero<10>(z); wd_acc_zero<11>(z);
    };

    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    // ---------------------------------------------------------------- prologue of the workgroup: both register sets, the first seven
    // weight groups, planes (0, 5) and (1, 2) of the first chunk
    setup_items(tile, it);
    load_quad(I0{}, 0, 0); load_quad(I0{}, 1, 0); load_pair(I0{}, 0, 0, 6);
#pragma unroll
    for (int g = 0; g < WD_RING - 1; ++g) issue_group(g / 6, g % 6, g);
#pragma unroll
    for (int k = 0; k < 2; ++k) { quad_plane(I0{}, k, 0); quad_plane(I0{}, k, 5); quad_plane(I0{}, k, 1); quad_plane(I0{}, k, 2); }
    pair_plane(I0{}, 0); pair_plane(I0{}, 5); pair_plane(I0{}, 1); pair_plane(I0{}, 2);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    float nonfinite = 0.f;

    while (true) {
        // the ticket after next: requested now, read at the tile's end
        int ticket_v = 0;
        if (tid == 192 && q_hops < 8) ticket_v = draw_request();
        // this tile's per-lane addresses, its second chunk's pixels (`it` is this tile's; the first chunk's were requested during the
        // tile before, or by the prologue), the first step's fragments, accumulators = 0
        setup_lane();
        load_quad(I1{}, 0, 1); load_quad(I1{}, 1, 1); load_pair(I1{}, 1, 0, 6);
        {
            const f32x4* const bh = s_u + (G & (WD_RING - 1)) * WD_UGRP + boff_hi, * const bl = s_u + (G & (WD_RING - 1)) * WD_UGRP + boff_lo;
            read_a(F0, 0, 0); read_b(F0, 0, bh, bl);
        }
        zero_acc();

        // One group of the chunk stream: plane position q of chunk c (register-set parity PAR = c & 1).
        //   matrix work    plane w14_plane(q) of chunk c: 3 kernel rows x 6 instructions (per accumulator the products in
        //                  conv_wino14.hip's order: hi x lo, lo x hi, hi x hi), fragments one step ahead (the last step reads the
        //                  next group's first fragments: its plane and its weights were published a barrier earlier)
        //   weight ring    group G + RING - 1 into the slot group G - 1 has left
        //   producer       groups 0-1: planes (3, 4) of chunk c (set PAR); 2-3: planes (0, 5), 4-5: planes (1, 2) of chunk c + 1 (set
        //                  PAR ^ 1); even groups quad item 0, odd groups quad item 1; the pair item's plane of the group
        //   requests       groups 2 and 3: the pixels of chunk c + 2 into set PAR (free since group 1)
        // The asm matrix instructions keep their order and every LDS / vector-memory operation stays on its side of them: the
        // placement below IS the schedule of the memory operations; the vector ALU work floats between them.
        auto group = [&](int c, auto qc, auto parc) {
            constexpr int q = decltype(qc)::value, PAR = decltype(parc)::value;
            constexpr int j = w14_plane(q), jn = w14_plane((q + 1) % 6);
            constexpr int ja = q < 2 ? 3 : q < 4 ? 0 : 1, jb = q < 2 ? 4 : q < 4 ? 5 : 2;
            constexpr int SET = q < 2 ? PAR : PAR ^ 1;
            constexpr int KQ = q & 1;
            // accumulator of the i-th matrix instruction of a step: block i & 1 of plane j (measurement 128: six different ones -- no
            // instruction waits for the result of the one before the last)
            auto acc_of = [](int i) constexpr { return (OM_WD_ABLATE & 128) ? (i < 3 ? (j + i) % 6 : 6 + (j + i) % 6) : 6 * (i & 1) + j; };
            using SetC = std::integral_constant<int, SET>;
            using ParC = std::integral_constant<int, PAR>;
            const int slot = G & (WD_RING - 1), slot1 = (G + 1) & (WD_RING - 1);
            const f32x4* const bh = s_u + slot * WD_UGRP + boff_hi, * const bl = s_u + slot * WD_UGRP + boff_lo;
            const f32x4* const bh1 = s_u + slot1 * WD_UGRP + boff_hi, * const bl1 = s_u + slot1 * WD_UGRP + boff_lo;
            auto h = [](const f32x4& v) { return __builtin_bit_cast(f16x8, v); };
            Split4 sa, sb;
#if OM_WD_TRACE
            unsigned long long t0, t1, t2, ts1, ts2;
            WD_STAMP(t0);
#endif
            asm volatile("" ::: "memory");
            issue_group(c + (q + WD_RING - 1) / 6, (q + WD_RING - 1) % 6, G + WD_RING - 1);
            asm volatile("" ::: "memory");
            // Every matrix instruction is followed by ONE gap of a few other instructions and a scheduling fence: alone on its SIMD,
            // the wave overlaps its matrix instructions only with what it issues in their shadow (about five instructions per 32
            // cycles); left to the scheduler, the producer's work gathers between the steps, where no matrix instruction is in flight.
#define WD_GAP __builtin_amdgcn_sched_barrier(0)
            // ---- kernel row 0 (set F0; reads F1 = kernel row 1); producer: plane ja of the quad item
            wd_mfma_first<acc_of(0)>(h(F0.b[0]), h(F0.a[1]));
            read_a1(F1, 0, j, 1); read_a1(F1, 1, j, 1); WD_GAP;
            wd_mfma<acc_of(1)>(h(F0.b[0]), h(F0.a[3]));
            read_a1(F1, 2, j, 1); read_a1(F1, 3, j, 1); WD_GAP;
            wd_mfma<acc_of(2)>(h(F0.b[1]), h(F0.a[0]));
            read_b(F1, 1, bh, bl); WD_GAP;
            wd_mfma<acc_of(3)>(h(F0.b[1]), h(F0.a[2]));
            const f32x4 va = point(xq[SET][KQ], ja); WD_GAP;
            wd_mfma<acc_of(4)>(h(F0.b[0]), h(F0.a[0]));
            split4_hi(va, sa); WD_GAP;
            wd_mfma<acc_of(5)>(h(F0.b[0]), h(F0.a[2]));
            split4_store(sa, KQ, ja); WD_GAP;
            // ---- kernel row 1 (set F1; reads F2 = kernel row 2); plane jb of the quad item
            wd_mfma_first<acc_of(0)>(h(F1.b[0]), h(F1.a[1]));
            read_a1(F2, 0, j, 2); read_a1(F2, 1, j, 2); WD_GAP;
            wd_mfma<acc_of(1)>(h(F1.b[0]), h(F1.a[3]));
            read_a1(F2, 2, j, 2); read_a1(F2, 3, j, 2); WD_GAP;
            wd_mfma<acc_of(2)>(h(F1.b[1]), h(F1.a[0]));
            read_b(F2, 2, bh, bl); WD_GAP;
            wd_mfma<acc_of(3)>(h(F1.b[1]), h(F1.a[2]));
            const f32x4 vb = point(xq[SET][KQ], jb); WD_GAP;
            wd_mfma<acc_of(4)>(h(F1.b[0]), h(F1.a[0]));
            split4_hi(vb, sb); WD_GAP;
            wd_mfma<acc_of(5)>(h(F1.b[0]), h(F1.a[2]));
            split4_store(sb, KQ, jb); WD_GAP;
            // ---- kernel row 2 (set F2; reads F0 = the next group's kernel row 0); the pair item's plane, the chunk's requests
            // (chunk c + 2; behind chunk nch - 2 the next tile's first chunk -- `it` is the next tile's by then; behind the last chunk
            // NOTHING, requests that go nowhere (same counts): the next tile's second chunk is requested at the top of that tile,
            // its register set is the epilogue's working space meanwhile)
            [[maybe_unused]] const int tc = c + 2 < p.nch ? c + 2 : c + 2 == p.nch ? 0 : -1;
            constexpr bool LOADS = q == 2 || q == 3;
            wd_mfma_first<acc_of(0)>(h(F2.b[0]), h(F2.a[1]));
            read_a1(F0, 0, jn, 0); read_a1(F0, 1, jn, 0); WD_GAP;
            wd_mfma<acc_of(1)>(h(F2.b[0]), h(F2.a[3]));
            read_a1(F0, 2, jn, 0); read_a1(F0, 3, jn, 0); WD_GAP;
            wd_mfma<acc_of(2)>(h(F2.b[1]), h(F2.a[0]));
            read_b(F0, 0, bh1, bl1); WD_GAP;
            wd_mfma<acc_of(3)>(h(F2.b[1]), h(F2.a[2]));
            pair_plane(SetC{}, KQ ? jb : ja);
            if constexpr (LOADS) load_quad_px(ParC{}, KQ, tc, 0, 2);
            WD_GAP;
            wd_mfma<acc_of(4)>(h(F2.b[0]), h(F2.a[0]));
            if constexpr (LOADS) load_quad_px(ParC{}, KQ, tc, 2, 4);
            WD_GAP;
            wd_mfma<acc_of(5)>(h(F2.b[0]), h(F2.a[2]));
            if constexpr (LOADS) { load_quad_px(ParC{}, KQ, tc, 4, 6); load_pair(ParC{}, tc, q == 2 ? 0 : 3, q == 2 ? 3 : 6); }
            WD_GAP;
#undef WD_GAP
            // my pieces of weight group G + 2 have landed; my LDS stores are done; the next group's first fragments are here
            // (a tile's first RING - 3 groups: the epilogue's stores are among the younger operations, the exact count exceeds the
            // counter's six bits -- the largest count is still a stronger wait than necessary)
            static_assert(wd_younger(q) <= 63, "vmcnt has six bits");
#if OM_WD_TRACE
            WD_STAMP(t1);
#endif
            if constexpr (PAR == 0 && q < WD_RING - 3)
                asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_waitcnt vmcnt(%1)\n1:\n\ts_waitcnt vmcnt(63) lgkmcnt(0)" ::"s"(c), "n"(wd_younger(q)) : "memory", "scc");
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(wd_younger(q)) : "memory");
#if OM_WD_TRACE
            WD_STAMP(t2);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(t0), "+s"(t1), "+s"(t2), "+s"(ts1), "+s"(ts2)::"memory");
            if (blockIdx.x < 8 && G < 512 && lane == 0) {
                unsigned long long* t = trace + ((blockIdx.x * 4 + wave) * 512 + G) * 4;
                t[0] = t0; t[1] = t1; t[2] = t2;
                unsigned long long* u = trace + 8 * 4 * 512 * 4 + ((blockIdx.x * 4 + wave) * 512 + G) * 2;
                u[0] = ts1; u[1] = ts2;
            }
#endif
            if constexpr (!(OM_WD_ABLATE & 64)) __builtin_amdgcn_s_barrier();
            ++G;
        };
        auto chunk = [&](int c, auto parc) {
            group(c, std::integral_constant<int, 0>{}, parc);
            group(c, std::integral_constant<int, 1>{}, parc);
            group(c, std::integral_constant<int, 2>{}, parc);
            group(c, std::integral_constant<int, 3>{}, parc);
            group(c, std::integral_constant<int, 4>{}, parc);
            group(c, std::integral_constant<int, 5>{}, parc);
        };
#pragma unroll 1
        for (int c = 0; c < p.nch; c += 2) {
            if (c == p.nch - 2) setup_items(tile_next, it);
            chunk(c, I0{});
            chunk(c + 1, I1{});
        }

        // ------------------------------------------------------------ the tile's end: planes (3, 4) are dead (20 KiB of staging).
        // The last matrix instruction's result needs 12 wait states before a copy reads it.
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        f32x4* sT = smem + 3 * WD_VPLANE + wave * 256;
        if constexpr (!(OM_WD_ABLATE & 1)) {
            __builtin_amdgcn_sched_barrier(0);
            wd_epilogue<MODE, 0>(p, tl, sT, 2 * wm, wn, lane, nonfinite);
            __builtin_amdgcn_sched_barrier(0);
            wd_epilogue<MODE, 1>(p, tl, sT, 2 * wm + 1, wn, lane, nonfinite);
            __builtin_amdgcn_sched_barrier(0);
        }
#if OM_WD_TRACE
        {
            unsigned long long te;
            WD_STAMP(te);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(te)::"memory");
            if (blockIdx.x < 8 && G - 1 < 512 && lane == 0) trace[((blockIdx.x * 4 + wave) * 512 + G - 1) * 4 + 3] = te;
        }
#endif
        if (tid == 192) s_ticket[2] = draw_finish(ticket_v);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();               // every wave has left the staging area; the ticket after next is published
        tile = tile_next;
        if (tile >= p.total_tiles) break;
        tile_next = __builtin_amdgcn_readfirstlane(s_ticket[2]);
        tl = tn;
        wino14_decode(p, tile_next, tn);
        ubase = ubase_next;
        ubase_next = tile_next < p.total_tiles ? tn.tile_n * p.nch * 6 * GRP_BYTES : -1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the ring's last requests (zeros for tiles that do not exist)
    if (p.status && nonfinite != nonfinite) atomicOr(p.status, OM_STATUS_SPLIT_RANGE);
}

bool wino14_dual_supported(const Wino14Params& p) {
    return p.fast_io && p.nch >= 2 && p.nch % 2 == 0 && (p.R + 2) * p.Ct <= W14_EMAX_ALL;
}

int launch_wino14_dual(const Wino14Params& p, bool has_res, hipStream_t stream) {
    const long long grid = p.total_tiles < 256 ? p.total_tiles : 256;        // one workgroup per CU (156 KiB of LDS, 512 registers per lane)
#if OM_WD_TRACE
    OM_REQUIRE(g_wd_trace, OM_EINVAL, "wino14 dual trace build: om_debug_wd_trace() first");
    if (has_res) hipLaunchKernelGGL(wino14_dual_kernel<1>, dim3((unsigned)grid), dim3(WD_THREADS), 0, stream, p, g_wd_trace);
    else hipLaunchKernelGGL(wino14_dual_kernel<0>, dim3((unsigned)grid), dim3(WD_THREADS), 0, stream, p, g_wd_trace);
#else
    if (has_res) hipLaunchKernelGGL(wino14_dual_kernel<1>, dim3((unsigned)grid), dim3(WD_THREADS), 0, stream, p);
    else hipLaunchKernelGGL(wino14_dual_kernel<0>, dim3((unsigned)grid), dim3(WD_THREADS), 0, stream, p);
#endif
    OM_CHECK_HIP(hipGetLastError());
    return OM_OK;
}

}  // namespace om
