// Pieces shared by the two fused F(4,3)-along-the-rows kernels with split operands: conv_wino14.hip (eight consumer waves + four
// producer waves, round 3) and conv_wino14d.hip (four dual-role waves, one per SIMD, round 5).  Same layer, same packed weights,
// same epilogue, the same sequence of fp32 operations per output -- the two kernels are bit-identical.
#pragma once
#include <cstdlib>
#include <type_traits>

#include "conv_f16_common.h"

namespace om {

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4v __attribute__((__vector_size__(4 * sizeof(unsigned))));


#ifndef OM_W14_ABLATE
#define OM_W14_ABLATE 0        // measurement builds only (wrong numerics): 2 no fragment reads, 4 no per-group barrier, 8 no weight
#endif                         // DMA, 16 no input loads / transform, 32 no epilogue stores, 1024 no matrix instructions, 262144 the
                               // input read at a channel-chunk-major layout's addresses (round 6, profiles/r06_experiments.md 1)
#ifndef OM_W14_TRACE
#define OM_W14_TRACE 0         // measurement builds only: s_memtime stamps of one tile's groups (tools/wino14_trace.py)
#endif
#if (OM_W14_ABLATE || OM_W14_TRACE) && !defined(OM_MEASUREMENT_BUILD)
#error "measurement switches (wrong numerics / trace stores) are only for ab/ variants: build them with tools/build_variant.sh, which defines OM_MEASUREMENT_BUILD and never writes orienmask_amd/lib/"
#endif
constexpr int W14_BM = 128, W14_BN = 64;
constexpr int W14_EMAX_ALL = 144;      // LDS entries per plane both kernels allow: (R + 2) * Ct <= 144 (wino14_geometry)
#ifndef W14_SPREAD
#define W14_SPREAD 1           // plane order of a chunk's six groups: 1 = 0, 5, 1, 2, 3, 4 (conv_wino14.hip: the producers' schedule)
#endif

struct Wino14Params {
    const float* in;        // NHWC fp32 view
    const _Float16* u;      // packed [n_tiles][cin / 16][6 j][3 ky][64 rows][32 halfs]  (pack.py: winograd14_weights_split)
    const float* scale;     // scale * 2^-e
    const float* shift;
    const float* res;
    float* out;
    int* ticket;
    int* status;
    int B, H, W, in_ps, in_bytes;
    int cout, out_ps, res_ps, leaky, fast_io, out_bytes, res_bytes;
    int R, Ct, ncb, gtot;   // block = R padded rows x Ct tile columns; ncb column blocks per row block; gtot = B * (H + 2)
    int n_tiles, total_tiles, nch;      // nch = cin / 16
    int u_bytes;
#if OM_W14_TRACE
    unsigned long long* trace;
#endif
};

#if OM_W14_TRACE
// Time stamps without disturbing the LDS queue: s_memtime is issued where the event happens and its result is only read behind
// a wait the kernel has anyway.  [block][wave 0 / 4 / 8][group 0..63][4 stamps]
static unsigned long long* g_w14_trace = nullptr;
extern "C" void om_debug_w14_trace(void* buf) { g_w14_trace = static_cast<unsigned long long*>(buf); }
#define W14_STAMP(x) asm volatile("s_memtime %0" : "=s"(x)::"memory")
#define W14_SETTLE(a, b, c, d) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b), "+s"(c), "+s"(d)::"memory")
#define W14_SETTLE2(a, b) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b)::"memory")
__device__ __forceinline__ void w14_trace_put(const Wino14Params& p, int slot, int g, unsigned long long a, unsigned long long b,
                                              unsigned long long c, unsigned long long d) {
    if (blockIdx.x < 8 && g < 64 && (threadIdx.x & 63) == 0) {
        unsigned long long* t = p.trace + ((blockIdx.x * 12 + slot) * 64 + g) * 4;
        t[0] = a; t[1] = b; t[2] = c; t[3] = d;
    }
}
#endif

// Order in which the six planes (transform points) of a chunk are multiplied: group g of a chunk works on plane w14_plane(g) of V
// and of U.  Pairs (1, 2), (3, 4), (0, 5): what the producers make in one group from shared differences of the same pixels.
__host__ __device__ constexpr int w14_plane(int g) {
    return W14_SPREAD ? (g == 0 ? 0 : g == 1 ? 5 : g - 1) : (g == 0 ? 1 : g == 1 ? 2 : g == 2 ? 3 : g == 3 ? 4 : g == 4 ? 0 : 5);
}

struct Wino14Tile {
    int g0, t0, n0, tile_n;
};

__device__ __forceinline__ void wino14_decode(const Wino14Params& p, int tile, Wino14Tile& t) {
    // N fastest: the workgroups that transform the same input block run at the same time (its pixels come from L2)
    t.tile_n = tile % p.n_tiles;
    const int tm = tile / p.n_tiles;
    const int cb = tm % p.ncb, rb = tm / p.ncb;
    t.g0 = rb * p.R; t.t0 = cb * p.Ct; t.n0 = t.tile_n * W14_BN;
}

// Epilogue of a consumer wave, from its six plane accumulators, with no workgroup barrier: the inverse transform position by
// position (A^T rows (1,1,1,1,1,0), (0,1,-1,2,-2,0), (0,1,1,4,4,0), (0,1,-1,8,-8,1)), a transpose of the wave's 32 entries x 32
// channels through 4 KiB of LDS of its own, then scale / shift, LeakyReLU, residual and 16-byte stores.  In the accumulators a
// lane holds ONE entry (lane & 31) and four runs of four channels (8 q + 4 (lane >> 5)): stored from there, an instruction writes
// 32-byte pieces (measured: 20 000 cycles per tile).  After the transpose lane L holds channels 4 (L & 7).. of entries 8 r + (L >> 3),
// r = 0..3 -- eight lanes complete a 128-byte line -- and needs one scale / shift quad for the whole tile.
// (The first version staged four fp32 C tiles of the whole workgroup in LDS: two barriers and a store phase of all 768 threads.)
// sT: this wave's 256 f32x4 of the V buffer that the next tile does not write before its prologue barrier (buffer 1).
//
// FAST (16-byte aligned views, cout a multiple of 4, views below 2 GiB): NOTHING in it waits for a store.  Loads and stores
// retire through one in-order counter (vmcnt), so waiting for any load -- a residual quad, a spilled register -- behind a store
// waits for that store's round trip: the first version of this function did so sixteen times per tile, 24 000 of a cin = 128
// tile's 88 000 cycles (profiles/r03_w14_trace_tile_phases.txt).  Here every request is unconditional (an out-of-range offset
// for entries outside the image: the buffer descriptor drops the store and returns zeros for the load), so the compiler's
// counts are exact; the residual of position px + 1 is requested BEFORE the stores of position px; no spills, no calls.
// before_requests(): called once, after the set-up (whose register traffic may wait for everything outstanding) and before the
// first store: the caller's requests for the next tile's first weight groups go there, so that exactly W14_EPI_OPS<MODE>
// vector-memory operations follow them.
template <int MODE> constexpr int W14_EPI_OPS = MODE == 0 ? 16 : MODE == 1 ? 28 : 0;
template <int MODE, typename F>
__device__ __forceinline__ void wino14_epilogue(const Wino14Params& p, const f32x16 (&acc)[6], const Wino14Tile& tl, f32x4* sT, int wm,
                                                int wn, int lane, F before_requests) {
    // everything below is recomputed per tile from an opaque copy of the lane id: hoisted out of the tile loop, the per-lane
    // entry coordinates would be live across the main loop, i.e. spilled, and reloaded here one round trip at a time
    asm volatile("" : "+v"(lane));
    const int fi = lane & 31, fk = lane >> 5;
    const int hp2 = p.H + 2;
    const int c8 = lane & 7;
    const int nb = tl.n0 + 32 * wn + 4 * c8;
    const int nvalid = p.cout - nb;
    const f32x4 sc = *reinterpret_cast<const f32x4*>(p.scale + nb);      // padded to cout_pad
    const f32x4 sh = *reinterpret_cast<const f32x4*>(p.shift + nb);
    // the four entries this lane stores: pixel index of the entry's first pixel, and its column (>= W: nothing to store)
    int pix0[4], oxe[4];
#pragma unroll
    for (int rd = 0; rd < 4; ++rd) {
        const int ml = 32 * wm + 8 * rd + (lane >> 3);
        const int r = ml / p.Ct, t = ml - r * p.Ct;
        const int gg = tl.g0 + r;
        const int b = gg / hp2;
        const int y = gg - b * hp2 - 1;
        const bool rowok = r < p.R && gg < p.gtot && y >= 0 && y < p.H && nvalid > 0;
        oxe[rd] = rowok ? 4 * (tl.t0 + t) : p.W;
        pix0[rd] = (b * p.H + y) * p.W + 4 * (tl.t0 + t);
    }
    float nonfinite = 0.f;          // range guard of the split representation (conv_igemm_split.hip: split_epilogue)
    auto position = [&](int px, int gq, int k) {
        const int i = 4 * gq + k;
        if (px == 0) return acc[0][i] + acc[1][i] + acc[2][i] + acc[3][i] + acc[4][i];
        if (px == 1) return (acc[1][i] - acc[2][i]) + 2.f * (acc[3][i] - acc[4][i]);
        if (px == 2) return (acc[1][i] + acc[2][i]) + 4.f * (acc[3][i] + acc[4][i]);
        return (acc[1][i] - acc[2][i]) + 8.f * (acc[3][i] - acc[4][i]) + acc[5][i];
    };
    auto transpose_in = [&](int px) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const f32x4 v = {position(px, gq, 0), position(px, gq, 1), position(px, gq, 2), position(px, gq, 3)};
            sT[fi * 8 + ((2 * gq + fk) ^ (fi & 7))] = v;        // row = entry, 16-byte chunk = channel quad, XOR-swizzled
        }
    };
    // the wave's own LDS operations complete in order: no wait between its writes and its reads, nor before the next
    // position's writes
    auto transpose_out = [&](int rd) {
        const int e = 8 * rd + (lane >> 3);
        return sT[e * 8 + (c8 ^ (e & 7))];
    };
    // entries beyond the block's R rows multiplied whatever the LDS held: they are neither stored nor range-checked
    auto activate = [&](f32x4 v, bool ok) {
        float nf = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float tv = fmaf(v[k], sc[k], sh[k]);
            nf = fmaf(tv, 0.f, nf);
            v[k] = p.leaky ? fmaxf(tv, tv * 0.1f) : tv;         // LeakyReLU(0.1): the larger of x and 0.1 x
        }
        nonfinite += ok ? nf : 0.f;
        return v;
    };
    if constexpr (MODE < 2) {
        const auto rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
        const auto rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res), 0, MODE == 1 ? p.res_bytes : 0, 0x00020000);
        auto offset = [&](int rd, int px, int ps) { return oxe[rd] + px < p.W ? ((pix0[rd] + px) * ps + nb) * 4 : (int)0x80000000; };
        f32x4 rc[4];
        if constexpr (MODE == 1) {
#pragma unroll
            for (int rd = 0; rd < 4; ++rd) rc[rd] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, offset(rd, 0, p.res_ps), 0, 0));
        }
        before_requests();
        // Positions in the order 0, 3, 1, 2: plane 0 is dead after the first, plane 5 after the second.  Four explicit copies
        // separated by compiler barriers: as one unrolled loop the scheduler hoists every residual request to the top (and
        // spills them), as a rolled loop all six planes stay live and the per-entry offsets spill instead -- and a spilled
        // register reloaded between two stores waits for the older store's round trip.
        auto one_position = [&](auto pxc, auto nextc) {
            constexpr int px = decltype(pxc)::value, nx = decltype(nextc)::value;
            transpose_in(px);
            f32x4 v[4];
#pragma unroll
            for (int rd = 0; rd < 4; ++rd) {
                v[rd] = activate(transpose_out(rd), oxe[rd] + px < p.W);
                if constexpr (MODE == 1) v[rd] += rc[rd];
            }
            if constexpr (MODE == 1 && nx >= 0) {       // the next position's residual BEFORE this position's stores
#pragma unroll
                for (int rd = 0; rd < 4; ++rd)
                    rc[rd] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, offset(rd, nx, p.res_ps), 0, 0));
            }
#pragma unroll
            for (int rd = 0; rd < 4; ++rd) {
                if ((OM_W14_ABLATE & 32) && v[rd][0] != 123.f) continue;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v[rd]), rs_out, offset(rd, px, p.out_ps), 0, 0);
            }
            asm volatile("" ::: "memory");
        };
        using std::integral_constant;
        one_position(integral_constant<int, 0>{}, integral_constant<int, 3>{});
        one_position(integral_constant<int, 3>{}, integral_constant<int, 1>{});
        one_position(integral_constant<int, 1>{}, integral_constant<int, 2>{});
        one_position(integral_constant<int, 2>{}, integral_constant<int, -1>{});
    } else {
        before_requests();
        // any view, any cout: one element at a time
#pragma unroll 1
        for (int px = 0; px < 4; ++px) {
            if (px == 0) transpose_in(0);
            else if (px == 1) transpose_in(1);
            else if (px == 2) transpose_in(2);
            else transpose_in(3);
#pragma unroll 1
            for (int rd = 0; rd < 4; ++rd) {
                const int ox = rd == 0 ? oxe[0] : rd == 1 ? oxe[1] : rd == 2 ? oxe[2] : oxe[3];
                const int px0 = rd == 0 ? pix0[0] : rd == 1 ? pix0[1] : rd == 2 ? pix0[2] : pix0[3];
                const bool ok = ox + px < p.W;
                const f32x4 v = activate(transpose_out(rd), ok);
                if (!ok) continue;
                float* o = p.out + (long long)(px0 + px) * p.out_ps + nb;
                const float* rp = p.res ? p.res + (long long)(px0 + px) * p.res_ps + nb : nullptr;
                for (int k = 0; k < 4 && k < nvalid; ++k) o[k] = rp ? v[k] + rp[k] : v[k];
            }
        }
    }
    if (p.status && nonfinite != nonfinite) atomicOr(p.status, OM_STATUS_SPLIT_RANGE);
}

// conv_wino14d.hip
bool wino14_dual_supported(const Wino14Params& p);
int launch_wino14_dual(const Wino14Params& p, bool has_res, hipStream_t stream);
int wino14_variant();               // 0: the twelve-wave kernel everywhere (default); 1: the dual-role kernel where it applies (OM_W14_VARIANT)
void wino14_set_variant(int v);

}  // namespace om
