// The first two layers of DarkNet-53 as ONE kernel in split-operand mode (precision mode 1):
//   backbone.conv1   = conv_bn_leaky(3, 32, 3, padding=1)              (/root/reference/model/backbone/darknet.py:20)
//   backbone.conv2.0 = conv_bn_leaky(32, 64, 3, stride=2, padding=1)   (darknet.py:21-22, model/base.py:104-137)
// Separately (conv_stem.hip, conv_igemm_split.hip) the 32-channel full-resolution activation between them is written and
// read through HBM: 1.2 GB each way at bs = 32, 544^2 -- 0.37 + 0.55 ms for two layers whose arithmetic needs 0.2 ms
// (VERDICT round 2, item 5).  Here a workgroup owns 8 x 16 outputs of conv2.0 x all 64 channels and
//   1. stages the 3 x 19 x 35 image patch under them in LDS (zero padded: out-of-range offsets of a buffer descriptor),
//   2. computes conv1's 17 x 33 x 32 activations of that patch -- on the matrix pipe too: blocks of 32 activations x 32 channels
//      x K = 27 (padded to 32) with split operands like every other layer of this mode (image window and weights as hi/lo fp16
//      pairs, three matrix instructions per 16 of K, fp32 accumulation) -- BatchNorm, LeakyReLU, splits them into hi/lo fp16
//      (conv_igemm_split.hip: split8) and writes them to LDS in the matrix instruction's operand order (zeros outside the image:
//      conv2.0's padding),
//   3. multiplies: 9 taps x 2 chunks of 16 channels, three v_mfma_f32_32x32x16_f16 per step in conv_igemm_split_kernel's order,
//      the stride-2 gather being LDS addressing; conv2.0's packed hi/lo weights (73.7 KB for 64 output channels) stay in LDS
//      for the kernel's life,
//   4. scale / shift, LeakyReLU, 16-byte stores through a per-wave LDS transpose (conv_wino14.hip's epilogue: nothing waits for a
//      store).
// LDS: weights 73 728 + activations 17 x 36 x 128 = 78 336 + patch 7 980 bytes: one 512-thread workgroup per CU, tiles by a
// static stride (every tile costs the same).
#if defined(OM_S2_TRACE) && !defined(OM_MEASUREMENT_BUILD)
#error "OM_S2_TRACE writes time stamps through the status word and disables the range guard: only for ab/ variants (tools/build_variant.sh defines OM_MEASUREMENT_BUILD and never writes orienmask_amd/lib/)"
#endif
#include "om_common.h"

namespace om {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4s __attribute__((__vector_size__(4 * sizeof(unsigned))));

constexpr int S2_TY = 8, S2_TX = 16;                    // conv2.0 outputs per tile
constexpr int S2_SR = 2 * S2_TY + 1, S2_SC = 2 * S2_TX + 1;      // conv1 activations under them: 17 x 33
constexpr int S2_SLOTS = 36;                            // LDS slots (128 B: a pixel's 32 channels as hi/lo) per activation row
constexpr int S2_PR = S2_SR + 2, S2_PC = S2_SC + 2;     // image patch: 19 x 35
constexpr int S2_W_BYTES = 18 * 64 * 64;                // [step = tap * 2 + chunk][row][64 B]
constexpr int S2_S_BYTES = S2_SR * S2_SLOTS * 128;
constexpr int S2_P_FLOATS = 3 * S2_PR * S2_PC;
constexpr int S2_THREADS = 512;

struct Stem2Params {
    const float* img;       // [B,3,H,W]
    const float* w1;        // [32][27]
    const float* sc1;
    const float* sh1;
    const _Float16* w2;     // conv_weights_split rows of conv2.0: [64][9][2][4][8] halfs
    const float* sc2;       // scale * 2^-e
    const float* sh2;
    float* out;             // NHWC [B, H/2, W/2, out_ps]
    int* status;
    int B, H, W, Ho, Wo, out_ps, leaky2;
    int tiles_x, tiles_y, total_tiles;
    // optional THIRD layer (round 5): the 1x1 convolution 64 -> 32 behind conv2.0 (backbone.conv2.1.conv.0, darknet.py:9-13) on the
    // tile's outputs while they are still in the workgroup -- conv2.0's activation is written once (the residual needs it) and not
    // read back: w3 = conv_weights_split rows [32][4][4][8] halfs, sc3 = scale * 2^-e, out3 NHWC [B, H/2, W/2, out3_ps]; nullptr: off
    const _Float16* w3;
    const float* sc3;
    const float* sh3;
    float* out3;
    int out3_ps, leaky3;
};

// rows r and r + 1 differ in bit 3 of the chunk, rows r and r + 2 in its low bits: a bijection of r & 15
__device__ __forceinline__ int s2_row_swizzle(int r) { return ((r & 1) << 3) | ((r >> 1) & 7); }

template <bool THIRD>
__global__ __launch_bounds__(S2_THREADS, 2) void conv_stem2_split_kernel(const Stem2Params p) {
    __shared__ f32x4 smem[(S2_W_BYTES + S2_S_BYTES + S2_P_FLOATS * 4 + 15) / 16];
    char* const sW = reinterpret_cast<char*>(smem);
    char* const sS = sW + S2_W_BYTES;
    float* const sP = reinterpret_cast<float*>(sS + S2_S_BYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- conv2.0's weights: global [row][step][chunk] -> LDS [step][row][chunk ^ ((row >> 2) & 3)], once
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(p.w2);
#pragma unroll
        for (int i = 0; i < S2_W_BYTES / 16 / S2_THREADS; ++i) {
            const int idx = tid + i * S2_THREADS;
            const int n = idx / 72, rem = idx - n * 72;
            const int step = rem >> 2, k = rem & 3;
            *reinterpret_cast<f32x4*>(sW + step * 4096 + n * 64 + ((k ^ ((n >> 2) & 3)) * 16)) = src[idx];
        }
    }
    // ---- conv1's role: ALSO on the matrix pipe.  (On the vector ALUs -- a thread per activation and channel quad, 27 x 4 fused
    // multiply-adds each, as conv_stem_kernel does -- this phase took 11 100 of a tile's 18 100 cycles: the 114 instructions
    // around the 56 packed multiply-adds of an activation cost as much as they do.)  A wave multiplies blocks of 32 activations
    // x 32 channels x K = 27 (padded to 32: two steps of 16): the weights as hi/lo fp16 in registers for the kernel's life, the
    // 3 x 3 x 3 image window of an activation gathered from the LDS patch, split, three matrix instructions per step.
    const int fi = lane & 31, fk = lane >> 5;
    f16x8 w1h[2], w1l[2];
    int koff[2][8];                 // patch offset of contraction index k = 16 s + 8 fk + i relative to the window's origin
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = 16 * s2 + 8 * fk + i;             // k = (kh * 3 + kw) * 3 + ci, the order of w1's rows
            const int kk = k < 27 ? k : 0;
            const int t = kk / 3, ci = kk - 3 * t;
            const int kh = t / 3, kw = t - 3 * kh;
            koff[s2][i] = (ci * S2_PR + kh) * S2_PC + kw;
            const float w = k < 27 ? p.w1[fi * 27 + kk] : 0.f;
            const _Float16 h = (_Float16)w;
            w1h[s2][i] = h;
            w1l[s2][i] = (_Float16)(w - (float)h);
        }
    // (the 16 channels a lane holds of an activation after the products: 8 (r >> 2) + 4 fk + (r & 3); their scale / shift are
    // requested per block of activations, in front of its window gather: 32 registers less across the kernel's life)
    // ---- conv2.0's role: wave = 32 outputs (two tile rows) x 32 channels
    const int wm = wave >> 1, wn = wave & 1;
    const int m = 32 * wm + fi;
    const int oy = m >> 4, ox = m & 15;
    const int nrow = 32 * wn + fi;
    const int boff_hi = nrow * 64 + ((fk ^ ((nrow >> 2) & 3)) * 16);
    const int boff_lo = nrow * 64 + (((2 + fk) ^ ((nrow >> 2) & 3)) * 16);
    // epilogue role (conv_wino14.hip): lane holds channels 4 (lane & 7).. of outputs 8 rd + (lane >> 3) of the wave's 32
    const int c8 = lane & 7;
    const int nb = 32 * wn + 4 * c8;
    const f32x4 sc2 = *reinterpret_cast<const f32x4*>(p.sc2 + nb);
    const f32x4 sh2 = *reinterpret_cast<const f32x4*>(p.sh2 + nb);
    // the pre-loop loads have landed as far as the compiler's wait bookkeeping goes (conv_stem.hip)
    asm volatile("" ::"v"(w1h[0]), "v"(w1h[1]), "v"(w1l[0]), "v"(w1l[1]), "v"(sc2), "v"(sh2));
    float nonfinite = 0.f;

    // the image patch of a tile travels through registers: requested before the PREVIOUS tile's matrix phase, so that its round
    // trip to HBM runs under that phase -- and the requests are older than that tile's output stores (one in-order counter: a
    // wait for a load behind a store would wait for the store's round trip too)
    // Round 6, with the third layer: waves 4-7 are idle while waves 0-3 multiply it, so THEY own the patch (twice the registers each,
    // nothing requested by waves 0-3: their offsets are out of range) and store the next tile's into LDS during that phase -- the
    // patch area is dead from the end of conv1 on -- instead of all eight waves at the top of the next tile (1 100 of a tile's
    // 17 500 cycles, tools/stem2_trace.py).
    constexpr int S2_PATCH_THREADS = THIRD ? S2_THREADS / 2 : S2_THREADS;
    constexpr int NSTAGE = (S2_P_FLOATS + S2_PATCH_THREADS - 1) / S2_PATCH_THREADS;
    const int ptid = THIRD ? tid - S2_THREADS / 2 : tid;        // < 0: not a patch thread
    float stage[NSTAGE];
    auto request_patch = [&](int tile) {
        const int b = tile / (p.tiles_x * p.tiles_y);
        const int tr = tile - b * (p.tiles_x * p.tiles_y);
        const int ty = tr / p.tiles_x, tx = tr - ty * p.tiles_x;
        const int y0 = 2 * ty * S2_TY - 1, x0 = 2 * tx * S2_TX - 1;
        const auto rs_img = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.img + (size_t)b * 3 * p.H * p.W), 0, 3 * p.H * p.W * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < NSTAGE; ++i) {
            const int e = ptid + i * S2_PATCH_THREADS;
            const int c = e / (S2_PR * S2_PC), r = e - c * (S2_PR * S2_PC);
            const int py = r / S2_PC, px = r - py * S2_PC;
            const int gy = y0 - 1 + py, gx = x0 - 1 + px;          // patch row 0 / column 0: one above / left of conv1's first
            const bool ok = tile < p.total_tiles && ptid >= 0 && e < S2_P_FLOATS && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
            stage[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_img, ok ? ((c * p.H + gy) * p.W + gx) * 4 : (int)0x80000000, 0, 0));
        }
    };
    auto store_patch = [&]() {
        if (ptid < 0) return;
#pragma unroll
        for (int i = 0; i < NSTAGE; ++i) {
            const int e = ptid + i * S2_PATCH_THREADS;
            if (e < S2_P_FLOATS) sP[e] = stage[i];
        }
    };
    request_patch(blockIdx.x);
    if constexpr (THIRD) store_patch();         // the first tile's; every later tile's during the third layer of the tile before it
    for (int tile = blockIdx.x; tile < p.total_tiles; tile += gridDim.x) {
        const int b = tile / (p.tiles_x * p.tiles_y);
        const int tr = tile - b * (p.tiles_x * p.tiles_y);
        const int ty = tr / p.tiles_x, tx = tr - ty * p.tiles_x;
        const int oy0 = ty * S2_TY, ox0 = tx * S2_TX;
        const int y0 = 2 * oy0 - 1, x0 = 2 * ox0 - 1;       // conv1 activation (row 0, column 0) of the tile
#ifdef OM_S2_TRACE
        unsigned long long t0, t1, t2, t3, t4;      // tools/stem2_trace.py
        asm volatile("s_memtime %0" : "=s"(t0)::"memory");
#endif
        // ---- 1. the image patch: rows y0 - 1 .., columns x0 - 1 ..
        if constexpr (!THIRD) store_patch();
        __syncthreads();
#ifdef OM_S2_TRACE
        asm volatile("s_memtime %0" : "=s"(t1)::"memory");
#endif
        // ---- 2. conv1 + BatchNorm + LeakyReLU, hi/lo split, into conv2.0's matrix operand layout
#pragma unroll 1
        for (int blk = wave; blk < (S2_SR * S2_SC + 31) / 32; blk += S2_THREADS / 64) {
            const int pix = blk * 32 + fi;
            const int pc = pix < S2_SR * S2_SC ? pix : S2_SR * S2_SC - 1;       // lanes beyond the last activation repeat it
            const int r = pc / S2_SC, col = pc - r * S2_SC;
            const float* win = sP + r * S2_PC + col;
            f32x4 sc1[4], sh1[4];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                sc1[gq] = *reinterpret_cast<const f32x4*>(p.sc1 + 8 * gq + 4 * fk);
                sh1[gq] = *reinterpret_cast<const f32x4*>(p.sh1 + 8 * gq + 4 * fk);
            }
            f32x16 acc1;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc1[q] = 0.f;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    v[i] = win[koff[s2][i]];
                    if (s2 == 1 && i >= 3) v[i] = fk ? 0.f : v[i];      // k = 27..31 do not exist
                }
                f16x8 ah, al;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    ah[i] = (_Float16)v[i];
                    al[i] = (_Float16)(v[i] - (float)ah[i]);
                }
                // weights first: D[i = channel][j = activation]
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1h[s2], al, acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1l[s2], ah, acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1h[s2], ah, acc1, 0, 0, 0);
            }
            const bool inside = pix < S2_SR * S2_SC && (unsigned)(y0 + r) < (unsigned)p.H && (unsigned)(x0 + col) < (unsigned)p.W;
            // slot: even columns first, then odd ones, so that the outputs of a tile row read consecutive slots for every tap
            const int slot = r * S2_SLOTS + (col & 1) * 17 + (col >> 1);
            const int sq = (slot >> 1) & 3, sg = (slot >> 3) & 1;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                // channels 8 gq + 4 fk ..: quad 2 gq + fk -> 16-channel group gq >> 1, operand chunk fk, half gq & 1 of the chunk
                f32x4 o;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float t = fmaf(acc1[4 * gq + k], sc1[gq][k], sh1[gq][k]);
                    o[k] = inside ? (t > 0.f ? t : t * 0.1f) : 0.f;         // outside the image: conv2.0's zero padding
                }
                const f16x4 h = __builtin_convertvector(o, f16x4);
                f16x4 l;
#pragma unroll
                for (int k = 0; k < 4; ++k) l[k] = (_Float16)(o[k] - (float)h[k]);
                if (pix < S2_SR * S2_SC) {
                    char* row = sS + slot * 128 + (((gq >> 1) ^ sg) * 64) + (gq & 1) * 8;
                    *reinterpret_cast<u32x2*>(row + ((fk ^ sq) * 16)) = __builtin_bit_cast(u32x2, h);
                    *reinterpret_cast<u32x2*>(row + (((2 + fk) ^ sq) * 16)) = __builtin_bit_cast(u32x2, l);
                }
            }
        }
        __syncthreads();
#ifdef OM_S2_TRACE
        asm volatile("s_memtime %0" : "=s"(t2)::"memory");
#endif
        request_patch(tile + gridDim.x);        // the patch is dead: its registers take the next tile's
        // the third layer's weights of this lane (fi = output channel, fk), hi and lo of its four k-steps: requested a matrix phase
        // and an epilogue before their use (~2 500 cycles from L2 under this kernel's load)
        f16x8 w3h[4], w3l[4];
        if constexpr (THIRD) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const f32x4* row = reinterpret_cast<const f32x4*>(p.w3 + fi * 128 + ks * 32);      // 64 B per k-step: [hi | hi | lo | lo]
                w3h[ks] = __builtin_bit_cast(f16x8, row[fk]);
                w3l[ks] = __builtin_bit_cast(f16x8, row[2 + fk]);
            }
        }
        // ---- 3. conv2.0
        f32x16 acc2;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
        // fragments one step ahead of the matrix instructions that use them
        f32x4 cur[4], nxt[4];
        // (the output's position made opaque per tile: as loop invariants the 18 steps' swizzled LDS addresses were kept in ~40
        // registers across the whole tile loop, and the kernel spilled)
        int oy_t = oy, ox_t = ox;
        asm volatile("" : "+v"(oy_t), "+v"(ox_t));
        auto read_step = [&](f32x4(&f)[4], int step) {
            const int tap = step >> 1, c = step & 1;
            const int kh = tap / 3, kw = tap - 3 * kh;
            const int slot = (2 * oy_t + kh) * S2_SLOTS + (kw & 1) * 17 + ox_t + (kw >> 1);
            const int sq = (slot >> 1) & 3, sg = (slot >> 3) & 1;
            const char* arow = sS + slot * 128 + ((c ^ sg) * 64);
            const char* brow = sW + step * 4096;
            f[0] = *reinterpret_cast<const f32x4*>(arow + ((fk ^ sq) * 16));
            f[1] = *reinterpret_cast<const f32x4*>(arow + (((2 + fk) ^ sq) * 16));
            f[2] = *reinterpret_cast<const f32x4*>(brow + boff_hi);
            f[3] = *reinterpret_cast<const f32x4*>(brow + boff_lo);
        };
        read_step(cur, 0);
#pragma unroll
        for (int step = 0; step < 18; ++step) {
            if (step + 1 < 18) read_step(nxt, step + 1);
            const f16x8 ah = __builtin_bit_cast(f16x8, cur[0]), al = __builtin_bit_cast(f16x8, cur[1]);
            const f16x8 bh = __builtin_bit_cast(f16x8, cur[2]), bl = __builtin_bit_cast(f16x8, cur[3]);
            // weights first: D[i = channel][j = output], the three products in conv_igemm_split_kernel's order
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, al, acc2, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, ah, acc2, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ah, acc2, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) cur[i] = nxt[i];
        }
#ifdef OM_S2_TRACE
        asm volatile("s_memtime %0" : "=s"(t3)::"memory");
#endif
        __syncthreads();        // every wave is done with the activations: their LDS takes the transposes
        // the third layer's operands of this lane (fi = output channel, fk): hi and lo weights of its four k-steps, scale and shift --
        // requested HERE, an epilogue before their use (requested where they are used, their round trip to L2 stood in front of every
        // tile's last stores: 0.12 ms per launch; held for the kernel's life, or across conv2.0's matrix phase, they cost that phase
        // its registers)
        f32x4 s3v[4], h3v[4];
        if constexpr (THIRD) {       // (every wave, no branch around the requests: the compiler's wait-count bookkeeping takes the worst path)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                s3v[ks] = *reinterpret_cast<const f32x4*>(p.sc3 + 8 * ks + 4 * fk);
                h3v[ks] = *reinterpret_cast<const f32x4*>(p.sh3 + 8 * ks + 4 * fk);
            }
        }
        // ---- 4. epilogue: 32 x 32 transpose through 4 KiB of the wave's own, then one channel quad of four outputs per lane
        {
            f32x4* sT = reinterpret_cast<f32x4*>(sS) + wave * 256;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const f32x4 v = {acc2[4 * gq], acc2[4 * gq + 1], acc2[4 * gq + 2], acc2[4 * gq + 3]};
                sT[fi * 8 + ((2 * gq + fk) ^ (fi & 7))] = v;
            }
            const auto rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out + (size_t)b * p.Ho * p.Wo * p.out_ps, 0, p.Ho * p.Wo * p.out_ps * 4, 0x00020000);
#pragma unroll
            for (int rd = 0; rd < 4; ++rd) {
                const int e = 8 * rd + (lane >> 3);
                f32x4 v = sT[e * 8 + (c8 ^ (e & 7))];
                const int mm = 32 * wm + e;
                const int oyy = oy0 + (mm >> 4), oxx = ox0 + (mm & 15);
                const bool ok = oyy < p.Ho && oxx < p.Wo;
                float nf = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float t = fmaf(v[k], sc2[k], sh2[k]);
                    nf = fmaf(t, 0.f, nf);
                    v[k] = p.leaky2 ? (t > 0.f ? t : t * 0.1f) : t;
                }
                nonfinite += ok ? nf : 0.f;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4s, v), rs_out,
                                                       ok ? ((oyy * p.Wo + oxx) * p.out_ps + nb) * 4 : (int)0x80000000, 0, 0);
                // the third layer's operand: the tile's activation as fp32 rows [output][64 channels] behind the transposes, the
                // 16-byte chunk index XOR-ed with s2_row_swizzle(row): a wave WRITES eight chunks of eight rows here and READS one
                // chunk of 32 rows below, both without bank conflicts (with the plain row & 15 the writes were 8-way conflicts,
                // eight waves at once: 3 000 cycles per tile, tools/stem2_trace.py)
                if constexpr (THIRD) *reinterpret_cast<f32x4*>(sS + 32768 + mm * 256 + (((nb >> 2) ^ s2_row_swizzle(mm)) * 16)) = v;
            }
        }
#ifdef OM_S2_TRACE
        unsigned long long t3b;
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t3b)::"memory");
#endif
        if constexpr (THIRD) {
            __syncthreads();        // the activation rows are complete
            if (wave >= 4) store_patch();       // the next tile's patch (requested before conv2.0: landed long ago) beside the third layer
            if (wave < 4) {
                // 32 outputs x 32 channels x K = 64 per wave, conv_igemm_split_kernel's arithmetic: per 16 channels the lane's two
                // chunks {4 fk .., 8 + 4 fk ..} split in registers (split8), three matrix instructions in its order
                const int mrow = 32 * wave + fi;
                const char* arow = sS + 32768 + mrow * 256;
                f32x16 acc3;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc3[r] = 0.f;
                f32x4 xr[4][2];          // all eight fragments first: one LDS round trip instead of four
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    xr[ks][0] = *reinterpret_cast<const f32x4*>(arow + (((4 * ks + fk) ^ s2_row_swizzle(mrow)) * 16));
                    xr[ks][1] = *reinterpret_cast<const f32x4*>(arow + (((4 * ks + 2 + fk) ^ s2_row_swizzle(mrow)) * 16));
                }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const f32x4 x0 = xr[ks][0], x1 = xr[ks][1];
                    f16x8 ah, al;
#pragma unroll
                    for (int i = 0; i < 4; ++i) { ah[i] = (_Float16)x0[i]; ah[4 + i] = (_Float16)x1[i]; }
#pragma unroll
                    for (int i = 0; i < 4; ++i) { al[i] = (_Float16)(x0[i] - (float)ah[i]); al[4 + i] = (_Float16)(x1[i] - (float)ah[4 + i]); }
                    acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w3h[ks], al, acc3, 0, 0, 0);
                    acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w3l[ks], ah, acc3, 0, 0, 0);
                    acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w3h[ks], ah, acc3, 0, 0, 0);
                }
                // D[i = channel][j = output]: the lane holds channels 8 g + 4 fk .. + 3 (g = 0 .. 3) of output fi
                const int oyy = oy0 + (mrow >> 4), oxx = ox0 + (mrow & 15);
                const bool ok = oyy < p.Ho && oxx < p.Wo;
                const auto rs_out3 = __builtin_amdgcn_make_buffer_rsrc(p.out3 + (size_t)b * p.Ho * p.Wo * p.out3_ps, 0, p.Ho * p.Wo * p.out3_ps * 4, 0x00020000);
                float nf = 0.f;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const f32x4 s3 = s3v[gq], h3 = h3v[gq];
                    f32x4 o;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float t = fmaf(acc3[4 * gq + k], s3[k], h3[k]);
                        nf = fmaf(t, 0.f, nf);
                        o[k] = p.leaky3 ? (t > 0.f ? t : t * 0.1f) : t;
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4s, o), rs_out3,
                                                           ok ? ((oyy * p.Wo + oxx) * p.out3_ps + 8 * gq + 4 * fk) * 4 : (int)0x80000000, 0, 0);
                }
                nonfinite += ok ? nf : 0.f;
            }
        }
#ifdef OM_S2_TRACE
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t4)::"memory");
        if (blockIdx.x == 0 && tid == 0 && tile == blockIdx.x + 3 * gridDim.x) {
            unsigned long long* tr = reinterpret_cast<unsigned long long*>(p.status);
            tr[0] = t0; tr[1] = t1; tr[2] = t2; tr[3] = t3; tr[4] = t4; tr[5] = t3b;
        }
#endif
        // the next tile's barrier (after its patch is staged) orders these transposes before the next activations
    }
#ifndef OM_S2_TRACE
    if (p.status && nonfinite != nonfinite) atomicOr(p.status, OM_STATUS_SPLIT_RANGE);
#endif
}

// ------------------------------------------------------------------------------------------------------------------------------
// The same two layers in the fp16-activation configuration (precision mode 2; round 5): conv1 from the fp32 image with fp32
// weights (split operands on the matrix pipe as above: fp32-level sums, then ONE rounding to fp16 -- what conv_stem_kernel<f16>
// stores), conv2.0 on fp16 operands (one matrix instruction per step, fp16 weight rows of pack.py: conv_weights_f16), fp16 NHWC
// out.  Separately the 32-channel full-resolution activation costs 0.30 + 0.30 ms at bs = 32 (606 MB written, 606 MB read).
// Without the lo halves a workgroup needs 36 864 (weights: 18 steps x 64 rows x 32 B) + 36 992 (activations: 17 rows x 34 slots
// of 64 B) + 7 980 (patch) = 81 836 bytes: TWO workgroups per CU, whose phases overlap by themselves.
constexpr int S2H_SLOTS = 34;                           // 17 even columns, then 16 odd ones (+ 1)
constexpr int S2H_W_BYTES = 18 * 64 * 32;
constexpr int S2H_S_BYTES = S2_SR * S2H_SLOTS * 64;
typedef unsigned u32x2s __attribute__((__vector_size__(2 * sizeof(unsigned))));

struct Stem2HParams {
    const float* img;       // [B,3,H,W]
    const float* w1;        // [32][27]
    const float* sc1;
    const float* sh1;
    const _Float16* w2;     // conv2.0's fp16 rows [64][9][32]
    const float* sc2;
    const float* sh2;
    _Float16* out;          // NHWC [B, H/2, W/2, out_ps]
    int B, H, W, Ho, Wo, out_ps, leaky2;
    int tiles_x, tiles_y, total_tiles;
};

__global__ __launch_bounds__(S2_THREADS, 4) void conv_stem2_f16_kernel(const Stem2HParams p) {
    __shared__ f32x4 smem[(S2H_W_BYTES + S2H_S_BYTES + S2_P_FLOATS * 4 + 15) / 16];
    char* const sW = reinterpret_cast<char*>(smem);
    char* const sS = sW + S2H_W_BYTES;
    float* const sP = reinterpret_cast<float*>(sS + S2H_S_BYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- conv2.0's weights: global [row][step][half] (16-byte pieces) -> LDS [step][row][half ^ ((row >> 3) & 1)], once
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(p.w2);
#pragma unroll
        for (int i = 0; i < (S2H_W_BYTES / 16 + S2_THREADS - 1) / S2_THREADS; ++i) {
            const int idx = tid + i * S2_THREADS;
            if (idx < S2H_W_BYTES / 16) {
                const int n = idx / 36, rem = idx - n * 36;
                const int step = rem >> 1, h = rem & 1;
                *reinterpret_cast<f32x4*>(sW + step * 2048 + n * 32 + ((h ^ ((n >> 3) & 1)) * 16)) = src[idx];
            }
        }
    }
    const int fi = lane & 31, fk = lane >> 5;
    f16x8 w1h[2], w1l[2];
    // patch offset of contraction index k = (kh * 3 + kw) * 3 + ci: a compile-time table, selected by fk where it is used (sixteen
    // registers of offsets held across the tile loop were what spilled at this kernel's 128)
    auto koff_of = [](int k) constexpr {
        const int kk = k < 27 ? k : 0;
        const int t = kk / 3, ci = kk - 3 * t;
        const int kh = t / 3, kw = t - 3 * kh;
        return (ci * S2_PR + kh) * S2_PC + kw;
    };
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = 16 * s2 + 8 * fk + i;
            const int kk = k < 27 ? k : 0;
            const float w = k < 27 ? p.w1[fi * 27 + kk] : 0.f;
            const _Float16 h = (_Float16)w;
            w1h[s2][i] = h;
            w1l[s2][i] = (_Float16)(w - (float)h);
        }
    // conv2.0's role: wave = 32 outputs (two tile rows) x 32 channels
    const int wm = wave >> 1, wn = wave & 1;
    const int m = 32 * wm + fi;
    const int oy = m >> 4, ox = m & 15;
    const int nrow = 32 * wn + fi;
    const int boff = nrow * 32 + ((fk ^ ((nrow >> 3) & 1)) * 16);
    const int c8 = lane & 7;
    const int nb = 32 * wn + 4 * c8;
    asm volatile("" ::"v"(w1h[0]), "v"(w1h[1]), "v"(w1l[0]), "v"(w1l[1]));

    constexpr int NSTAGE = (S2_P_FLOATS + S2_THREADS - 1) / S2_THREADS;
    float stage[NSTAGE];
    auto request_patch = [&](int tile) {
        const int b = tile / (p.tiles_x * p.tiles_y);
        const int tr = tile - b * (p.tiles_x * p.tiles_y);
        const int ty = tr / p.tiles_x, tx = tr - ty * p.tiles_x;
        const int y0 = 2 * ty * S2_TY - 1, x0 = 2 * tx * S2_TX - 1;
        const auto rs_img = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.img + (size_t)b * 3 * p.H * p.W), 0, 3 * p.H * p.W * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < NSTAGE; ++i) {
            const int e = tid + i * S2_THREADS;
            const int c = e / (S2_PR * S2_PC), r = e - c * (S2_PR * S2_PC);
            const int py = r / S2_PC, px = r - py * S2_PC;
            const int gy = y0 - 1 + py, gx = x0 - 1 + px;
            const bool ok = tile < p.total_tiles && e < S2_P_FLOATS && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
            stage[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_img, ok ? ((c * p.H + gy) * p.W + gx) * 4 : (int)0x80000000, 0, 0));
        }
    };
    request_patch(blockIdx.x);
    for (int tile = blockIdx.x; tile < p.total_tiles; tile += gridDim.x) {
        const int b = tile / (p.tiles_x * p.tiles_y);
        const int tr = tile - b * (p.tiles_x * p.tiles_y);
        const int ty = tr / p.tiles_x, tx = tr - ty * p.tiles_x;
        const int oy0 = ty * S2_TY, ox0 = tx * S2_TX;
        const int y0 = 2 * oy0 - 1, x0 = 2 * ox0 - 1;
        // ---- 1. the image patch
#pragma unroll
        for (int i = 0; i < NSTAGE; ++i) {
            const int e = tid + i * S2_THREADS;
            if (e < S2_P_FLOATS) sP[e] = stage[i];
        }
        __syncthreads();
        // ---- 2. conv1 + BatchNorm + LeakyReLU -> fp16, into conv2.0's operand layout: slot x 64 B, the 16-byte chunk (8 channels)
        // index XOR-ed with (slot >> 2) & 3 (conv_f16_common.h's 64-byte rows)
#pragma unroll 1
        for (int blk = wave; blk < (S2_SR * S2_SC + 31) / 32; blk += S2_THREADS / 64) {
            const int pix = blk * 32 + fi;
            const int pc = pix < S2_SR * S2_SC ? pix : S2_SR * S2_SC - 1;
            const int r = pc / S2_SC, col = pc - r * S2_SC;
            const float* win = sP + r * S2_PC + col;
            f32x4 s1v[4], h1v[4];          // requested in front of the window gather: their round trip runs under it
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                s1v[gq] = *reinterpret_cast<const f32x4*>(p.sc1 + 8 * gq + 4 * fk);
                h1v[gq] = *reinterpret_cast<const f32x4*>(p.sh1 + 8 * gq + 4 * fk);
            }
            int fk_here = fk;
            asm volatile("" : "+v"(fk_here));          // the selects below stay in the loop
            f32x16 acc1;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc1[q] = 0.f;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    v[i] = win[fk_here ? koff_of(16 * s2 + 8 + i) : koff_of(16 * s2 + i)];
                    if (s2 == 1 && i >= 3) v[i] = fk ? 0.f : v[i];
                }
                f16x8 ah, al;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    ah[i] = (_Float16)v[i];
                    al[i] = (_Float16)(v[i] - (float)ah[i]);
                }
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1h[s2], al, acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1l[s2], ah, acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1h[s2], ah, acc1, 0, 0, 0);
            }
            const bool inside = pix < S2_SR * S2_SC && (unsigned)(y0 + r) < (unsigned)p.H && (unsigned)(x0 + col) < (unsigned)p.W;
            const int slot = r * S2H_SLOTS + (col & 1) * 17 + (col >> 1);
            const int sw = (slot >> 2) & 3;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                // channels 8 gq + 4 fk ..: chunk gq, half fk of the chunk
                const f32x4 s1 = s1v[gq], h1 = h1v[gq];
                f32x4 o;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float t = fmaf(acc1[4 * gq + k], s1[k], h1[k]);
                    o[k] = inside ? (t > 0.f ? t : t * 0.1f) : 0.f;
                }
                const f16x4 h = __builtin_convertvector(o, f16x4);
                if (pix < S2_SR * S2_SC) *reinterpret_cast<u32x2*>(sS + slot * 64 + ((gq ^ sw) * 16) + fk * 8) = __builtin_bit_cast(u32x2, h);
            }
        }
        __syncthreads();
        request_patch(tile + gridDim.x);
        // ---- 3. conv2.0: 9 taps x 2 chunks of 16 channels, fragments one step ahead
        f32x16 acc2;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
        f32x4 cur[2], nxt[2];
        int oy_t = oy, ox_t = ox;          // opaque per tile (see conv_stem2_split_kernel)
        asm volatile("" : "+v"(oy_t), "+v"(ox_t));
        auto read_step = [&](f32x4(&f)[2], int step) {
            const int tap = step >> 1, c = step & 1;
            const int kh = tap / 3, kw = tap - 3 * kh;
            const int slot = (2 * oy_t + kh) * S2H_SLOTS + (kw & 1) * 17 + ox_t + (kw >> 1);
            const int sw = (slot >> 2) & 3;
            f[0] = *reinterpret_cast<const f32x4*>(sS + slot * 64 + (((2 * c + fk) ^ sw) * 16));
            f[1] = *reinterpret_cast<const f32x4*>(sW + step * 2048 + boff);
        };
        read_step(cur, 0);
#pragma unroll
        for (int step = 0; step < 18; ++step) {
            if (step + 1 < 18) read_step(nxt, step + 1);
            // weights first: D[i = channel][j = output]
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, cur[1]), __builtin_bit_cast(f16x8, cur[0]), acc2, 0, 0, 0);
            cur[0] = nxt[0]; cur[1] = nxt[1];
        }
        __syncthreads();        // every wave is done with the activations: their LDS takes the transposes
        // ---- 4. epilogue: 32 x 32 transpose through 4 KiB of the wave's own, then one channel quad of four outputs per lane
        {
            f32x4* sT = reinterpret_cast<f32x4*>(sS) + wave * 256;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const f32x4 v = {acc2[4 * gq], acc2[4 * gq + 1], acc2[4 * gq + 2], acc2[4 * gq + 3]};
                sT[fi * 8 + ((2 * gq + fk) ^ (fi & 7))] = v;
            }
            const f32x4 sc2 = *reinterpret_cast<const f32x4*>(p.sc2 + nb);      // per tile, from cache: eight registers less across the loop
            const f32x4 sh2 = *reinterpret_cast<const f32x4*>(p.sh2 + nb);
            const auto rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out + (size_t)b * p.Ho * p.Wo * p.out_ps, 0, p.Ho * p.Wo * p.out_ps * 2, 0x00020000);
#pragma unroll
            for (int rd = 0; rd < 4; ++rd) {
                const int e = 8 * rd + (lane >> 3);
                const f32x4 v = sT[e * 8 + (c8 ^ (e & 7))];
                const int mm = 32 * wm + e;
                const int oyy = oy0 + (mm >> 4), oxx = ox0 + (mm & 15);
                const bool ok = oyy < p.Ho && oxx < p.Wo;
                f32x4 o;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float t = fmaf(v[k], sc2[k], sh2[k]);
                    o[k] = p.leaky2 ? (t > 0.f ? t : t * 0.1f) : t;
                }
                const f16x4 h = __builtin_convertvector(o, f16x4);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2s, h), rs_out,
                                                      ok ? ((oyy * p.Wo + oxx) * p.out_ps + nb) * 2 : (int)0x80000000, 0, 0);
            }
        }
    }
}

int launch_conv_stem2_f16(const float* in_nchw, int B, int H, int W, const float* w1, const float* scale1, const float* shift1,
                          const void* w2_f16, const float* scale2, const float* shift2, int cout2, int leaky2, void* out_nhwc_f16,
                          int out_pix_stride, hipStream_t stream) {
    OM_REQUIRE(in_nchw && w1 && scale1 && shift1 && w2_f16 && scale2 && shift2 && out_nhwc_f16, OM_EINVAL, "stem2 f16: null pointer");
    OM_REQUIRE(cout2 == 64, OM_EINVAL, "stem2 f16: cout=%d, only 64 supported", cout2);
    OM_REQUIRE(B > 0 && H > 1 && W > 1 && H % 2 == 0 && W % 2 == 0, OM_EINVAL, "stem2 f16: bad shape B=%d H=%d W=%d", B, H, W);
    OM_REQUIRE(out_pix_stride % 4 == 0 && out_pix_stride >= 64 && (reinterpret_cast<uintptr_t>(out_nhwc_f16) & 7) == 0 &&
                   (reinterpret_cast<uintptr_t>(w2_f16) & 15) == 0,
               OM_EINVAL, "stem2 f16: output view must be 8-byte, the weights 16-byte aligned");
    OM_REQUIRE((long long)(H / 2) * (W / 2) * out_pix_stride * 2 < 0x7FFFFFF0ll && (long long)3 * H * W * 4 < 0x7FFFFFF0ll, OM_EINVAL,
               "stem2 f16: an image of %d x %d exceeds a buffer descriptor", H, W);
    Stem2HParams p;
    p.img = in_nchw; p.w1 = w1; p.sc1 = scale1; p.sh1 = shift1;
    p.w2 = static_cast<const _Float16*>(w2_f16); p.sc2 = scale2; p.sh2 = shift2;
    p.out = static_cast<_Float16*>(out_nhwc_f16);
    p.B = B; p.H = H; p.W = W; p.Ho = H / 2; p.Wo = W / 2; p.out_ps = out_pix_stride; p.leaky2 = leaky2;
    p.tiles_x = (p.Wo + S2_TX - 1) / S2_TX; p.tiles_y = (p.Ho + S2_TY - 1) / S2_TY;
    const long long total = (long long)B * p.tiles_x * p.tiles_y;
    OM_REQUIRE(total < (1ll << 31), OM_EINVAL, "stem2 f16: %lld tiles out of range", total);
    p.total_tiles = (int)total;
    const unsigned grid = (unsigned)(total < 512 ? total : 512);
    hipLaunchKernelGGL(conv_stem2_f16_kernel, dim3(grid), dim3(S2_THREADS), 0, stream, p);
    OM_CHECK_HIP(hipGetLastError());
    return OM_OK;
}

// w2_split / scale2_split: conv2.0's packed hi/lo weights and scale * 2^-e (include/orienmask_hip.h: om_layer_info.wsplit_off)
int launch_conv_stem2_split(const float* in_nchw, int B, int H, int W, const float* w1, const float* scale1, const float* shift1,
                            const void* w2_split, const float* scale2_split, const float* shift2, int cout2, int leaky2,
                            float* out_nhwc, int out_pix_stride, int* status, hipStream_t stream, const Stem2Third* third) {
    OM_REQUIRE(in_nchw && w1 && scale1 && shift1 && w2_split && scale2_split && shift2 && out_nhwc, OM_EINVAL, "stem2: null pointer");
    OM_REQUIRE(cout2 == 64, OM_EINVAL, "stem2: cout=%d, only 64 supported", cout2);
    OM_REQUIRE(B > 0 && H > 1 && W > 1 && H % 2 == 0 && W % 2 == 0, OM_EINVAL, "stem2: bad shape B=%d H=%d W=%d", B, H, W);
    OM_REQUIRE(out_pix_stride % 4 == 0 && out_pix_stride >= 64 && (reinterpret_cast<uintptr_t>(out_nhwc) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(w2_split) & 15) == 0,
               OM_EINVAL, "stem2: output view / weights must be 16-byte aligned");
    OM_REQUIRE((long long)(H / 2) * (W / 2) * out_pix_stride * 4 < 0x7FFFFFF0ll && (long long)3 * H * W * 4 < 0x7FFFFFF0ll, OM_EINVAL,
               "stem2: an image of %d x %d exceeds a buffer descriptor", H, W);
    Stem2Params p;
    p.img = in_nchw; p.w1 = w1; p.sc1 = scale1; p.sh1 = shift1;
    p.w2 = static_cast<const _Float16*>(w2_split); p.sc2 = scale2_split; p.sh2 = shift2;
    p.out = out_nhwc; p.status = status;
    p.B = B; p.H = H; p.W = W; p.Ho = H / 2; p.Wo = W / 2; p.out_ps = out_pix_stride; p.leaky2 = leaky2;
    p.tiles_x = (p.Wo + S2_TX - 1) / S2_TX; p.tiles_y = (p.Ho + S2_TY - 1) / S2_TY;
    const long long total = (long long)B * p.tiles_x * p.tiles_y;
    OM_REQUIRE(total < (1ll << 31), OM_EINVAL, "stem2: %lld tiles out of range", total);
    p.total_tiles = (int)total;
    p.w3 = nullptr; p.sc3 = nullptr; p.sh3 = nullptr; p.out3 = nullptr; p.out3_ps = 0; p.leaky3 = 0;
    if (third) {
        OM_REQUIRE(third->w_split && third->scale_split && third->shift && third->out && third->cout == 32 && third->out_pix_stride % 4 == 0 &&
                       third->out_pix_stride >= 32 && (reinterpret_cast<uintptr_t>(third->out) & 15) == 0 &&
                       (reinterpret_cast<uintptr_t>(third->w_split) & 15) == 0 &&
                       (long long)(H / 2) * (W / 2) * third->out_pix_stride * 4 < 0x7FFFFFF0ll,
                   OM_EINVAL, "stem2: the third layer must be a 64 -> 32 1x1 with a 16-byte aligned output view");
        p.w3 = static_cast<const _Float16*>(third->w_split); p.sc3 = third->scale_split; p.sh3 = third->shift;
        p.out3 = third->out; p.out3_ps = third->out_pix_stride; p.leaky3 = third->leaky;
    }
    const unsigned grid = (unsigned)(total < 256 ? total : 256);
    if (third) hipLaunchKernelGGL(conv_stem2_split_kernel<true>, dim3(grid), dim3(S2_THREADS), 0, stream, p);
    else hipLaunchKernelGGL(conv_stem2_split_kernel<false>, dim3(grid), dim3(S2_THREADS), 0, stream, p);
    OM_CHECK_HIP(hipGetLastError());
    return OM_OK;
}

}  // namespace om
