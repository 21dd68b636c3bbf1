// Fused convolution for gfx950 with SPLIT fp32 operands (precision mode 1, include/orienmask_hip.h: om_model_set_precision):
// fp32 activations in, fp32 activations out, but the products run on the fp16 matrix pipe.
//
//   out[m][n] = act( (sum_k A[m][k] * Wt[n][k]) * scale[n] + shift[n] ) (+ res[m][n])            (conv_igemm.hip's layer)
//   A[m][k] = a_hi + a_lo,  Wt[n][k] * 2^e[n] = w_hi + w_lo   (hi = fp16(x), lo = fp16(x - hi): x to ~2^-22 |x|)
//   A * Wt  ~=  a_hi w_hi + a_hi w_lo + a_lo w_hi             (three v_mfma_f32_32x32x16_f16 with fp32 accumulation; the
//                                                              dropped a_lo w_lo is <= 2^-22 of the product)
//
// The f32-input matrix instruction (v_mfma_f32_32x32x2_f32) peaks at 157 TFLOP/s, the fp16 one at 2.5 PFLOP/s: three fp16
// instructions per product group are 5.3x faster than the fp32 form, at fp32-level accuracy (measured against float64:
// tools/split_error.py) -- the 1x1 and stride-2 layers stop being bound by the matrix pipe.
//
// Same layer as conv_igemm.hip (Conv2d -> BatchNorm2d(eval) -> LeakyReLU(0.1), residual add, nearest upsample, concat by
// slice, NCHW orientation head: /root/reference/model/base.py:95-137, backbone/darknet.py:14-15,
// orienmask_yolo_fpnplus.py:78-86); the ring / LDS-DMA / tile-queue structure is conv_igemm_f16.hip's, because the byte
// rates per matrix instruction are those of the fp16 kernels:
//   * a k-step is 16 input channels: a 64-byte row of fp32 activations (four 16-byte chunks) and a 64-byte row of packed
//     weights [8 hi | 8 hi | 8 lo | 8 lo] halfs.  Lane half fk reads chunks fk and 2 + fk of both: for A that is channels
//     {4 fk .. 4 fk + 3, 8 + 4 fk .. 8 + 4 fk + 3}, and the packer stores the weights' hi and lo halfs in exactly that order
//     (orienmask_amd/pack.py: conv_weights_split), so A is split in registers (one fp16 conversion for hi, a subtraction and
//     a second conversion for lo) and B needs no arithmetic at all;
//   * 3-deep operand ring filled by LDS-DMA two k-steps ahead (counted vmcnt, raw s_barrier); per k-step one barrier:
//     convert(s+1) and the LDS reads of step s+1 run under the 3 TM TN matrix instructions of step s;
//   * fp32 epilogue through LDS one wave-row at a time, eight channels (two 16-byte stores) per thread.
#include <cstdlib>

#include "conv_f16_common.h"

namespace om {

#ifndef OM_SPLIT_WIDE
#define OM_SPLIT_WIDE 1        // 128 x 128 tiles with 128-byte operand rows (conv_igemm_split_wide_kernel) where cin % 32 == 0
#endif

#if defined(OM_SPLIT_NO_XCD_PLACEMENT) && !defined(OM_MEASUREMENT_BUILD)
#error "OM_SPLIT_NO_XCD_PLACEMENT is a measurement switch: tools/build_variant.sh only"
#endif
#ifndef OM_SPLIT_TRACE
#define OM_SPLIT_TRACE 0       // measurement builds only: s_memtime stamps per tile of the wide kernel (tools/split_trace.py)
#endif
#if (OM_SPLIT_TRACE) && !defined(OM_MEASUREMENT_BUILD)
#error "measurement switches (wrong numerics / trace stores) are only for ab/ variants: build them with tools/build_variant.sh, which defines OM_MEASUREMENT_BUILD and never writes orienmask_amd/lib/"
#endif
#if OM_SPLIT_TRACE
static unsigned long long* g_split_trace = nullptr;
extern "C" void om_debug_split_trace(void* buf) { g_split_trace = static_cast<unsigned long long*>(buf); }
#define SPLIT_STAMP(x) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(x)::"memory")
#endif

struct IgemmSParams {
    const _Float16* in;   // the fp32 activations, addressed in halfs (pixel stride and channel counts doubled)
    const _Float16* w;    // packed hi/lo weights: [cout_pad][taps][cin / 16][4][8] halfs
    const float* scale;   // scale * 2^-e per output channel
    const float* shift;
    const float* res;
    float* out;
    int* ticket;
    int* status;          // OM_STATUS_SPLIT_RANGE is OR-ed here when a tile stores a non-finite value (nullptr: not reported)
    int H, W, cin_h, in_pix_stride_h;      // in halfs (= 2 x the float counts)
    int Ho, Wo, HoWo, cout;
    int ks, stride, pad;
    int M, kc, ksteps, taps;               // kc = cin / 16
    int n_tiles, total_tiles;
    // split-K (the deep-ring forms only, conv_igemm_split_kernel<..., NBUF > 3>): a tile's k loop is cut into `ksplit` parts, one
    // ticket each; every part leaves its raw accumulators in `partial`, the part that arrives LAST at the tile's counter
    // (kflags[tile], zeroed with the ticket) sums all of them in part order -- a fixed order, whoever is last -- and stores the tile
    float* partial;
    int* kflags;
    int ksplit, total_tickets, m_tiles;
    int leaky, res_pix_stride, out_pix_stride, out_mode, up;
    int vec_io;
    int total_in_pixels;
    int w_bytes;
    // gathered input of a 1x1 layer (conv_igemm_split_wide_kernel<..., true>): the cin channels are the concatenation of nseg
    // tensors; segment g holds channels [32 seg_end[g-1], 32 seg_end[g]) and is stored at 1 / 2^seg_shift[g] of this layer's
    // resolution -- the nearest-neighbour up-sampling of the reference's routes and skips (orienmask_yolo_fpnplus.py:78-86,
    // F.interpolate + torch.cat) happens in the operand addresses instead of in replicated stores
    int nseg, nimg;
#if OM_SPLIT_TRACE
    unsigned long long* trace;      // [workgroup < 16][tile < 16][8]
#endif
    const _Float16* seg_ptr[4];
    int seg_stride_h[4], seg_shift[4], seg_end[4];
};

template <int BM, int BN>
constexpr int split_blocks_per_cu() { return BM * BN >= 256 * 128 ? 2 : (BM * BN >= 128 * 128 ? 3 : 4); }

// fp32 epilogue of a finished BM x BN tile (accumulators in the transposed 32x32 layout: pixel = lane & 31,
// channel = 8*(r>>2) + 4*(lane>>5) + (r&3)): one wave-row (WM pixels x BN channels) at a time through LDS.
//
// FAST (launch_conv_igemm_split: plain NHWC output, no residual, 16-byte aligned views, cout == cout_pad -- every 1x1 and
// stride-2 layer of the forward but the up-sampling producers and the two heads): the row sweeps contain NO load.  In the
// generic form the residual's conditional loads sit in the sweep loop, and the compiler's wait-count bookkeeping then waits for
// vmcnt(0) at the top of every sweep -- which, loads and stores sharing one in-order counter, is the round trip of the PREVIOUS
// sweep's stores: eight store round trips per tile, 10-18 k cycles of a 40 k-cycle 1x1 tile (tools/split_trace.py,
// profiles/r03_experiments.md section 14; the same serialisation as conv_wino14.hip's first epilogue).
template <int BM, int BN, int WM, int WN, bool FAST = false>
__device__ __forceinline__ void split_epilogue(const IgemmSParams& p, f32x4* smem, const f32x16 (&acc)[WM / 32][WN / 32],
                                               int m0, int n0, int tid, int wm, int wn, int fi, int fk, float (&sc)[8], float (&sh)[8]) {
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int CH8 = BN / 8;
    constexpr int RP = 256 / CH8;
    constexpr int CH = BN / 4;
    f32x4* sC = smem;
    const int n8 = tid % CH8, r0 = tid / CH8;
    const int n = n0 + n8 * 8;
    const int nvalid = p.cout - n;
    const bool vec = p.vec_io && nvalid >= 8;
    if constexpr (!FAST) {      // FAST: requested by the kernel before its k loop (split_scale_shift)
#pragma unroll
        for (int k = 0; k < 8; ++k) { sc[k] = p.scale[n + k]; sh[k] = p.shift[n + k]; }   // padded to cout_pad
    }
    // RANGE GUARD of the split representation (include/orienmask_hip.h: OM_STATUS_SPLIT_RANGE).  An activation beyond fp16's
    // range converts to hi = +-inf, lo = -+inf, whose products sum to NaN in every output channel of that pixel (also against
    // zero weights: 0 * inf), so "this tile stores a non-finite value" is exactly "an operand left the representable range, or the
    // fp32 result itself is non-finite".  v * 0 is NaN for v = +-inf / NaN and +-0 otherwise: one multiply-add per stored value in
    // the store-bound epilogue, nothing in the k-loop.
    float nonfinite = 0.f;
#pragma unroll 1
    for (int pass = 0; pass < BM / WM; ++pass) {
        if (wm == pass) {
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                const int ml = a * 32 + fi;
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int n4 = (wn * WN + b * 32) / 4 + 2 * g + fk;
                        f32x4 v = {acc[a][b][4 * g], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]};
                        sC[ml * CH + (n4 ^ (ml & 7))] = v;
                    }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // LDS-only barrier: the previous pass's stores keep flying
        __builtin_amdgcn_s_barrier();
        if constexpr (FAST) {
#pragma unroll
            for (int ps = 0; ps < (WM + RP - 1) / RP; ++ps) {
                const int ml = ps * RP + r0;
                const int m = m0 + pass * WM + ml;
                if (ml >= WM) continue;
                const f32x4 v0 = sC[ml * CH + ((2 * n8) ^ (ml & 7))];
                const f32x4 v1 = sC[ml * CH + ((2 * n8 + 1) ^ (ml & 7))];
                float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float t = fmaf(v[k], sc[k], sh[k]);
                    v[k] = p.leaky ? (t > 0.f ? t : t * 0.1f) : t;
                    nonfinite = fmaf(t, 0.f, nonfinite);
                }
                if (m < p.M) {
                    float* o = p.out + (size_t)m * p.out_pix_stride + n;
                    *reinterpret_cast<f32x4*>(o) = f32x4{v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<f32x4*>(o + 4) = f32x4{v[4], v[5], v[6], v[7]};
                }
            }
        } else if (p.out_mode != 2) {
#pragma unroll 2
            for (int ps = 0; ps < (WM + RP - 1) / RP; ++ps) {
                const int ml = ps * RP + r0;
                const int m = m0 + pass * WM + ml;
                if (ml >= WM || m >= p.M || nvalid <= 0) continue;
                const f32x4 v0 = sC[ml * CH + ((2 * n8) ^ (ml & 7))];
                const f32x4 v1 = sC[ml * CH + ((2 * n8 + 1) ^ (ml & 7))];
                float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float t = fmaf(v[k], sc[k], sh[k]);
                    v[k] = p.leaky ? (t > 0.f ? t : t * 0.1f) : t;
                    nonfinite = fmaf(t, 0.f, nonfinite);
                }
                if (p.out_mode == 0) {
                    float* o = p.out + (size_t)m * p.out_pix_stride + n;
                    if (p.res) {
                        const float* rp = p.res + (size_t)m * p.res_pix_stride + n;
                        if (vec) {
                            const f32x4 ra = *reinterpret_cast<const f32x4*>(rp), rb = *reinterpret_cast<const f32x4*>(rp + 4);
#pragma unroll
                            for (int k = 0; k < 4; ++k) { v[k] += ra[k]; v[4 + k] += rb[k]; }
                        } else {
                            for (int k = 0; k < 8 && k < nvalid; ++k) v[k] += rp[k];
                        }
                    }
                    if (vec) {
                        *reinterpret_cast<f32x4*>(o) = f32x4{v[0], v[1], v[2], v[3]};
                        *reinterpret_cast<f32x4*>(o + 4) = f32x4{v[4], v[5], v[6], v[7]};
                    } else {
                        for (int k = 0; k < 8 && k < nvalid; ++k) o[k] = v[k];
                    }
                } else {
                    const int bi = m / p.HoWo;
                    const int rr = m - bi * p.HoWo;
                    const int oy = rr / p.Wo, ox = rr - oy * p.Wo;
                    const int Wu = p.Wo * p.up;
                    const size_t base = ((size_t)bi * p.Ho * p.up + (size_t)oy * p.up) * Wu + (size_t)ox * p.up;
                    for (int dy = 0; dy < p.up; ++dy)
                        for (int dx = 0; dx < p.up; ++dx) {
                            float* o = p.out + (base + (size_t)dy * Wu + dx) * p.out_pix_stride + n;
                            if (vec) {
                                *reinterpret_cast<f32x4*>(o) = f32x4{v[0], v[1], v[2], v[3]};
                                *reinterpret_cast<f32x4*>(o + 4) = f32x4{v[4], v[5], v[6], v[7]};
                            } else {
                                for (int k = 0; k < 8 && k < nvalid; ++k) o[k] = v[k];
                            }
                        }
                }
            }
        } else {
            // NCHW output (orientation head): consecutive threads walk pixels of one channel
            const float* sCf = reinterpret_cast<const float*>(smem);
            const int nch = min(BN, p.cout - n0);
            for (int idx = tid; idx < nch * WM; idx += 256) {
                const int nl = idx / WM, ml = idx - nl * WM;
                const int m = m0 + pass * WM + ml;
                if (m >= p.M) continue;
                const int nn = n0 + nl;
                float t = fmaf(sCf[(ml * CH + ((nl >> 2) ^ (ml & 7))) * 4 + (nl & 3)], p.scale[nn], p.shift[nn]);
                nonfinite = fmaf(t, 0.f, nonfinite);
                if (p.leaky) t = t > 0.f ? t : t * 0.1f;
                const int bi = m / p.HoWo;
                const int rr = m - bi * p.HoWo;
                p.out[((size_t)bi * p.cout + nn) * p.HoWo + rr] = t;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    if (p.status && nonfinite != nonfinite) atomicOr(p.status, OM_STATUS_SPLIT_RANGE);
}

// the epilogue's per-channel constants of this thread (split_epilogue: n = n0 + (tid % (BN / 8)) * 8), requested before the k loop
template <int BN>
__device__ __forceinline__ void split_scale_shift(const IgemmSParams& p, int n0, int tid, float (&sc)[8], float (&sh)[8]) {
    const int n = n0 + (tid % (BN / 8)) * 8;
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(p.scale + n), a1 = *reinterpret_cast<const f32x4*>(p.scale + n + 4);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.shift + n), b1 = *reinterpret_cast<const f32x4*>(p.shift + n + 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) { sc[k] = a0[k]; sc[4 + k] = a1[k]; sh[k] = b0[k]; sh[4 + k] = b1[k]; }
}

// hi = fp16(x) (round to nearest even), lo = fp16(x - hi) for the eight channels a lane holds of one pixel
__device__ __forceinline__ void split8(const f32x4& x0, const f32x4& x1, f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        hi[i] = (_Float16)x0[i];
        hi[4 + i] = (_Float16)x1[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        lo[i] = (_Float16)(x0[i] - (float)hi[i]);
        lo[4 + i] = (_Float16)(x1[i] - (float)hi[4 + i]);
    }
}

// NBUF: ring stages.  3 (two k-steps in flight) where a launch has rounds of tiles to overlap; the LATENCY form (launches of
// at most 128 tiles: a batch of one or a few images) has one tile per CU and nothing else to hide a stage's landing time
// behind, so with 3 to 12 matrix instructions per k-step a k-step took ~700 cycles = that landing time / 2: eight stages for
// the 64 x 64 tile (64 KiB), five for 128 x 64 -- two workgroups per CU, which such a launch does not fill anyway.
template <int BM, int BN, int WM, int WN, bool FAST = false, int NBUF = 3>
__global__ __launch_bounds__(256, (NBUF > 3 ? 2 : split_blocks_per_cu<BM, BN>())) void conv_igemm_split_kernel(const IgemmSParams p) {
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int NWN = BN / WN;
    constexpr int A_CH = BM / 64, B_CH = (BN + 63) / 64, NP = A_CH + B_CH;   // 64 rows x 64 B per workgroup-wide piece
    constexpr int STAGE = (BM + (BN < 64 ? 64 : BN)) * 4;   // f32x4 (16-byte) units per ring stage
    static_assert((BM / WM) * (BN / WN) == 4, "four waves per workgroup");
    static_assert(BM % 64 == 0, "A tile = whole 64-row pieces");
    static_assert(WM * BN / 4 <= NBUF * STAGE, "one wave-row of the fp32 C tile must fit in the operand ring");
    __shared__ f32x4 smem[NBUF * STAGE + 1];      // ONE LDS object (see conv_igemm.hip)
    int* const s_ticket = reinterpret_cast<int*>(smem + NBUF * STAGE);      // [0] the ticket, [1] split-K: arrivals before this part
    constexpr bool KSPLIT = NBUF > 3;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave / NWN, wn = wave % NWN;
    const int lrow = tid >> 2, lcol = tid & 3;     // loader: row within a 64-row piece, 16-byte position in the row
    const int scol = lcol ^ ((lrow >> 2) & 3);     // logical chunk this lane fetches (the LDS image stays lane-linear)
    const int fi = lane & 31, fk = lane >> 5;
    const int fsw = (fi >> 2) & 3;

    // ONE chip-wide tile queue here: per-XCD queues as in conv_wino24.hip were measured on these layers and lost 5 % end to end
    // (same-box A/B: forward kernels 19.6 -> 21.6 ms) -- most of them have one or two N tiles, so there is no panel to share and
    // the static eighths only unbalance the last round.
    //
    // The deep-ring forms (KSPLIT: launches of at most 512 units = tiles x parts, all resident at once) have NO queue: one unit per
    // workgroup, picked by blockIdx so that the workgroups of one XCD (blockIdx % 8, observed; only speed depends on it) take a
    // CONTIGUOUS range of units in (part, N tile, M tile) order -- an XCD then reads about an eighth of the weights and an eighth of
    // the input's k range instead of all of both.  One 544 x 544 image's 17 x 17 layers stream 19 MB of weights: drawn from a
    // chip-wide queue every XCD pulled all of them through its 4 MiB L2 (8 x 19 MB from the Infinity Cache per layer).
#if OM_SPLIT_TRACE
    // deep-ring forms (tools/deep_trace.py): start, first stage landed, k loop done, arrival counted, parts summed, stored
    unsigned long long dts[6] = {0, 0, 0, 0, 0, 0};
    auto deep_dump = [&]() {
        if (KSPLIT && tid == 0 && p.trace && blockIdx.x < 512)
            for (int i = 0; i < 6; ++i) p.trace[blockIdx.x * 8 + i] = dts[i];
    };
#endif
    for (int round = 0;; ++round) {
        int tile;
        int s_begin = 0, nsteps = p.ksteps;      // this unit's part of the k loop
        [[maybe_unused]] int part = 0;
        int tile_n, tile_m;
        if constexpr (KSPLIT) {
            if (round) break;
#if OM_SPLIT_TRACE
            SPLIT_STAMP(dts[0]);
#endif
            const int b = blockIdx.x, x = b & 7;
            int unit = b >> 3;
            for (int y = 0; y < x; ++y) unit += (p.total_tickets - y + 7) >> 3;      // units of the XCDs before this one
#ifdef OM_SPLIT_NO_XCD_PLACEMENT      // measurement build: units in launch order, i.e. round-robin over the XCDs
            unit = b;
#endif
            tile_m = unit % p.m_tiles;
            const int t = unit / p.m_tiles;
            tile_n = t % p.n_tiles;
            part = t / p.n_tiles;
            tile = tile_m * p.n_tiles + tile_n;
            s_begin = part * p.ksteps / p.ksplit;
            nsteps = (part + 1) * p.ksteps / p.ksplit - s_begin;
        } else {
            if (tid == 0) *s_ticket = atomicAdd(p.ticket, 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            tile = *s_ticket;
            if (tile >= p.total_tiles) break;
            tile = __builtin_amdgcn_readfirstlane(tile);
            tile_n = tile % p.n_tiles;
            tile_m = tile / p.n_tiles;
        }
        const int m0 = tile_m * BM, n0 = tile_n * BN;

        // ---- loader role (conv_igemm_f16.hip): per A row the byte offset of tap (0,0)'s chunk relative to the tile's first
        // image and a mask whose bit t says "tap t of this row is padding / beyond M"
        int rowoff[A_CH];
        unsigned invmask[A_CH];
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
            rowoff[j] = (((b - b_first) * p.H * p.W + iy0 * p.W + ix0) * p.in_pix_stride_h + scol * 8) * 2;
            unsigned badrow = 0, badcol = 0;
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
        const int row_halfs = p.taps * p.cin_h;
#pragma unroll
        for (int j = 0; j < B_CH; ++j) rowoffB[j] = ((n0 + lrow + 64 * j) * row_halfs + scol * 8) * 2;
        const _Float16* in_base = p.in + (size_t)b_first * p.H * p.W * p.in_pix_stride_h;
        const size_t in_left = ((size_t)p.total_in_pixels - (size_t)b_first * p.H * p.W) * p.in_pix_stride_h * 2;
        const int in_bytes = in_left < 0x7FFFFFFFull ? (int)in_left : 0x7FFFFFFF;

        int n_kh = 0, n_kw = 0, n_cc = 0;          // step being fetched: (tap row, tap col, 16-channel chunk)
        if constexpr (KSPLIT) {
            const int tap0 = s_begin / p.kc;
            n_cc = s_begin - tap0 * p.kc;
            n_kh = tap0 / p.ks;
            n_kw = tap0 - n_kh * p.ks;
        }
        auto advance = [&]() {
            if (++n_cc == p.kc) {
                n_cc = 0;
                if (++n_kw == p.ks) { n_kw = 0; ++n_kh; }
            }
        };
        auto issue_piece = [&](int piece, int buf, bool live) {
            f32x4* dst = smem + buf * STAGE + wave_u * 64;
            if (piece < A_CH) {
                const int j = piece;
                const auto rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(in_base), 0, live ? in_bytes : 0, 0x00020000);
                const int tap = n_kh * p.ks + n_kw;
                const int tap_off = ((n_kh * p.W + n_kw) * p.in_pix_stride_h + n_cc * 32) * 2;      // scalar
                const int voff = (rowoff[j] + tap_off) | ((invmask[j] << (31 - tap)) & 0x80000000u);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr_t)(dst + j * 256), 16, voff, 0, 0, 0);
            } else {
                const int j = piece - A_CH;
                const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.w), 0, live ? p.w_bytes : 0, 0x00020000);
                const int koff = ((n_kh * p.ks + n_kw) * p.cin_h + n_cc * 32) * 2;                  // scalar
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
        f32x4 xa0[TM], xa1[TM], nbh[TN], nbl[TN];      // raw operands of the NEXT step
        f16x8 ah[TM], al[TM], bh[TN], bl[TN];          // operands of the current step
        auto read_raw = [&](int buf) {
            const int c0 = fk ^ fsw, c1 = (2 + fk) ^ fsw;
            const int bo = buf * STAGE;
#pragma unroll
            for (int a = 0; a < TM; ++a) { xa0[a] = fragA[bo + a * 32 * 4 + c0]; xa1[a] = fragA[bo + a * 32 * 4 + c1]; }
#pragma unroll
            for (int b = 0; b < TN; ++b) { nbh[b] = fragB[bo + b * 32 * 4 + c0]; nbl[b] = fragB[bo + b * 32 * 4 + c1]; }
        };
        auto convert = [&]() {
#pragma unroll
            for (int a = 0; a < TM; ++a) split8(xa0[a], xa1[a], ah[a], al[a]);
#pragma unroll
            for (int b = 0; b < TN; ++b) { bh[b] = __builtin_bit_cast(f16x8, nbh[b]); bl[b] = __builtin_bit_cast(f16x8, nbl[b]); }
        };

        // prologue: steps 0 .. NBUF - 1 requested; step 0 waited for and converted
#pragma unroll
        for (int st = 0; st < NBUF; ++st) {
#pragma unroll
            for (int piece = 0; piece < NP; ++piece) issue_piece(piece, st, st < nsteps);
            advance();
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"i"((NBUF - 1) * NP) : "memory");
        __builtin_amdgcn_s_barrier();
#if OM_SPLIT_TRACE
        SPLIT_STAMP(dts[1]);
#endif
        read_raw(0);
        convert();
        int buf = 0;
        for (int s = 0; s < nsteps; ++s) {
            const int buf1 = buf == NBUF - 1 ? 0 : buf + 1;
            // step s+1 has landed (only the NBUF - 2 steps behind it may still fly); every wave's reads of `buf` are complete
            // (they fed its convert) -> `buf` can take step s + NBUF
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"i"((NBUF - 2) * NP) : "memory");
            __builtin_amdgcn_s_barrier();
            read_raw(buf1);
            const bool live3 = s + NBUF < nsteps;
#pragma unroll
            for (int piece = 0; piece < NP; ++piece) issue_piece(piece, buf, live3);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    // weights first: D[i = channel][j = pixel]
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[b], al[a], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[b], ah[a], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[b], ah[a], acc[a][b], 0, 0, 0);
                }
            convert();                               // operands of step s+1 (VALU under the matrix instructions above)
            buf = buf1;
            advance();
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();

#if OM_SPLIT_TRACE
        SPLIT_STAMP(dts[2]);
#endif
        if constexpr (KSPLIT) {
            if (p.ksplit > 1) {
                // publish this part's raw accumulators (write-through stores, as conv_wino24.hip's stream-K form), count the arrival
                const auto rs_part = __builtin_amdgcn_make_buffer_rsrc(p.partial, 0, 0x7FFFFFFF, 0x00020000);
                constexpr int PART_BYTES = TM * TN * 4 * 256 * 16;
                const int pbase = (tile * p.ksplit + part) * PART_BYTES + tid * 16;
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x4 v = {acc[a][b][4 * g], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]};
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v),
                                                                   rs_part, pbase, ((a * TN + b) * 4 + g) * 256 * 16, 16);
                        }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                // (write-through 16-byte stores, drained by every wave above, then ONE agent-scope atomic: no release fence -- a
                // buffer_wbl2 would also write back the previous layer's dirty lines, microseconds per part; the last arrival reads
                // with sc1 loads, which pass this CU's L1: no acquire either.  MI355X_MICROARCH.md, inter-workgroup visibility)
                if (tid == 0) s_ticket[1] = __hip_atomic_fetch_add(p.kflags + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __syncthreads();
#if OM_SPLIT_TRACE
                SPLIT_STAMP(dts[3]);
                if (s_ticket[1] != p.ksplit - 1) deep_dump();
#endif
                if (s_ticket[1] != p.ksplit - 1) break;         // another part stores the tile
                // the last arrival: every part has published; the sum runs in part order (this part's own copy included), so the
                // result does not depend on which part came last
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
                for (int q = 0; q < p.ksplit; ++q) {
                    const int qbase = (tile * p.ksplit + q) * PART_BYTES + tid * 16;
#pragma unroll
                    for (int a = 0; a < TM; ++a)
#pragma unroll
                        for (int b = 0; b < TN; ++b)
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_part, qbase, ((a * TN + b) * 4 + g) * 256 * 16, 16));
#pragma unroll
                                for (int k = 0; k < 4; ++k) acc[a][b][4 * g + k] += v[k];
                            }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }

#if OM_SPLIT_TRACE
        SPLIT_STAMP(dts[4]);
#endif
        float sc[8], sh[8];      // (four blocks per CU: no registers to hold them across the k loop)
        if constexpr (FAST) split_scale_shift<BN>(p, n0, tid, sc, sh);
        split_epilogue<BM, BN, WM, WN, FAST>(p, smem, acc, m0, n0, tid, wm, wn, fi, fk, sc, sh);
#if OM_SPLIT_TRACE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SPLIT_STAMP(dts[5]);
        deep_dump();
#endif
    }
}

// The same tile with WHOLE-LINE operand rows: a ring stage holds 128 bytes of every row -- two k-steps of 16 channels -- instead
// of 64.  A 16-channel fp32 chunk of a pixel (and 16 channels of a packed weight row) is half a 128-byte cache line, and the CU's
// vector-memory path is bound by the NUMBER of requests it can keep in flight (conv_wino14.hip / DESIGN.md 3.6): fetched 64 bytes
// per k-step every line is requested twice, one k-step apart.  LDS-DMA streams with twelve waves per CU: 37 B/clk/CU in half
// lines, 70 in whole lines (tools/scratch/ldsdma_rowbytes.hip).  Two stages of (BM + BN) x 128 B (64 KiB for 128 x 128: two
// workgroups per CU), one barrier per stage = per two k-steps, the stage after next requested in the middle of a stage (two
// k-steps of flight, as in the three-stage form).  Needs cin % 32 == 0 (every k-pair inside one kernel tap).
//
// GATHER: the A rows come from up to four tensors at their own resolutions (IgemmSParams::nseg; 1x1 layers only).  The row
// offsets are recomputed when the k loop crosses into the next segment (at most three times per tile); everything else is the
// same instruction stream, and the products are summed in the same order as over the materialised concat (bit-identical).
// NBUF > 2: the DEEP-RING form of the small launches (a batch of one or a few images; conv_igemm_split_kernel's KSPLIT form with
// whole-line rows): no tile queue -- one unit (tile x part of the k loop) per workgroup, placed XCD-contiguously --, NBUF stages of
// two k-steps in the ring, split-K parts summed by the last arrival.  The 64-byte-row form of those launches ran its k loop at
// ~30 B/clk/CU, the half-line request rate (tools/deep_trace.py: 545 cycles per 8 KB k-step, two workgroups per CU).
template <int BM, int BN, int WM, int WN, bool GATHER = false, bool FAST = false, int NBUF = 2>
__global__ __launch_bounds__(256, 2) void conv_igemm_split_wide_kernel(const IgemmSParams p) {
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int NWN = BN / WN;
    constexpr int A_CH = BM / 32, B_CH = BN / 32, NP = A_CH + B_CH;     // 32 rows x 128 B per workgroup-wide piece
    constexpr bool KSPLIT = NBUF > 2;
    static_assert(!(KSPLIT && GATHER), "the deep-ring form has no gathered input");
    constexpr int STAGE = (BM + BN) * 8;            // f32x4 (16-byte) units per ring stage
    static_assert((BM / WM) * (BN / WN) == 4, "four waves per workgroup");
    static_assert(BM % 32 == 0 && BN % 32 == 0, "whole 32-row pieces");
    static_assert(WM * BN / 4 <= NBUF * STAGE, "one wave-row of the fp32 C tile must fit in the operand ring");
    __shared__ f32x4 smem[NBUF * STAGE + 1];      // ONE LDS object (see conv_igemm.hip)
    int* const s_ticket = reinterpret_cast<int*>(smem + NBUF * STAGE);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave / NWN, wn = wave % NWN;
    const int lrow = tid >> 3, lcol = tid & 7;     // loader: row within a 32-row piece, 16-byte position in the 128-byte row
    const int scol = lcol ^ ((lrow >> 1) & 7);     // logical chunk this lane fetches (the LDS image stays lane-linear)
    const int fi = lane & 31, fk = lane >> 5;
    const int fsw = (fi >> 1) & 7;
#if OM_SPLIT_TRACE
    int n_traced = 0;
#endif

    // The tile queue's NEXT ticket is drawn while this tile's k loop runs and handed over through LDS before the epilogue (its
    // value must not be consumed behind the epilogue's stores: loads, atomics and stores retire through one in-order counter).
    if constexpr (!KSPLIT) {
        if (tid == 0) *s_ticket = atomicAdd(p.ticket, 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
#if OM_SPLIT_TRACE
    // deep-ring forms (tools/deep_trace.py): start, first stage landed, k loop done, arrival counted, parts summed, stored
    unsigned long long dts[6] = {0, 0, 0, 0, 0, 0};
    auto deep_dump = [&]() {
        if (KSPLIT && tid == 0 && p.trace && blockIdx.x < 512)
            for (int i = 0; i < 6; ++i) p.trace[blockIdx.x * 8 + i] = dts[i];
    };
#endif
    for (int round = 0;; ++round) {
#if OM_SPLIT_TRACE
        unsigned long long ts0, ts1, ts2, ts3, twait = 0, wa, wb;
        SPLIT_STAMP(ts0);
#endif
        int tile, tile_n, tile_m;
        [[maybe_unused]] int part = 0;
        if constexpr (KSPLIT) {
            if (round) break;
#if OM_SPLIT_TRACE
            SPLIT_STAMP(dts[0]);
#endif
            const int b = blockIdx.x, x = b & 7;      // (conv_igemm_split_kernel: XCD-contiguous units in (part, N tile, M tile) order)
            int unit = b >> 3;
            for (int y = 0; y < x; ++y) unit += (p.total_tickets - y + 7) >> 3;
            tile_m = unit % p.m_tiles;
            const int t = unit / p.m_tiles;
            tile_n = t % p.n_tiles;
            part = t / p.n_tiles;
            tile = tile_m * p.n_tiles + tile_n;
        } else {
            tile = *s_ticket;
            if (tile >= p.total_tiles) break;
            tile = __builtin_amdgcn_readfirstlane(tile);
            tile_n = tile % p.n_tiles;
            tile_m = tile / p.n_tiles;
        }
        const int m0 = tile_m * BM, n0 = tile_n * BN;

        int rowoff[A_CH];
        unsigned invmask[A_CH];
        [[maybe_unused]] int g_img[A_CH], g_y[A_CH], g_x[A_CH];      // GATHER: the row's pixel, for the segments' own resolutions
        const int b_first = (m0 < p.M ? m0 : p.M - 1) / p.HoWo;
#pragma unroll
        for (int j = 0; j < A_CH; ++j) {
            int m = m0 + lrow + 32 * j;
            const bool mok = m < p.M;
            if (!mok) m = p.M - 1;
            const int b = m / p.HoWo;
            const int rr = m - b * p.HoWo;
            const int oy = rr / p.Wo;
            const int ox = rr - oy * p.Wo;
            const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
            rowoff[j] = (((b - b_first) * p.H * p.W + iy0 * p.W + ix0) * p.in_pix_stride_h + scol * 8) * 2;
            if constexpr (GATHER) { g_img[j] = b - b_first; g_y[j] = oy; g_x[j] = ox; }
            unsigned badrow = 0, badcol = 0;
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
        const int row_halfs = p.taps * p.cin_h;
#pragma unroll
        for (int j = 0; j < B_CH; ++j) rowoffB[j] = ((n0 + lrow + 32 * j) * row_halfs + scol * 8) * 2;
        const _Float16* in_base = p.in + (size_t)b_first * p.H * p.W * p.in_pix_stride_h;
        const size_t in_left = ((size_t)p.total_in_pixels - (size_t)b_first * p.H * p.W) * p.in_pix_stride_h * 2;
        int in_bytes = in_left < 0x7FFFFFFFull ? (int)in_left : 0x7FFFFFFF;
        [[maybe_unused]] int seg = 0, seg_c0 = 0, seg_c1 = 0x7FFFFFFF;      // GATHER: current segment, its 32-channel chunk range
        [[maybe_unused]] auto set_segment = [&](int g) {
            // (selects, not p.seg_x[g]: a run-time index into the by-value parameter block would copy the arrays to scratch)
            auto pick = [g](const auto (&v)[4]) { return g == 0 ? v[0] : g == 1 ? v[1] : g == 2 ? v[2] : v[3]; };
            const int sh = pick(p.seg_shift), Hs = p.Ho >> sh, Ws = p.Wo >> sh, st = pick(p.seg_stride_h);
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

        const int kc2 = p.kc >> 1;                  // pairs of 16-channel chunks per tap
        int nstages = p.taps * kc2;
        int n_kh = 0, n_kw = 0, n_cc = 0;          // stage being fetched: (tap row, tap col, 32-channel chunk)
        if constexpr (KSPLIT) {                     // this part's stages
            const int s_begin = part * nstages / p.ksplit;
            nstages = (part + 1) * nstages / p.ksplit - s_begin;
            const int tap0 = s_begin / kc2;
            n_cc = s_begin - tap0 * kc2;
            n_kh = tap0 / p.ks;
            n_kw = tap0 - n_kh * p.ks;
        }
        auto advance = [&]() {
            if (++n_cc == kc2) {
                n_cc = 0;
                if (++n_kw == p.ks) { n_kw = 0; ++n_kh; }
            }
        };
        auto issue_piece = [&](int piece, int buf, bool live) {
            f32x4* dst = smem + buf * STAGE + wave_u * 64;
            if (piece < A_CH) {
                const int j = piece;
                if constexpr (GATHER) {
                    if (piece == 0 && live && n_cc >= seg_c1) set_segment(seg + 1);      // uniform; the stages walk the channels in order
                }
                const auto rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(in_base), 0, live ? in_bytes : 0, 0x00020000);
                const int tap = n_kh * p.ks + n_kw;
                const int tap_off = GATHER ? (n_cc - seg_c0) * 128      // 1x1: one tap; the chunk within its segment
                                           : ((n_kh * p.W + n_kw) * p.in_pix_stride_h + n_cc * 64) * 2;      // scalar
                const int voff = (rowoff[j] + tap_off) | ((invmask[j] << (31 - tap)) & 0x80000000u);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr_t)(dst + j * 256), 16, voff, 0, 0, 0);
            } else {
                const int j = piece - A_CH;
                const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.w), 0, live ? p.w_bytes : 0, 0x00020000);
                const int koff = ((n_kh * p.ks + n_kw) * p.cin_h + n_cc * 64) * 2;                  // scalar
                const int vo = rowoffB[j] + koff;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(dst + BM * 8 + j * 256), 16, vo, 0, 0, 0);
            }
        };

        float sc[8], sh[8];
        if constexpr (FAST) split_scale_shift<BN>(p, n0, tid, sc, sh);
        f32x16 acc[TM][TN];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

        const f32x4* fragA = smem + (wm * WM + fi) * 8;
        const f32x4* fragB = smem + BM * 8 + (wn * WN + fi) * 8;
        f32x4 xa0[TM], xa1[TM], nbh[TN], nbl[TN];      // raw operands of the NEXT k-step
        f16x8 ah[TM], al[TM], bh[TN], bl[TN];          // operands of the current k-step
        auto read_raw = [&](int buf, int sub) {
            const int c0 = (4 * sub + fk) ^ fsw, c1 = (4 * sub + 2 + fk) ^ fsw;
            const int bo = buf * STAGE;
#pragma unroll
            for (int a = 0; a < TM; ++a) { xa0[a] = fragA[bo + a * 32 * 8 + c0]; xa1[a] = fragA[bo + a * 32 * 8 + c1]; }
#pragma unroll
            for (int b = 0; b < TN; ++b) { nbh[b] = fragB[bo + b * 32 * 8 + c0]; nbl[b] = fragB[bo + b * 32 * 8 + c1]; }
        };
        auto convert = [&]() {
#pragma unroll
            for (int a = 0; a < TM; ++a) split8(xa0[a], xa1[a], ah[a], al[a]);
#pragma unroll
            for (int b = 0; b < TN; ++b) { bh[b] = __builtin_bit_cast(f16x8, nbh[b]); bl[b] = __builtin_bit_cast(f16x8, nbl[b]); }
        };
        auto multiply = [&]() {
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    // weights first: D[i = channel][j = pixel]; the order of conv_igemm_split_kernel
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[b], al[a], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[b], ah[a], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[b], ah[a], acc[a][b], 0, 0, 0);
                }
        };

        // prologue: stages 0 .. NBUF - 1 requested; stage 0 waited for, its first k-step converted
#pragma unroll
        for (int st = 0; st < NBUF; ++st) {
#pragma unroll
            for (int piece = 0; piece < NP; ++piece) issue_piece(piece, st, st < nstages);
            advance();
        }
        [[maybe_unused]] int next_ticket = 0;
        if constexpr (!KSPLIT) {
            if (tid == 0) next_ticket = atomicAdd(p.ticket, 1);      // newer than stage 1's pieces: the wait below still covers stage 0
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"i"((NBUF - 1) * NP) : "memory");
        __builtin_amdgcn_s_barrier();
#if OM_SPLIT_TRACE
        SPLIT_STAMP(ts1);
#endif
#if OM_SPLIT_TRACE
        SPLIT_STAMP(dts[1]);
#endif
        read_raw(0, 0);
        convert();
        int buf = 0;
        for (int s = 0; s < nstages; ++s) {
            read_raw(buf, 1);
            multiply();                              // k-step (s, 0)
            convert();                               // operands of (s, 1); my reads of `buf` are complete
            // stage s + 1 has landed (only the NBUF - 2 stages behind it may still fly); every wave is done reading `buf`: it takes
            // stage s + NBUF
#if OM_SPLIT_TRACE
            SPLIT_STAMP(wa);
#endif
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"i"((NBUF - 2) * NP) : "memory");
            __builtin_amdgcn_s_barrier();
#if OM_SPLIT_TRACE
            SPLIT_STAMP(wb);
            twait += wb - wa;
#endif
            const bool live2 = s + NBUF < nstages;
#pragma unroll
            for (int piece = 0; piece < NP; ++piece) issue_piece(piece, buf, live2);
            advance();
            const int buf1 = buf == NBUF - 1 ? 0 : buf + 1;
            read_raw(buf1, 0);
            multiply();                              // k-step (s, 1)
            convert();                               // operands of (s + 1, 0)
            buf = buf1;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if constexpr (!KSPLIT) {
            if (tid == 0) *s_ticket = next_ticket;      // every wave read the current ticket many barriers ago
        }
        __syncthreads();
#if OM_SPLIT_TRACE
        SPLIT_STAMP(dts[2]);
#endif
        if constexpr (KSPLIT) {
            if (p.ksplit > 1) {
                // publish / count / sum in part order: conv_igemm_split_kernel's split-K hand-off (write-through stores, drained
                // waves, one agent-scope atomic, sc1 loads; the parts are whole STAGES here)
                const auto rs_part = __builtin_amdgcn_make_buffer_rsrc(p.partial, 0, 0x7FFFFFFF, 0x00020000);
                constexpr int PART_BYTES = TM * TN * 4 * 256 * 16;
                const int pbase = (tile * p.ksplit + part) * PART_BYTES + tid * 16;
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x4 v = {acc[a][b][4 * g], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]};
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v),
                                                                   rs_part, pbase, ((a * TN + b) * 4 + g) * 256 * 16, 16);
                        }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) s_ticket[1] = __hip_atomic_fetch_add(p.kflags + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __syncthreads();
#if OM_SPLIT_TRACE
                SPLIT_STAMP(dts[3]);
                if (s_ticket[1] != p.ksplit - 1) deep_dump();
#endif
                if (s_ticket[1] != p.ksplit - 1) break;         // another part stores the tile
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
                for (int q = 0; q < p.ksplit; ++q) {
                    const int qbase = (tile * p.ksplit + q) * PART_BYTES + tid * 16;
#pragma unroll
                    for (int a = 0; a < TM; ++a)
#pragma unroll
                        for (int b = 0; b < TN; ++b)
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_part, qbase, ((a * TN + b) * 4 + g) * 256 * 16, 16));
#pragma unroll
                                for (int kk = 0; kk < 4; ++kk) acc[a][b][4 * g + kk] += v[kk];
                            }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
#if OM_SPLIT_TRACE
        SPLIT_STAMP(ts2);
#endif

#if OM_SPLIT_TRACE
        SPLIT_STAMP(dts[4]);
#endif
        split_epilogue<BM, BN, WM, WN, FAST>(p, smem, acc, m0, n0, tid, wm, wn, fi, fk, sc, sh);
#if OM_SPLIT_TRACE
        if constexpr (KSPLIT) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            SPLIT_STAMP(dts[5]);
            deep_dump();
        }
        SPLIT_STAMP(ts3);
        if (p.trace && blockIdx.x < 16 && n_traced < 16 && tid == 0) {
            unsigned long long* t = p.trace + ((size_t)blockIdx.x * 16 + n_traced) * 8;
            t[0] = ts0; t[1] = ts1; t[2] = ts2; t[3] = ts3; t[4] = twait; t[5] = (unsigned long long)tile;
        }
        ++n_traced;
#endif
    }
}

// NBUF: ring stages of the 64-byte-row kernel (> 3: its deep-ring form); WSTAGES: ring stages of the whole-line kernel's deep-ring
// form (0: the two-stage form with a tile queue)
template <int BM, int BN, int WM, int WN, bool WIDE = false, bool GATHER = false, int NBUF = 3, int WSTAGES = 0>
static int launch_tile_split(IgemmSParams p, int cout_pad, int blocks_per_cu, hipStream_t stream) {
    const int m_tiles = (p.M + BM - 1) / BM;
    p.n_tiles = cout_pad / BN;
    const long long total = (long long)m_tiles * p.n_tiles;
    OM_REQUIRE(total > 0 && total < (1ll << 31), OM_EINVAL, "conv split: %lld tiles out of range", total);
    p.total_tiles = (int)total;
    // split-K (deep-ring forms; p.ksplit arrives as the caller's upper bound): as many parts as two workgroups per CU take in one
    // round, at least sixteen k-steps each (bs = 1, per layer: 1x1 layers of 16 / 32 / 64 k-steps are fastest in 1 / 2 / 4 parts, a
    // part costing ~2 us of prologue and its share of the last arrival's sum; profiles/r04_experiments.md section 9)
    constexpr bool DEEP = WIDE ? WSTAGES > 2 : NBUF > 3;      // the small launches' forms: one unit per workgroup, split-K
    int parts = 1;
    if constexpr (DEEP) {
        if (p.ksplit > 1 && p.partial && total <= SK_SLOTS) {
            parts = p.ksplit;
            if (parts > 512 / (int)total) parts = 512 / (int)total;
            if (parts > p.ksteps / 16) parts = p.ksteps / 16;
            if (parts > 8) parts = 8;
            if (parts < 1) parts = 1;
        }
    }
    p.ksplit = parts;
    p.total_tickets = p.total_tiles * parts;
    p.m_tiles = m_tiles;
    const long long tickets = total * parts;
    long long grid = tickets < 256ll * blocks_per_cu ? tickets : 256ll * blocks_per_cu;
    if constexpr (DEEP) {      // one unit per workgroup, no queue
        OM_REQUIRE(tickets <= 512, OM_EINVAL, "conv split: %lld units in a deep-ring launch", tickets);
        grid = tickets;
    }
    // the epilogue without loads in its row sweeps (split_epilogue: FAST) wherever the layer allows it
    const bool fast = p.out_mode == 0 && !p.res && p.vec_io && p.cout == cout_pad;
    if constexpr (WIDE) {
        constexpr int WNBUF = DEEP ? WSTAGES : 2;
        if (fast) hipLaunchKernelGGL((conv_igemm_split_wide_kernel<BM, BN, WM, WN, GATHER, true, WNBUF>), dim3((unsigned)grid), dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((conv_igemm_split_wide_kernel<BM, BN, WM, WN, GATHER, false, WNBUF>), dim3((unsigned)grid), dim3(256), 0, stream, p);
    } else {
        if (fast) hipLaunchKernelGGL((conv_igemm_split_kernel<BM, BN, WM, WN, true, NBUF>), dim3((unsigned)grid), dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((conv_igemm_split_kernel<BM, BN, WM, WN, false, NBUF>), dim3((unsigned)grid), dim3(256), 0, stream, p);
    }
    OM_CHECK_HIP(hipGetLastError());
    return OM_OK;
}

void conv_tile_for_split(int M, int cout_pad, int* bm, int* bn) {
    // time over all tiles ~ tiles x tile area / how well the shape feeds the pipe (tools/split_tile_sweep.py on the forward's
    // layer shapes: 128 x 128 is the fastest wherever cout allows it, 256 x 128 -- two workgroups per CU, 256 registers -- 9 %
    // behind, then 128 x 64, 64 x 64, 128 x 32)
    struct Cand { int bm, bn; double eff; };
    const Cand cands[] = {{128, 128, 1.00}, {256, 128, 0.91}, {128, 64, 0.82}, {64, 64, 0.69}, {128, 32, 0.57}};
    double best = 1e300;
    for (const Cand& c : cands) {
        if (cout_pad % c.bn) continue;
        const double cost = (double)((M + c.bm - 1) / c.bm) * (cout_pad / c.bn) * c.bm * c.bn / c.eff;
        if (cost < best) { best = cost; *bm = c.bm; *bn = c.bn; }
    }
#ifndef OM_SPLIT_NO_LATENCY_TILES
    // Few tiles (a batch of one or a few images: /root/reference/infer.py:143-172 runs bs = 1): a launch is then ONE round and
    // its time is a tile's latency, which is k-steps x the time of a k-step -- 12 matrix instructions per SIMD for 128 x 128, 3 for
    // 64 x 64.  While the chosen shape leaves more than half of the CUs without a tile, take the next smaller one.  (An output
    // element's sum does not depend on the tile shape: same products, same order.)
    const Cand small[] = {{128, 64, 0}, {64, 64, 0}};
    for (const Cand& c : small) {
        const long long tiles = (long long)((M + *bm - 1) / *bm) * (cout_pad / *bn);
        if (tiles > 128 || cout_pad % c.bn || c.bm * c.bn >= *bm * *bn) continue;
        *bm = c.bm; *bn = c.bn;
    }
#endif
}

// a.w: packed hi/lo weights (include/orienmask_hip.h: om_layer_info.wsplit_off); a.scale: scale * 2^-e
int launch_conv_igemm_split(const ConvArgs& a, hipStream_t stream) {
    OM_REQUIRE((a.in || a.nseg > 0) && a.w && a.scale && a.shift && a.out, OM_EINVAL, "conv split: null pointer");
    OM_REQUIRE(a.cin % 16 == 0 && a.cin >= 16, OM_EINVAL, "conv split: cin=%d must be a multiple of 16", a.cin);
    OM_REQUIRE(a.ks == 1 || a.ks == 3, OM_EINVAL, "conv split: ksize=%d not supported", a.ks);
    OM_REQUIRE(a.stride == 1 || a.stride == 2, OM_EINVAL, "conv split: stride=%d not supported", a.stride);
    OM_REQUIRE((a.nseg > 0 || (a.in_pix_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(a.in) & 15) == 0)) &&
                   (reinterpret_cast<uintptr_t>(a.w) & 15) == 0,
               OM_EINVAL, "conv split: input view / weights must be 16-byte aligned");
    OM_REQUIRE(a.cout_pad % 32 == 0 && a.cout <= a.cout_pad, OM_EINVAL, "conv split: cout_pad=%d", a.cout_pad);
    OM_REQUIRE((long long)a.B * a.H * a.W * a.in_pix_stride < (1ll << 40) && (long long)a.B * a.H * a.W < (1ll << 31) &&
                   (long long)a.B * a.Ho * a.Wo < (1ll << 31),
               OM_EINVAL, "conv split: problem too large");
    OM_REQUIRE(!(a.res && a.out_mode != 0), OM_EINVAL, "conv split: residual only with plain NHWC output");
    OM_REQUIRE(a.ticket, OM_EINVAL, "conv split: the tile queue needs a zeroed ticket word");
    IgemmSParams p;
    p.in = reinterpret_cast<const _Float16*>(a.in); p.w = reinterpret_cast<const _Float16*>(a.w);
    p.scale = a.scale; p.shift = a.shift; p.res = a.res; p.out = a.out; p.ticket = a.ticket; p.status = a.status;
    p.H = a.H; p.W = a.W; p.cin_h = 2 * a.cin; p.in_pix_stride_h = 2 * a.in_pix_stride;
    p.Ho = a.Ho; p.Wo = a.Wo; p.HoWo = a.Ho * a.Wo; p.cout = a.cout;
    p.ks = a.ks; p.stride = a.stride; p.pad = a.ks / 2;
    p.M = a.B * a.Ho * a.Wo; p.taps = a.ks * a.ks;
    p.kc = a.cin / 16;
    p.ksteps = p.taps * p.kc;
    p.leaky = a.leaky; p.res_pix_stride = a.res_pix_stride; p.out_pix_stride = a.out_pix_stride;
    p.out_mode = a.out_mode; p.up = a.up;
    p.n_tiles = 0; p.total_tiles = 0;
    p.partial = a.sk_partial; p.kflags = a.ticket + SK_FLAG_OFF; p.ksplit = a.ksplit_max; p.total_tickets = 0;
    p.total_in_pixels = a.B * a.H * a.W;
    p.w_bytes = a.cout_pad * p.taps * a.cin * 4;
    p.vec_io = (a.out_mode != 2 && a.out_pix_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0 &&
                (!a.res || (a.res_pix_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(a.res) & 15) == 0)))
                   ? 1 : 0;
    p.nseg = 0; p.nimg = a.B;
#if OM_SPLIT_TRACE
    p.trace = g_split_trace;
#endif
    for (int g = 0; g < 4; ++g) { p.seg_ptr[g] = p.in; p.seg_stride_h[g] = 0; p.seg_shift[g] = 0; p.seg_end[g] = 0x7FFFFFFF; }
    if (a.nseg > 0) {
        // gathered input: 1x1, whole 32-channel chunks per segment, every segment's resolution a power-of-two fraction of this one
        OM_REQUIRE(a.nseg <= 4 && a.ks == 1 && a.stride == 1 && a.cout_pad % 128 == 0, OM_EINVAL,
                   "conv split: a gathered input needs a 1x1 stride-1 layer with cout_pad %% 128 == 0 and at most 4 segments (nseg=%d ks=%d "
                   "stride=%d cout_pad=%d)", a.nseg, a.ks, a.stride, a.cout_pad);
        int end = 0;
        for (int g = 0; g < a.nseg; ++g) {
            const int up = a.seg_up[g];
            int sh = 0;
            while ((1 << sh) < up) ++sh;
            OM_REQUIRE(a.seg_ptr[g] && up >= 1 && (1 << sh) == up && a.H % up == 0 && a.W % up == 0 && a.seg_channels[g] > 0 &&
                           a.seg_channels[g] % 32 == 0 && a.seg_pix_stride[g] % 4 == 0 && a.seg_pix_stride[g] >= a.seg_channels[g] &&
                           (reinterpret_cast<uintptr_t>(a.seg_ptr[g]) & 15) == 0,
                       OM_EINVAL, "conv split: segment %d (channels=%d pix_stride=%d up=%d) of a gathered input", g, a.seg_channels[g],
                       a.seg_pix_stride[g], up);
            end += a.seg_channels[g] / 32;
            p.seg_ptr[g] = reinterpret_cast<const _Float16*>(a.seg_ptr[g]);
            p.seg_stride_h[g] = 2 * a.seg_pix_stride[g]; p.seg_shift[g] = sh; p.seg_end[g] = end;
        }
        OM_REQUIRE(end * 32 == a.cin, OM_EINVAL, "conv split: the segments hold %d channels, the layer reads %d", end * 32, a.cin);
        p.nseg = a.nseg;
        p.in = p.seg_ptr[0];
        return launch_tile_split<128, 128, 64, 64, true, true>(p, a.cout_pad, 2, stream);
    }
    int bm, bn;
    conv_tile_for_split(p.M, a.cout_pad, &bm, &bn);
    if (a.force_bm || a.force_bn) {      // unit-test entry: this call's tile shape
        const bool built = (a.force_bm == 256 && a.force_bn == 128) || (a.force_bm == 128 && (a.force_bn == 128 || a.force_bn == 64 || a.force_bn == 32)) ||
                           (a.force_bm == 64 && a.force_bn == 64);
        OM_REQUIRE(built && a.cout_pad % a.force_bn == 0, OM_EINVAL, "conv split: %d x %d is not a built tile shape for cout_pad=%d",
                   a.force_bm, a.force_bn, a.cout_pad);
        bm = a.force_bm; bn = a.force_bn;
    }
    if (bm == 256 && bn == 128) return launch_tile_split<256, 128, 128, 64>(p, a.cout_pad, 2, stream);
    if (bm == 128 && bn == 128 && a.cin % 32 == 0 && !a.force_bm && OM_SPLIT_WIDE) return launch_tile_split<128, 128, 64, 64, true>(p, a.cout_pad, 2, stream);
    if (bm == 128 && bn == 128) return launch_tile_split<128, 128, 64, 64>(p, a.cout_pad, 3, stream);
    // the latency form (deep ring, conv_igemm_split_kernel's NBUF) for launches that cannot fill the chip anyway
    const long long ntile = (long long)((p.M + bm - 1) / bm) * (a.cout_pad / bn);
    const bool deep = ntile <= 256 && (!a.force_bm || a.ksplit_max >= 1);      // (om_conv2d_split_k: a forced shape in its deep-ring form)
#ifndef OM_DEEP_WIDE
#define OM_DEEP_WIDE 1          // whole-line rows in the deep-ring forms where cin % 32 == 0
#endif
    if (OM_DEEP_WIDE && deep && a.cin % 32 == 0) {      // three stages of 24 KiB / four of 16 KiB: two workgroups per CU
        if (bm == 128 && bn == 64) return launch_tile_split<128, 64, 64, 32, true, false, 3, 3>(p, a.cout_pad, 2, stream);
        if (bm == 64 && bn == 64) return launch_tile_split<64, 64, 32, 32, true, false, 3, 4>(p, a.cout_pad, 2, stream);
    }
    if (bm == 128 && bn == 64 && deep) return launch_tile_split<128, 64, 64, 32, false, false, 5>(p, a.cout_pad, 2, stream);
    if (bm == 64 && bn == 64 && deep) return launch_tile_split<64, 64, 32, 32, false, false, 8>(p, a.cout_pad, 2, stream);
#ifndef OM_SPLIT_WIDE_SMALLN
#define OM_SPLIT_WIDE_SMALLN 0
#endif
#if OM_SPLIT_WIDE_SMALLN
    if (bm == 128 && bn == 64 && a.cin % 32 == 0 && !a.force_bm) return launch_tile_split<128, 64, 64, 32, true>(p, a.cout_pad, OM_SPLIT_WIDE_SMALLN, stream);
    if (bm == 128 && bn == 32 && a.cin % 32 == 0 && !a.force_bm) return launch_tile_split<128, 32, 32, 32, true>(p, a.cout_pad, OM_SPLIT_WIDE_SMALLN + 1, stream);
#endif
    if (bm == 128 && bn == 64) return launch_tile_split<128, 64, 64, 32>(p, a.cout_pad, 4, stream);
    if (bm == 64 && bn == 64) return launch_tile_split<64, 64, 32, 32>(p, a.cout_pad, 4, stream);
    return launch_tile_split<128, 32, 32, 32>(p, a.cout_pad, 4, stream);
}

}  // namespace om
