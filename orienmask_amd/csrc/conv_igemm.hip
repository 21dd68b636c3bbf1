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
//     goes to which slot is a free permutation as long as A and B agree.)
//   * K loop = taps x cin/32.  Register-staged double buffer: the loads of step s+1 are issued
//     before the 16*TM*TN MFMAs of step s (>= 4096 matrix-pipe cycles per wave: HBM/L2 latency is
//     hidden), written to the other LDS buffer after them; one barrier per step.
//   * LDS rows are 128 B (32 floats); the 16-B chunk index is XOR-swizzled with (row>>1)&7 so
//     that the 16 rows a ds_read_b128 lane group touches land on 16 distinct bank quads.
//   * 1-D grid remapped so that each XCD (private 4 MiB L2) owns a contiguous run of tiles and
//     the N tiles that share an A panel run back-to-back on the same XCD.
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
    int H, W, cin, in_pix_stride;
    int Ho, Wo, HoWo, cout;
    int ks, stride, pad;
    int M, kc, ksteps, taps;
    int n_tiles;
    int leaky, res_pix_stride, out_pix_stride, out_mode, up;
};

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_igemm_f32_kernel(const IgemmParams p) {
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int NWN = BN / WN;
    constexpr int A_CH = BM / 32;   // 16-byte chunks of the A tile each thread stages
    constexpr int B_CH = BN / 32;
    static_assert((BM / WM) * (BN / WN) == 4, "four waves per workgroup");
    __shared__ f32x4 smem[2 * (BM + BN) * 8];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / NWN, wn = wave % NWN;

    // XCD-aware, bijective remap of the workgroup id (block b runs on XCD b % 8).
    int L;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, i = bid >> 3;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
    }
    const int tile_n = L % p.n_tiles;
    const int tile_m = L / p.n_tiles;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- loader role: thread -> (row lrow + 32*j, 16-byte chunk lcol) --------------------
    const int lrow = tid >> 3, lcol = tid & 7;
    const int lsw = lcol ^ ((lrow >> 1) & 7);
    int pixbase[A_CH], iy0[A_CH], ix0[A_CH];
    unsigned mokmask = 0;
#pragma unroll
    for (int j = 0; j < A_CH; ++j) {
        int m = m0 + lrow + 32 * j;
        const bool ok = m < p.M;
        if (!ok) m = p.M - 1;
        const int b = m / p.HoWo;
        const int rr = m - b * p.HoWo;
        const int oy = rr / p.Wo;
        const int ox = rr - oy * p.Wo;
        pixbase[j] = b * p.H * p.W;
        iy0[j] = oy * p.stride - p.pad;
        ix0[j] = ox * p.stride - p.pad;
        mokmask |= (ok ? 1u : 0u) << j;
    }
    const float* wrow[B_CH];
#pragma unroll
    for (int j = 0; j < B_CH; ++j)
        wrow[j] = p.w + (size_t)(n0 + lrow + 32 * j) * p.taps * p.cin + lcol * 4;

    f32x4 ra[A_CH], rb[B_CH];
    unsigned okmask = 0;

    auto load_step = [&](int s) {
        const int tap = s / p.kc;
        const int cc = s - tap * p.kc;
        const int kh = tap / p.ks;
        const int kw = tap - kh * p.ks;
        const int coff = cc * 32 + lcol * 4;
        okmask = 0;
#pragma unroll
        for (int j = 0; j < A_CH; ++j) {
            const int iy = iy0[j] + kh, ix = ix0[j] + kw;
            const bool ok = ((mokmask >> j) & 1u) && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const int pix = ok ? pixbase[j] + iy * p.W + ix : pixbase[j];
            ra[j] = *reinterpret_cast<const f32x4*>(p.in + (size_t)pix * p.in_pix_stride + coff);
            okmask |= (ok ? 1u : 0u) << j;
        }
        const int woff = tap * p.cin + cc * 32;
#pragma unroll
        for (int j = 0; j < B_CH; ++j) rb[j] = *reinterpret_cast<const f32x4*>(wrow[j] + woff);
    };

    auto store_step = [&](int buf) {
        f32x4* sA = smem + buf * (BM + BN) * 8;
        f32x4* sB = sA + BM * 8;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < A_CH; ++j) sA[(lrow + 32 * j) * 8 + lsw] = ((okmask >> j) & 1u) ? ra[j] : zero;
#pragma unroll
        for (int j = 0; j < B_CH; ++j) sB[(lrow + 32 * j) * 8 + lsw] = rb[j];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int fi = lane & 31, fk = lane >> 5;
    const int fsw = (fi >> 1) & 7;

    auto compute = [&](int buf) {
        const f32x4* sA = smem + buf * (BM + BN) * 8 + (wm * WM + fi) * 8;
        const f32x4* sB = smem + buf * (BM + BN) * 8 + BM * 8 + (wn * WN + fi) * 8;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ch = (2 * q + fk) ^ fsw;
            f32x4 fa[TM], fb[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) fa[a] = sA[a * 32 * 8 + ch];
#pragma unroll
            for (int b = 0; b < TN; ++b) fb[b] = sB[b * 32 * 8 + ch];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a][t], fb[b][t], acc[a][b], 0, 0, 0);
        }
    };

    load_step(0);
    store_step(0);
    __syncthreads();
    for (int s = 0; s < p.ksteps; ++s) {
        const bool more = s + 1 < p.ksteps;
        if (more) load_step(s + 1);
        compute(s & 1);
        if (more) store_step((s + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int colw = lane & 31, rowq = (lane >> 5) * 4;
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int n = n0 + wn * WN + b * 32 + colw;
        const bool nok = n < p.cout;
        const float sc = p.scale[n], sh = p.shift[n];
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            const int mb = m0 + wm * WM + a * 32 + rowq;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mb + (r & 3) + 8 * (r >> 2);
                if (m < p.M && nok) {
                    float v = fmaf(acc[a][b][r], sc, sh);
                    if (p.leaky) v = v > 0.f ? v : v * 0.1f;
                    if (p.out_mode == 0) {
                        if (p.res) v += p.res[(size_t)m * p.res_pix_stride + n];
                        p.out[(size_t)m * p.out_pix_stride + n] = v;
                    } else {
                        const int bi = m / p.HoWo;
                        const int rr = m - bi * p.HoWo;
                        if (p.out_mode == 2) {
                            p.out[((size_t)bi * p.cout + n) * p.HoWo + rr] = v;
                        } else {
                            const int oy = rr / p.Wo, ox = rr - oy * p.Wo;
                            const int Wu = p.Wo * p.up;
                            const size_t base = ((size_t)bi * p.Ho * p.up + (size_t)oy * p.up) * Wu + (size_t)ox * p.up;
                            for (int dy = 0; dy < p.up; ++dy)
                                for (int dx = 0; dx < p.up; ++dx)
                                    p.out[(base + (size_t)dy * Wu + dx) * p.out_pix_stride + n] = v;
                        }
                    }
                }
            }
        }
    }
}

template <int BM, int BN, int WM, int WN>
static int launch_cfg(const IgemmParams& p0, hipStream_t stream) {
    IgemmParams p = p0;
    const int m_tiles = (p.M + BM - 1) / BM;
    const long long nblk = (long long)m_tiles * p.n_tiles;
    OM_REQUIRE(nblk > 0 && nblk < (1ll << 31), OM_EINVAL, "conv: grid of %lld workgroups out of range", nblk);
    hipLaunchKernelGGL((conv_igemm_f32_kernel<BM, BN, WM, WN>), dim3((unsigned)nblk), dim3(256), 0, stream, p);
    OM_CHECK_HIP(hipGetLastError());
    return OM_OK;
}

int launch_conv_igemm(const ConvArgs& a, hipStream_t stream) {
    OM_REQUIRE(a.in && a.w && a.scale && a.shift && a.out, OM_EINVAL, "conv: null pointer");
    OM_REQUIRE(a.cin % 32 == 0 && a.cin >= 32, OM_EINVAL, "conv: cin=%d must be a multiple of 32", a.cin);
    OM_REQUIRE(a.ks == 1 || a.ks == 3, OM_EINVAL, "conv: ksize=%d not supported", a.ks);
    OM_REQUIRE(a.stride == 1 || a.stride == 2, OM_EINVAL, "conv: stride=%d not supported", a.stride);
    OM_REQUIRE(a.in_pix_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(a.in) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(a.w) & 15) == 0,
               OM_EINVAL, "conv: input view / weights must be 16-byte aligned");
    OM_REQUIRE(a.cout_pad % 32 == 0 && a.cout <= a.cout_pad, OM_EINVAL, "conv: cout_pad=%d", a.cout_pad);
    OM_REQUIRE((long long)a.B * a.H * a.W < (1ll << 31) && (long long)a.B * a.Ho * a.Wo < (1ll << 31), OM_EINVAL,
               "conv: more than 2^31 pixels");
    IgemmParams p;
    p.in = a.in; p.w = a.w; p.scale = a.scale; p.shift = a.shift; p.res = a.res; p.out = a.out;
    p.H = a.H; p.W = a.W; p.cin = a.cin; p.in_pix_stride = a.in_pix_stride;
    p.Ho = a.Ho; p.Wo = a.Wo; p.HoWo = a.Ho * a.Wo; p.cout = a.cout;
    p.ks = a.ks; p.stride = a.stride; p.pad = a.ks / 2;
    p.M = a.B * a.Ho * a.Wo; p.kc = a.cin / 32; p.taps = a.ks * a.ks; p.ksteps = p.taps * p.kc;
    p.leaky = a.leaky; p.res_pix_stride = a.res_pix_stride; p.out_pix_stride = a.out_pix_stride;
    p.out_mode = a.out_mode; p.up = a.up;
    OM_REQUIRE(!(a.res && a.out_mode != 0), OM_EINVAL, "conv: residual only with plain NHWC output");
    if (a.cout_pad % 128 == 0) {
        p.n_tiles = a.cout_pad / 128;
        return launch_cfg<128, 128, 64, 64>(p, stream);
    } else if (a.cout_pad % 64 == 0) {
        p.n_tiles = a.cout_pad / 64;
        return launch_cfg<128, 64, 64, 32>(p, stream);
    }
    p.n_tiles = a.cout_pad / 32;
    return launch_cfg<128, 32, 32, 32>(p, stream);
}

}  // namespace om
