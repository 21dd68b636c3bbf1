// Postprocess of OrienMask on gfx950: box decode, confidence threshold, exact top-k, class-aware
// greedy NMS and orientation-map mask assembly, with no host synchronisation.
//
// Restates OrienMaskYOLOPostProcess.apply (/root/reference/eval/orienmask_yolo_postprocess.py:66-166),
// batched_nms (/root/reference/eval/function.py:77-103) and the CPU NMS backend
// (/root/reference/eval/src/nms_cpu.cpp:4-63: corners cx +- w/2, area (x2-x1)*(y2-y1), suppress when
// IoU >= threshold, keep reported in ascending input order).  The reference runs ~30 torch ops per
// image with four host round trips (nonzero, two numel() branches, the CUDA NMS bit-matrix copy,
// nms_kernel.cu:105-139) and materialises ~1 GB of temporaries per image; here it is three launches:
//
//   post_decode_kernel  grid (tiles, B)   conf = sigmoid(cls) * sigmoid(obj) for every
//                       (candidate, class) pair; keys (float bits, 0 = below threshold) of the tiles
//                       that have a passing pair to the workspace, per-tile pass counts, level-1 radix
//                       histogram -- and the pairs that
//                       pass, as (key, pair) words, appended to a per-image list (chip-wide compaction:
//                       a few hundred to a few thousand of the 1.46 M pairs of a 544^2 image pass).
//   post_select_kernel  grid (B), 1024 thr  the list, when it holds every passing pair (<= 4096), is
//                       sorted in LDS: its head is the exact top-nms_pre (ties -> lowest pair index).
//                       Otherwise (dense heads): 3-level radix select on the key bits over the image's
//                       whole key array (three passes of 5.8 MB by one workgroup), index-ordered
//                       compaction, bitonic sort.  Then
//                       box decode of the <= nms_pre survivors, 64-bit suppression bit-matrix in LDS
//                       (one u64 = one row segment of the reference's 64-wide CUDA tiling), serial
//                       wave-level reduction, top-nms_post, per-detection mask constants.
//   post_mask_kernel    grid (pixels/4096, B*nms_post)  bilinear x4 of the two orientation planes
//                       of the detection's anchor evaluated on the fly + the two |P - c| < t tests;
//                       16 pixels per thread, one 16-byte store each (the only HBM-heavy step).
//
// All comparisons that decide indices use IEEE fp32 operations in the reference's order; this file
// is compiled with -ffp-contract=off and the only fused multiply-adds are the explicit fmaf calls of
// the bilinear taps (the placement torch's CPU kernel compiles to; see oracle/orienmask_ref.py).
#include <type_traits>

#include "om_common.h"
#include "ref_math.h"

namespace om {

constexpr int DEC_TILE = 2048;          // pairs per decode workgroup
constexpr int SEL_THREADS = 1024;
constexpr int SEL_MAXN = 1024;          // nms_pre limit of the fused path
constexpr int SEL_LDS_MASK_N = 512;     // up to this many candidates the suppression bit-matrix stays in LDS (32 KiB); above,
                                        // it lives in the workspace (what the reference's CUDA backend always does)
constexpr int L1_BINS = 2048;           // key >> 19
constexpr int MASK_PX = 16;             // pixels per thread in the mask kernel
constexpr int DEC_VISITS = 10;          // visits per thread and sweep of the decode kernel: (DEC_TILE / C + 2) * max(CV, CS) <= 2560 for C <= 251
constexpr int DEC_VISITS_MANY = 24;     // ... and of its instantiation for larger class counts (every C < 2048: LVIS's 1203 takes 15)
constexpr int SEL_LIST_MAX = 4096;      // passing pairs per image the compacted list holds (= the u64 words of the LDS bit-matrix)

struct PostParams {
    om_post_cfg cfg;
    const float* bbox[OM_MAX_SCALES];
    const float* oriens;
    int B;
    int ncand, npairs, ntiles;
    int cand_off[OM_MAX_SCALES + 1];
    unsigned* keys;        // [B][ntiles*DEC_TILE]
    int* tile_count;       // [B][ntiles]
    unsigned* hist1;       // [B][L1_BINS]
    unsigned* list_count;  // [B] passing pairs of the image (zeroed with hist1); may exceed SEL_LIST_MAX: the list is then incomplete
    unsigned long long* list;   // [B][SEL_LIST_MAX] (key << 32) | ~pair of the passing pairs, in no particular order
    float* det_par;        // [B][nms_post][8]
    unsigned long long* nms_mask;   // [B][SEL_MAXN * SEL_MAXN / 64], used when nms_pre > SEL_LDS_MASK_N
    float* out_bbox;
    int64_t* out_cls;
    uint8_t* out_mask;
    int32_t* out_count;
    int32_t* out_keep;
    // anchors per scale (om_post_cfg.anchors_of_scale, or anchors_per_scale for every scale) and their prefix sums: anchor FIELD
    // a_off[s] + a owns orientation channels 2 * field and 2 * field + 1 (the heads' orientation maps concatenated, scale after scale)
    int a_cnt[OM_MAX_SCALES];
    int a_off[OM_MAX_SCALES + 1];
    // om_postprocess_candidates: the select kernel stops after the candidate list (for a caller-supplied NMS callable)
    float* cand_dets;      // [B][nms_pre][5] in the reference's LIST order (postprocess.py:102-122); NULL: the fused path
    int64_t* cand_cls;     // [B][nms_pre]
    int32_t* cand_field;   // [B][nms_pre] anchor field of the candidate
    int mask_chunk;        // post_mask_kernel: detections of a field per workgroup (launch_post_mask)
    int dec_visits;        // which post_decode_kernel instantiation runs (fill_params)
};

// candidate index -> (scale, anchor slot, pixel)
__device__ __forceinline__ void locate(const PostParams& p, int cand, int& s, int& a, int& pix) {
    s = (cand >= p.cand_off[1]) + (cand >= p.cand_off[2]);
    const int local = cand - p.cand_off[s];
    const int hw = p.cfg.grid_h[s] * p.cfg.grid_w[s];
    // anchors_per_scale <= 3 (fill_params): the quotient local / hw by two compares instead of a 32-bit division
    const int a1 = local >= hw, a2 = local >= 2 * hw;
    a = a1 + a2;
    pix = local - (a1 ? hw : 0) - (a2 ? hw : 0);
}

// ------------------------------------------------------------------------------------------------
// Index-deciding comparisons without compare instructions (round 4; DESIGN.md 3.2, tools/hazard_probe).
// On MI355X a dense run of VALU compares into SGPR pairs that SALU combines (v_cmp_*_e64 -> s_and_b64 / s_or_b64 ->
// v_cndmask) returned STALE lane masks while a co-resident wave of another kernel issued gfx950's wide-K matrix instructions;
// single compares through VCC and divergent branches never did (profiles/r02_experiments.md section 6).  Whatever decides an
// INDEX here -- which pair passes conf_thresh, which box suppresses which, the order of the sort -- is therefore evaluated on
// bit patterns in VGPRs: a predicate is bit 0 of an unsigned, conjunctions are ANDs, and the only compare left is the "!= 0" of
// the branch that consumes the result (the VCC / exec-mask class).  The subtraction is inline asm so that the compiler cannot
// fold the borrow test back into a compare.  Bit-identical to the IEEE comparisons for every operand (NaN: false; -0 == +0).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned sub_u32_opaque(unsigned a, unsigned b) {
    unsigned r;
    asm("v_sub_u32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a < b for any two 32-bit unsigned values: the borrow out of a - b
__device__ __forceinline__ unsigned lt_u32_bit(unsigned a, unsigned b) {
    return ((~a & b) | ((~a | b) & sub_u32_opaque(a, b))) >> 31;
}
__device__ __forceinline__ unsigned eq_u32_bit(unsigned a, unsigned b) {
    const unsigned x = a ^ b;
    return ((x | sub_u32_opaque(0u, x)) >> 31) ^ 1u;
}
// a < b for non-negative ints below 2^31 (pair indices, keys of confidences <= 1): the sign of the difference
__device__ __forceinline__ unsigned lt_i31_bit(int a, int b) { return sub_u32_opaque((unsigned)a, (unsigned)b) >> 31; }
// float -> unsigned key with the IEEE order of the numbers (sign-magnitude -> biased two's complement; -0 and +0 share a key)
__device__ __forceinline__ unsigned f32_order_key(float x) {
    const int b = __float_as_int(x), sg = b >> 31;
    return (unsigned)(((b & 0x7fffffff) ^ sg) - sg) ^ 0x80000000u;
}
__device__ __forceinline__ unsigned f32_is_num_bit(float x) { return lt_u32_bit(__float_as_uint(x) & 0x7fffffffu, 0x7f800001u); }
__device__ __forceinline__ unsigned f32_gt_bit(float a, float t) {      // a > t
    return lt_u32_bit(f32_order_key(t), f32_order_key(a)) & f32_is_num_bit(a) & f32_is_num_bit(t);
}
__device__ __forceinline__ unsigned f32_ge_bit(float a, float t) {      // a >= t
    return (lt_u32_bit(f32_order_key(a), f32_order_key(t)) ^ 1u) & f32_is_num_bit(a) & f32_is_num_bit(t);
}
// a < b for two 64-bit sort words (key << 32 | ~position)
__device__ __forceinline__ unsigned lt_u64_bit(unsigned long long a, unsigned long long b) {
    const unsigned ah = (unsigned)(a >> 32), bh = (unsigned)(b >> 32);
    return lt_u32_bit(ah, bh) | (eq_u32_bit(ah, bh) & lt_u32_bit((unsigned)a, (unsigned)b));
}

// ------------------------------------------------------------------------------------------------
// decode: postprocess.py:126-139 (confidence part) and :102 (threshold)
// ------------------------------------------------------------------------------------------------
template <int VISITS>
__global__ __launch_bounds__(256) void post_decode_kernel(const PostParams p) {
    __shared__ unsigned hist[L1_BINS];
    __shared__ int wcnt[4];
    __shared__ unsigned long long s_tab[32];     // glibc's 2^(k/32) table (ref_math.h), indexed per lane
    __shared__ float s_obj[DEC_TILE + 2];       // sigmoid(objectness) of the candidates this tile touches
    __shared__ __attribute__((aligned(16))) unsigned s_keys[DEC_TILE];      // the tile's keys; they reach memory only if one of them passed
    __shared__ int s_any;
    const int tid = threadIdx.x, b = blockIdx.y, tile = blockIdx.x;
    if (tid < 32) s_tab[tid] = kExp2fTab[tid];
    if (tid == 0) s_any = 0;
    __syncthreads();
    const int C = p.cfg.num_classes, per = 5 + C;
    const int P0 = tile * DEC_TILE, P1 = min(P0 + DEC_TILE, p.npairs);      // this tile's pairs
    const int cand_first = P0 / C;
    const int cand_last = min(p.ncand - 1, (P0 + DEC_TILE - 1) / C);
    const int ncand_t = cand_last - cand_first + 1;
    unsigned long long* const list = p.list + (size_t)b * SEL_LIST_MAX;
    int cnt = 0;
    // One (candidate, class) pair in two steps, so that a sweep can have ALL its loads in flight before the first sigmoid (a
    // workgroup is a few microseconds of work behind one dependent load per visit: the kernel was bound by exactly that latency).
    // fetch: the class logit, or 0 for a lane whose pair is not this tile's;  finish: confidence, threshold, key into LDS, histogram,
    // and the wave's passing pairs appended to the image's list (one atomic per wave that has any).  Called by whole waves.
    // A candidate whose objectness is not above the threshold cannot pass in any class: conf = fl(sigmoid(cls) * sigmoid(obj)) with
    // sigmoid(cls) <= 1, and rounding is monotone, so conf <= sigmoid(obj) <= conf_thresh (a NaN is "not above" either way).  Its
    // class logits are neither loaded nor evaluated -- on real heads that is nearly every candidate (background cells), and in
    // the vectorised sweep a wave is one candidate, so the skip is a whole-wave branch.
    const float thr = p.cfg.conf_thresh;
    auto in_tile = [&](int cand_l, int cls, bool live) -> unsigned {
        const int pair = (cand_first + cand_l) * C + cls;
        return (live ? 1u : 0u) & (lt_i31_bit(pair, P0) ^ 1u) & lt_i31_bit(pair, P1) & f32_gt_bit(s_obj[cand_l], thr);
    };
    // The per-scale constants by STATIC index into the parameter block (scalar loads, once) and selects: p.bbox[s] with a run-time s
    // is a memory load per visit, and the logit's load then waits for it -- one dependent round trip per visit again.
    const int off1 = p.cand_off[1], off2 = p.cand_off[2];
    const int hw0 = p.cfg.grid_h[0] * p.cfg.grid_w[0], hw1 = p.cfg.grid_h[1] * p.cfg.grid_w[1], hw2 = p.cfg.grid_h[2] * p.cfg.grid_w[2];
    const float* const bb0 = p.bbox[0];
    const float* const bb1 = p.bbox[1];
    const float* const bb2 = p.bbox[2];
    const int pstride = p.cfg.bbox_pix_stride;
    typedef const __attribute__((address_space(1))) float* global_f32;      // (the blend below loses the pointers' address space)
    auto logit_ptr = [&](int cand) -> global_f32 {      // element 0 of the candidate's 5 + C values
        // blends by masks (cand >= off2 implies cand >= off1), not selects: the compiler turns a select chain over the three scales
        // into a table in scratch memory indexed by the scale -- a dependent load per visit, which is what this kernel must not have
        const int m1 = -(int)(cand >= off1), m2 = -(int)(cand >= off2);
        const int local = cand - (off1 & m1) - ((off2 - off1) & m2);
        const int hw = hw0 + ((hw1 - hw0) & m1) + ((hw2 - hw1) & m2);
        const global_f32 base = (global_f32)(reinterpret_cast<uintptr_t>(bb0) +
                                             ((reinterpret_cast<uintptr_t>(bb1) - reinterpret_cast<uintptr_t>(bb0)) & (uintptr_t)(long long)m1) +
                                             ((reinterpret_cast<uintptr_t>(bb2) - reinterpret_cast<uintptr_t>(bb1)) & (uintptr_t)(long long)m2));
        const bool a1 = local >= hw, a2 = local >= 2 * hw;      // anchors_per_scale <= 3
        const int pix = local - (a1 ? hw : 0) - (a2 ? hw : 0);
        // b * hw + pix < 2^24 pixels per scale in the batch (om_postprocess checks), strides and a * per far below: 24-bit multiplies
        return base + (size_t)__umul24(__umul24(b, hw) + pix, pstride) + __umul24((int)a1 + (int)a2, per);
    };
    auto fetch = [&](int cand_l, int cls, bool live) -> float {
        if (!in_tile(cand_l, cls, live)) return 0.f;
        return logit_ptr(cand_first + cand_l)[5 + cls];
    };
    auto finish = [&](int cand_l, int cls, bool live, float x, auto vector_path) {
        const int pair = (cand_first + cand_l) * C + cls;
        unsigned key = 0;
        if (in_tile(cand_l, cls, live)) {
            const float conf = (decltype(vector_path)::value ? sigmoid_vector_ref(x) : sigmoid_scalar_ref(x, s_tab)) * s_obj[cand_l];
            if (f32_gt_bit(conf, thr)) {
                key = __float_as_uint(conf);
                ++cnt;
                s_keys[pair - P0] = key;
            }
        }
        const unsigned long long pass = __ballot(key != 0);
        if (pass) {
            // level-1 histogram.  On dense heads (random-init-like logits, every pair passes with a confidence near
            // 0.25) all 64 lanes of a wave hit ONE bin, and 64 same-address LDS atomics serialise: the lanes that share the first
            // passing lane's bin are counted by a ballot and added once, twice over; whatever is left adds itself.
            const unsigned bin = key >> 19;
            unsigned long long todo = pass;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                if (todo) {
                    const int leader = __ffsll((long long)todo) - 1;
                    const unsigned lb = __shfl(bin, leader);
                    const unsigned long long same = __ballot(key != 0 && bin == lb) & todo;
                    if ((tid & 63) == leader) atomicAdd(&hist[lb], (unsigned)__popcll(same));
                    todo &= ~same;
                }
            }
            if ((todo >> (tid & 63)) & 1ull) atomicAdd(&hist[bin], 1u);
            // the compacted list only serves images with at most SEL_LIST_MAX passing pairs (post_select_kernel: from_list); its
            // counter is compared with nms_pre and SEL_LIST_MAX, never used as a number beyond them.  Once it is seen above
            // SEL_LIST_MAX nothing is appended any more: 22 760 waves per dense image adding to ONE word cost 8 ms per batch.
            unsigned at = 0;
            if ((tid & 63) == __ffsll((long long)pass) - 1) {
                const unsigned seen = __hip_atomic_load(&p.list_count[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                at = seen > (unsigned)SEL_LIST_MAX ? seen : atomicAdd(&p.list_count[b], (unsigned)__popcll(pass));
            }
            at = __shfl(at, __ffsll((long long)pass) - 1);
            if (key != 0) {
                const unsigned slot = at + (unsigned)__popcll(pass & ((1ull << (tid & 63)) - 1ull));
                if (slot < (unsigned)SEL_LIST_MAX) list[slot] = ((unsigned long long)key << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)pair);
            }
        }
    };
    // torch evaluates a row of C class logits with its vectorised sigmoid (Sleef) for the first CV = (C / 32) * 32 classes and
    // with the scalar one (glibc) for the tail (ref_math.h: sigmoid_class_ref).  In pair order every wave would hold classes of
    // both kinds and execute both functions; the tile's pairs are therefore visited in two sweeps, the vectorised classes of
    // its candidates (whole waves: CV is a multiple of 32, the sweep is padded to one of 64) and then the scalar tails.  One
    // division per thread and sweep; the next visit is 256 items further: + (256 / n, 256 % n) with a carry.
    // A sweep has at most VISITS (DEC_VISITS, or DEC_VISITS_MANY for class counts beyond 251) visits per thread (n_c <= DEC_TILE / C + 2 candidates: the host checks the bound).
    const int CV = C & ~31, CS = C - CV;
    float xv[VISITS], xs[VISITS];
    // (plain unrolled loops with compile-time indices: the arrays must stay in registers)
#define OM_DEC_STEP(n_per) { cand_l += dq; cls += dr; if (cls >= (n_per)) { cls -= (n_per); ++cand_l; } }
#define OM_DEC_FETCH(n_per, cls0, x)                                                              \
    if ((n_per) != 0) {                                                                           \
        const int n = ncand_t * (n_per), dq = 256 / (n_per), dr = 256 - dq * (n_per);             \
        int cand_l = tid / (n_per), cls = tid - cand_l * (n_per);                                 \
        _Pragma("unroll") for (int v = 0; v < VISITS; ++v) {                                  \
            x[v] = fetch(cand_l, (cls0) + cls, tid + 256 * v < n);                                \
            OM_DEC_STEP(n_per)                                                                    \
        }                                                                                         \
    }
#define OM_DEC_FINISH(n_per, cls0, x, vector_path)                                                \
    if ((n_per) != 0) {                                                                           \
        const int n = ncand_t * (n_per), n_pad = (n + 63) & ~63, dq = 256 / (n_per), dr = 256 - dq * (n_per); \
        int cand_l = tid / (n_per), cls = tid - cand_l * (n_per);                                 \
        _Pragma("unroll") for (int v = 0; v < VISITS; ++v) {                                  \
            if (tid + 256 * v < n_pad) finish(cand_l, (cls0) + cls, tid + 256 * v < n, x[v], vector_path);   /* wave-uniform */ \
            OM_DEC_STEP(n_per)                                                                    \
        }                                                                                         \
    }
    // sigmoid(obj) once per candidate (postprocess.py:128: a strided view -> torch's scalar loop -> glibc expf)
    for (int i = tid; i < ncand_t; i += 256) {
        const float so = sigmoid_scalar_ref(logit_ptr(cand_first + i)[4], s_tab);
        s_obj[i] = so;
        if (f32_gt_bit(so, thr)) s_any = 1;
    }
    __syncthreads();
    if (!s_any) {      // no candidate of this tile can pass (the usual case on real heads): no keys, no histogram
        if (tid == 0) p.tile_count[b * p.ntiles + tile] = 0;
        return;
    }
    for (int i = tid; i < L1_BINS; i += 256) hist[i] = 0;
    for (int i = tid; i < DEC_TILE; i += 256) s_keys[i] = 0;      // pairs beyond npairs (last tile) stay 0
    __syncthreads();
    OM_DEC_FETCH(CV, 0, xv)
    OM_DEC_FETCH(CS, CV, xs)
    OM_DEC_FINISH(CV, 0, xv, std::true_type{})
    OM_DEC_FINISH(CS, CV, xs, std::false_type{})
#undef OM_DEC_FINISH
#undef OM_DEC_FETCH
#undef OM_DEC_STEP
    // workgroup total
    for (int d = 32; d > 0; d >>= 1) cnt += __shfl_down(cnt, d);
    if ((tid & 63) == 0) wcnt[tid >> 6] = cnt;
    __syncthreads();
    const int total = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    if (tid == 0) p.tile_count[b * p.ntiles + tile] = total;
    if (total) {
        // the select kernel's radix passes (dense heads only) read the keys of the tiles with tile_count > 0, nothing else
        uint4* keys = reinterpret_cast<uint4*>(p.keys + (size_t)b * p.ntiles * DEC_TILE + (size_t)tile * DEC_TILE);
        for (int i = tid; i < DEC_TILE / 4; i += 256) keys[i] = reinterpret_cast<const uint4*>(s_keys)[i];
        unsigned* g = p.hist1 + (size_t)b * L1_BINS;
        for (int i = tid; i < L1_BINS; i += 256)
            if (hist[i]) atomicAdd(&g[i], hist[i]);
    }
}

// ------------------------------------------------------------------------------------------------
// workgroup helpers (1024 threads = 16 waves)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int block_scan_excl(int v, int* s_wave, int& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(x, d);
        if (lane >= d) x += y;
    }
    if (lane == 63) s_wave[wave] = x;
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SEL_THREADS / 64; ++w) {
        const int t = s_wave[w];
        if (w < wave) off += t;
        tot += t;
    }
    __syncthreads();
    total = tot;
    return off + x - v;
}

// Largest bin t with sum_{i >= t} hist[i] >= need; also the count strictly above it.
// hist has 2 * SEL_THREADS bins at most (nbins <= 2048).  Results through s_res[0..1].
__device__ __forceinline__ void find_bin_from_top(const unsigned* hist, int nbins, int need, int* s_wave,
                                                  int* s_res) {
    const int tid = threadIdx.x;
    const int ba = nbins - 1 - 2 * tid, bb = ba - 1;
    const int ha = ba >= 0 ? (int)hist[ba] : 0, hb = bb >= 0 ? (int)hist[bb] : 0;
    int total;
    const int excl = block_scan_excl(ha + hb, s_wave, total);
    if (ba >= 0 && excl < need && excl + ha >= need) { s_res[0] = ba; s_res[1] = excl; }
    if (bb >= 0 && excl + ha < need && excl + ha + hb >= need) { s_res[0] = bb; s_res[1] = excl + ha; }
    __syncthreads();
}

// Descending bitonic sort of n_pad (power of two, any multiple of the workgroup or below it) u64 values in LDS.
__device__ __forceinline__ void bitonic_sort_desc(unsigned long long* v, int n_pad) {
    for (int k = 2; k <= n_pad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n_pad; i += SEL_THREADS) {
                const int partner = i ^ j;
                if (partner > i) {
                    const unsigned long long a = v[i], c = v[partner];
                    const unsigned desc = sub_u32_opaque((unsigned)(i & k), 1u) >> 31;      // (i & k) == 0
                    if ((desc & lt_u64_bit(a, c)) | ((desc ^ 1u) & lt_u64_bit(c, a))) { v[i] = c; v[partner] = a; }
                }
            }
            __syncthreads();
        }
    }
}

// Suppression bit-matrix.  Boxes are already in visiting (score-descending) order.  mask: [n][words] u64 (LDS or
// global); bit jj of word wd of row pi = "box pi suppresses box 64 wd + jj".  Two semantics (om_post_cfg.nms_semantics):
//   0  the reference's CPU backend, eval/src/nms_cpu.cpp:38-60: areas from the corners, suppress when IoU >= thr
//   1  the reference's CUDA backend, eval/src/nms_kernel.cu:13-23,62: areas w * h, suppress when IoU > thr
// (sarea holds whichever area the semantics asks for; the corner arithmetic is the same in both.)
__device__ __forceinline__ unsigned long long nms_mask_word(const float* sx1, const float* sy1, const float* sx2,
                                                            const float* sy2, const float* sarea, int n, float thr,
                                                            bool strict, int pi, int wd) {
    unsigned long long bits = 0;
    const int j0 = wd * 64;
    if (j0 + 63 > pi) {
        const float ix1 = sx1[pi], iy1 = sy1[pi], ix2 = sx2[pi], iy2 = sy2[pi], ia = sarea[pi];
        const int jend = min(64, n - j0);
        for (int jj = max(0, pi + 1 - j0); jj < jend; ++jj) {
            const int pj = j0 + jj;
            const float xx1 = fmaxf(ix1, sx1[pj]), yy1 = fmaxf(iy1, sy1[pj]);
            const float xx2 = fminf(ix2, sx2[pj]), yy2 = fminf(iy2, sy2[pj]);
            const float w = fmaxf(0.0f, xx2 - xx1), h = fmaxf(0.0f, yy2 - yy1);
            const float inter = w * h;
            const float ovr = inter / (ia + sarea[pj] - inter);
            bits |= (unsigned long long)(strict ? f32_gt_bit(ovr, thr) : f32_ge_bit(ovr, thr)) << jj;
        }
    }
    return bits;
}

__device__ __forceinline__ void nms_bitmask(const float* sx1, const float* sy1, const float* sx2, const float* sy2,
                                            const float* sarea, int n, float thr, bool strict, unsigned long long* mask,
                                            int words) {
    for (int item = threadIdx.x; item < n * words; item += blockDim.x) {
        const int pi = item / words, wd = item - pi * words;
        mask[item] = nms_mask_word(sx1, sy1, sx2, sy2, sarea, n, thr, strict, pi, wd);
    }
    __syncthreads();
}

// Serial part of greedy NMS (nms_cpu.cpp:38-60 / the host loop of nms_kernel.cu:123-134), 64 rows at a time:
// wave 0 walks the diagonal word of the block with scalar operations (64 dependent steps in registers), then every
// wave ORs the kept rows of the block into the removed-words to the right.  s_removed: [words] u64 in LDS;
// keep_flag[pi] = 1 when box pi survives (LDS or global).  blockDim.x must be a multiple of 64.
__device__ __forceinline__ void nms_reduce_blocked(const unsigned long long* mask, int n, int words,
                                                   unsigned long long* s_removed, unsigned long long* s_kept_bits,
                                                   unsigned char* keep_flag) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    for (int w = tid; w < words; w += blockDim.x) s_removed[w] = 0ull;
    __syncthreads();
    for (int blk = 0; blk < words; ++blk) {
        const int row = blk * 64 + lane;
        if (wave == 0) {
            const unsigned long long diag = row < n ? mask[(size_t)row * words + blk] : 0ull;
            unsigned long long rem = s_removed[blk];
            unsigned long long kept = 0ull;
            const int rows_here = min(64, n - blk * 64);
            for (int i = 0; i < rows_here; ++i) {
                const unsigned long long d = __shfl(diag, i);      // uniform
                if (!((rem >> i) & 1ull)) { kept |= 1ull << i; rem |= d; }
            }
            if (lane == 0) *s_kept_bits = kept;
            if (row < n) keep_flag[row] = (unsigned char)((kept >> lane) & 1ull);
        }
        __syncthreads();
        const unsigned long long kept = *s_kept_bits;
        const bool mine = row < n && ((kept >> lane) & 1ull);
        for (int w = blk + 1 + wave; w < words; w += nwaves) {
            unsigned long long v = mine ? mask[(size_t)row * words + w] : 0ull;
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) v |= __shfl_xor(v, d);
            if (lane == 0) s_removed[w] |= v;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// select + NMS: postprocess.py:102-122, :146-154; function.py:77-103; nms_cpu.cpp:4-63
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(SEL_THREADS) void post_select_kernel(const PostParams p) {
    __shared__ unsigned hist[L1_BINS];
    __shared__ int s_wave[SEL_THREADS / 64];
    __shared__ int s_res[4];
    __shared__ unsigned long long s_comp[SEL_MAXN];         // (key << 32) | ~position
    __shared__ unsigned s_key[SEL_MAXN];
    __shared__ int s_pair[SEL_MAXN];
    __shared__ float s_bx[SEL_MAXN], s_by[SEL_MAXN], s_bw[SEL_MAXN], s_bh[SEL_MAXN];
    __shared__ float s_x1[SEL_MAXN], s_y1[SEL_MAXN], s_x2[SEL_MAXN], s_y2[SEL_MAXN], s_area[SEL_MAXN];
    __shared__ unsigned long long s_mask[SEL_LDS_MASK_N * (SEL_LDS_MASK_N / 64)];
    __shared__ unsigned long long s_removed[SEL_MAXN / 64];
    __shared__ unsigned long long s_kept_bits;
    __shared__ float s_red[SEL_THREADS / 64];
    __shared__ unsigned char s_keep[SEL_MAXN];
    __shared__ unsigned char s_keep_pos[SEL_MAXN];
    __shared__ short s_ord[SEL_MAXN];

    const int tid = threadIdx.x, b = blockIdx.x;
    const int nms_pre = p.cfg.nms_pre, nms_post = p.cfg.nms_post;
    const unsigned* keys = p.keys + (size_t)b * p.ntiles * DEC_TILE;
    const int* tcount = p.tile_count + (size_t)b * p.ntiles;

    // ---- how many pairs passed the threshold
    const int total = (int)min(p.list_count[b], 0x7FFFFFFFu);
    if (total == 0) {
        if (tid == 0) p.out_count[b] = 0;
        return;
    }
    const int n = total > nms_pre ? nms_pre : total;
    // case A (more than nms_pre passed): the reference's list IS the sorted top-k, so list position = pi.
    // case B: the reference's list is the index-ordered one, NMS visits it through the argsort.
    const bool caseA = total > nms_pre;
    const bool from_list = total <= SEL_LIST_MAX;      // the compacted list holds every passing pair

    if (from_list) {
        // ---- the image's passing pairs, (key << 32) | ~pair, sorted descending in LDS: score descending, ties by pair index
        // ascending -- postprocess.py:102-122 (index-ordered nonzero(), then topk / argsort) visits them in exactly this order,
        // and the first n are the selection (ties at the cut -> lowest pair index)
        unsigned long long* const s_list = s_mask;      // the bit-matrix is built after the list is consumed
        int l_pad = 64;
        while (l_pad < total) l_pad <<= 1;
        const unsigned long long* glist = p.list + (size_t)b * SEL_LIST_MAX;
        for (int i = tid; i < l_pad; i += SEL_THREADS) s_list[i] = i < total ? glist[i] : 0ull;
        __syncthreads();
        bitonic_sort_desc(s_list, l_pad);
        unsigned long long mine = 0ull;
        int pos = tid;
        if (tid < n) {
            mine = s_list[tid];
            if (!caseA) {      // list position = rank of the pair index among the n pairs (the index-ordered list of case B)
                const unsigned my_lo = (unsigned)mine;
                pos = 0;
                for (int i = 0; i < n; ++i) pos += (int)lt_u32_bit(my_lo, (unsigned)s_list[i]);      // ~pair larger <=> pair smaller
            }
        }
        __syncthreads();
        if (tid < n) {
            const unsigned key = (unsigned)(mine >> 32);
            s_key[pos] = key;
            s_pair[pos] = (int)(0xFFFFFFFFu - (unsigned)mine);
            s_comp[tid] = ((unsigned long long)key << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)pos);
        }
        __syncthreads();
    } else {
    // ---- exact threshold key T and number r of ties to take (3-level radix select on the bits)
    unsigned T = 0;      // take every key > T, plus the first r keys == T in index order
    int r = 0, above = total;
    if (total > nms_pre) {
        for (int i = tid; i < L1_BINS; i += SEL_THREADS) hist[i] = p.hist1[(size_t)b * L1_BINS + i];
        __syncthreads();
        find_bin_from_top(hist, L1_BINS, nms_pre, s_wave, s_res);
        const unsigned t1 = s_res[0];
        int need = nms_pre - s_res[1];
        above = s_res[1];
        __syncthreads();
        // Round 4: the three passes over the image's keys (5.8 MB on dense heads) were one uint4 per thread and iteration with a
        // workgroup scan -- two barriers -- per 4096 keys in the last one: 1.5 ms per batch on all-pass heads (bench.py --heads
        // allpass), bound by the latency of one load in flight per thread.  Now every wave owns a contiguous range of tiles, has
        // four 1-KiB rows in flight, and the compaction needs no barrier at all: the level-3 pass also counts, per wave, the keys
        // above the 24-bit prefix and a private 256-bin histogram of the keys on it, from which each wave's number of selected
        // keys -- its offset in the index-ordered list -- follows once T is known.
        // level 2: bits 18..8 of keys whose top bits equal t1
        for (int i = tid; i < 2048; i += SEL_THREADS) hist[i] = 0;
        __syncthreads();
        const int lane = tid & 63, wave = tid >> 6;
        const int tw0 = (int)((long long)p.ntiles * wave / (SEL_THREADS / 64)), tw1 = (int)((long long)p.ntiles * (wave + 1) / (SEL_THREADS / 64));
        // rows of 256 keys (64 lanes x uint4); a tile has DEC_TILE / 256 = 8 of them; four rows per step
        auto for_rows = [&](auto body) {
            for (int t = tw0; t < tw1; ++t) {
                if (tcount[t] == 0) continue;           // a tile without a pass has no keys in memory
                const unsigned* kt = keys + (size_t)t * DEC_TILE + lane * 4;
#pragma unroll
                for (int r0 = 0; r0 < DEC_TILE / 256; r0 += 4) {
                    uint4 k4[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) k4[q] = *reinterpret_cast<const uint4*>(kt + (r0 + q) * 256);
#pragma unroll
                    for (int q = 0; q < 4; ++q) body(k4[q], t * DEC_TILE + (r0 + q) * 256 + lane * 4);
                }
            }
        };
        for_rows([&](const uint4& k4, int) {
            const unsigned kk[4] = {k4.x, k4.y, k4.z, k4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if ((eq_u32_bit(kk[e], 0u) ^ 1u) & eq_u32_bit(kk[e] >> 19, t1)) atomicAdd(&hist[(kk[e] >> 8) & 0x7FFu], 1u);
        });
        __syncthreads();
        find_bin_from_top(hist, 2048, need, s_wave, s_res);
        const unsigned t2 = s_res[0];
        above += s_res[1];
        need -= s_res[1];
        __syncthreads();
        // level 3: low 8 bits of the keys on the prefix hi, per wave (s_mask is free until the NMS); keys above the prefix counted
        unsigned* const whist = reinterpret_cast<unsigned*>(s_mask);          // [16 waves][256]
        int* const wgt = reinterpret_cast<int*>(whist + (SEL_THREADS / 64) * 256);      // [16] keys above the prefix, per wave
        for (int i = tid; i < (SEL_THREADS / 64) * 256 + SEL_THREADS / 64; i += SEL_THREADS) whist[i] = 0;
        __syncthreads();
        const unsigned hi = (t1 << 11) | t2;
        int my_above = 0;
        for_rows([&](const uint4& k4, int) {
            const unsigned kk[4] = {k4.x, k4.y, k4.z, k4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                my_above += (int)lt_u32_bit(hi, kk[e] >> 8);
                if (eq_u32_bit(kk[e] >> 8, hi)) atomicAdd(&whist[wave * 256 + (kk[e] & 0xFFu)], 1u);      // (hi != 0: keys of passing pairs only)
            }
        });
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) my_above += __shfl_xor(my_above, d);
        if (lane == 0) wgt[wave] = my_above;
        __syncthreads();
        for (int i = tid; i < 256; i += SEL_THREADS) {
            unsigned v = 0;
            for (int w = 0; w < SEL_THREADS / 64; ++w) v += whist[w * 256 + i];
            hist[i] = v;
        }
        __syncthreads();
        find_bin_from_top(hist, 256, need, s_wave, s_res);
        const unsigned tlow = (unsigned)s_res[0];
        T = (hi << 8) | tlow;
        above += s_res[1];
        r = nms_pre - above;
        __syncthreads();
        // ---- index-ordered compaction (row-major (candidate, class) order, postprocess.py:102): this wave's first positions
        int run_gt = 0, run_eq = 0;
        for (int w = 0; w < wave; ++w) {
            int g = wgt[w];
            for (unsigned bin = tlow + 1; bin < 256; ++bin) g += (int)whist[w * 256 + bin];
            run_gt += g;
            run_eq += (int)whist[w * 256 + tlow];
        }
        for_rows([&](const uint4& k4, int pair0) {
            const unsigned kk[4] = {k4.x, k4.y, k4.z, k4.w};
            int ngt = 0, neq = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ngt += (int)lt_u32_bit(T, kk[e]);
                neq += (int)eq_u32_bit(kk[e], T);
            }
            const int mine = ngt | (neq << 16);
            if (__ballot(mine != 0) == 0ull) return;          // nearly every row of a dense image
            int incl = mine;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int y = __shfl_up(incl, d);
                if (lane >= d) incl += y;
            }
            const int tot = __shfl(incl, 63);
            int pg = run_gt + ((incl - mine) & 0xFFFF), pe = run_eq + ((incl - mine) >> 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (lt_u32_bit(T, kk[e])) {
                    if (pg < SEL_MAXN) { s_key[pg] = kk[e]; s_pair[pg] = pair0 + e; }
                    ++pg;
                } else if (eq_u32_bit(kk[e], T)) {
                    if (pe < r) { s_key[above + pe] = kk[e]; s_pair[above + pe] = pair0 + e; }
                    ++pe;
                }
            }
            run_gt += tot & 0xFFFF;
            run_eq += tot >> 16;
        });
    }
    __syncthreads();

    // ---- visiting order: score descending, ties by list position ascending
    int n_pad = 64;
    while (n_pad < n) n_pad <<= 1;
    if (tid < n_pad)
        s_comp[tid] = tid < n ? (((unsigned long long)s_key[tid] << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)tid)) : 0ull;
    __syncthreads();
    bitonic_sort_desc(s_comp, n_pad);
    }      // !from_list

    // ---- decode the survivors' boxes (postprocess.py:126-139) in visiting order
    if (tid < n) {
        const int pos = (int)(0xFFFFFFFFu - (unsigned)(s_comp[tid] & 0xFFFFFFFFull));
        s_ord[tid] = (short)pos;
        const int pair = s_pair[pos];
        const int C = p.cfg.num_classes;
        const int cand = pair / C;
        int s, a, pix;
        locate(p, cand, s, a, pix);
        const int gh = p.cfg.grid_h[s], gw = p.cfg.grid_w[s];
        const int gy = pix / gw, gx = pix - gy * gw;
        const float* q = p.bbox[s] + ((size_t)b * gh * gw + pix) * p.cfg.bbox_pix_stride + a * (5 + C);
        const int aid = p.cfg.anchor_mask[s][a];
        // tx, ty: strided views -> torch's scalar sigmoid (glibc expf), bit-exact; tw, th: MKL vsExp in the reference, matched
        // to one ulp by the correctly rounded value (ref_math.h)
        const float bx = (sigmoid_scalar_ref(q[0]) + (float)gx) / (float)gw;
        const float by = (sigmoid_scalar_ref(q[1]) + (float)gy) / (float)gh;
        const float bw = expf_cr(q[2]) * (p.cfg.anchor_w[aid] / (float)p.cfg.image_w);
        const float bh = expf_cr(q[3]) * (p.cfg.anchor_h[aid] / (float)p.cfg.image_h);
        s_bx[tid] = bx; s_by[tid] = by; s_bw[tid] = bw; s_bh[tid] = bh;
        if (p.cand_dets) {      // the candidate list itself, at its LIST position: what the reference hands to self.nms
            // (more than SEL_LIST_MAX passing pairs: the radix path's `pos` is the index-ordered compaction slot, but such an image
            // is always case A, whose list is the sorted top-k -- postprocess.py:107-110 -- i.e. the visiting rank)
            const size_t o = (size_t)b * nms_pre + ((caseA && !from_list) ? tid : pos);
            float* cd = p.cand_dets + o * 5;
            cd[0] = bx; cd[1] = by; cd[2] = bw; cd[3] = bh;
            cd[4] = __uint_as_float((unsigned)(s_comp[tid] >> 32));
            p.cand_cls[o] = pair - cand * C;
            p.cand_field[o] = p.a_off[s] + a;
        }
    }
    if (p.cand_dets) {
        if (tid == 0) p.out_count[b] = n;
        return;
    }
    // class offset, function.py:91-96: cls * (max_coordinate + 0.5); max_coordinate = 1.5 when normalized, else
    // dets[:, :2].max() + dets[:, 2:4].max() / 2 over the candidates of this image
    float class_step = 2.0f;
    if (!p.cfg.nms_normalized) {
        float mxy = -__builtin_inff(), mwh = -__builtin_inff();
        if (tid < n) { mxy = fmaxf(s_bx[tid], s_by[tid]); mwh = fmaxf(s_bw[tid], s_bh[tid]); }
        for (int pass = 0; pass < 2; ++pass) {
            float v = pass == 0 ? mxy : mwh;
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) v = fmaxf(v, __shfl_xor(v, d));
            __syncthreads();
            if ((tid & 63) == 0) s_red[tid >> 6] = v;
            __syncthreads();
            v = s_red[0];
            for (int w = 1; w < SEL_THREADS / 64; ++w) v = fmaxf(v, s_red[w]);
            if (pass == 0) mxy = v; else mwh = v;
        }
        class_step = (mxy + mwh / 2.0f) + 0.5f;
    }
    const bool cuda_sem = p.cfg.nms_semantics == 1;
    if (tid < n) {
        const int pair = s_pair[s_ord[tid]];
        const int cls = pair - (pair / p.cfg.num_classes) * p.cfg.num_classes;
        const float bx = s_bx[tid], by = s_by[tid], bw = s_bw[tid], bh = s_bh[tid];
        // corners (nms_cpu.cpp:17-20 == devIoU's a[0] -+ a[2] / 2, nms_kernel.cu:14-17) and the semantics' area
        const float off = (float)cls * class_step;
        const float ox = bx + off, oy = by + off;
        const float x1 = ox - bw / 2.0f, y1 = oy - bh / 2.0f, x2 = ox + bw / 2.0f, y2 = oy + bh / 2.0f;
        s_x1[tid] = x1; s_y1[tid] = y1; s_x2[tid] = x2; s_y2[tid] = y2;
        s_area[tid] = cuda_sem ? bw * bh : (x2 - x1) * (y2 - y1);       // nms_kernel.cu:20-21 / nms_cpu.cpp:22
    }
    __syncthreads();

    const int words = (n + 63) >> 6;
    unsigned long long* const mask = n <= SEL_LDS_MASK_N ? s_mask : p.nms_mask + (size_t)b * SEL_MAXN * (SEL_MAXN / 64);
    nms_bitmask(s_x1, s_y1, s_x2, s_y2, s_area, n, p.cfg.nms_thresh, cuda_sem, mask, words);
    nms_reduce_blocked(mask, n, words, s_removed, &s_kept_bits, s_keep);
    __syncthreads();

    // ---- output order (nms_cpu.cpp:62 ascending list position; postprocess.py:150-154 top-nms_post)
    int kept_total;
    const int my_keep = (tid < n) ? s_keep[tid] : 0;
    const int rank_sorted = block_scan_excl(my_keep, s_wave, kept_total);
    int K, slot = -1;
    // the CUDA backend returns keep in visiting (score-descending) order (nms_kernel.cu:136-139), so dets[keep] is sorted
    // and the top-nms_post of postprocess.py:150-154 is its head in every case
    if (caseA || cuda_sem || kept_total > nms_post) {
        K = kept_total < nms_post ? kept_total : nms_post;
        if (my_keep && rank_sorted < K) slot = rank_sorted;
    } else {
        K = kept_total;
        int* const s_slot = reinterpret_cast<int*>(s_key);      // s_key is dead once s_comp exists
        if (tid < n) s_keep_pos[tid] = 0;
        __syncthreads();
        if (my_keep) s_keep_pos[s_ord[tid]] = 1;
        __syncthreads();
        int dummy;
        const int flag = (tid < n) ? s_keep_pos[tid] : 0;        // thread tid speaks for list position tid
        const int rank_pos = block_scan_excl(flag, s_wave, dummy);
        if (flag) s_slot[tid] = rank_pos;
        __syncthreads();
        if (my_keep) slot = s_slot[s_ord[tid]];
    }
    if (tid == 0) p.out_count[b] = K;
    if (slot >= 0) {
        const int pos = s_ord[tid];
        const int pair = s_pair[pos];
        const int C = p.cfg.num_classes;
        const int cand = pair / C, cls = pair - cand * C;
        int s, a, pix;
        locate(p, cand, s, a, pix);
        const int aid = p.cfg.anchor_mask[s][a];
        const size_t o = (size_t)b * nms_post + slot;
        const float bx = s_bx[tid], by = s_by[tid], bw = s_bw[tid], bh = s_bh[tid];
        float* ob = p.out_bbox + o * 5;
        ob[0] = bx; ob[1] = by; ob[2] = bw; ob[3] = bh;
        ob[4] = __uint_as_float((unsigned)(s_comp[tid] >> 32));
        p.out_cls[o] = cls;
        if (p.out_keep) p.out_keep[o] = caseA ? tid : pos;
        // mask constants (postprocess.py:156-164)
        const float nW = (float)p.cfg.grid_w[s], nH = (float)p.cfg.grid_h[s];
        float* dp = p.det_par + o * 8;
        dp[0] = nW * bx;
        dp[1] = nH * by;
        dp[2] = (p.cfg.orien_thresh * bw) * nW;
        dp[3] = (p.cfg.orien_thresh * bh) * nH;
        dp[4] = (p.cfg.anchor_w[aid] / (float)p.cfg.image_w) * nW;      // grid_anchors, postprocess.py:21-25
        dp[5] = (p.cfg.anchor_h[aid] / (float)p.cfg.image_h) * nH;
        dp[6] = __int_as_float((p.a_off[s] + a) * 2);                   // first orientation channel
        dp[7] = __int_as_float(s);
    }
}

// ------------------------------------------------------------------------------------------------
// mask assembly: postprocess.py:69-72 (bilinear x4), :141-144 (get_orien_grid), :156-164 (predicate)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float bil_row(float v0, float v1, float w0, float w1) { return fmaf(v0, w0, v1 * w1); }

// The mask predicate (postprocess.py:156-164), |Px - cx| < tx && |Py - cy| < ty, WITHOUT floating-point compares.
// The obvious form compiles to 32 v_cmp_lt_f32_e64 per detection, each writing an SGPR pair, combined by s_and_b64 and
// expanded again by v_cndmask.  On MI355X that sequence returns wrong lane masks (the previous detection's) in a few waves per
// launch while waves of ANOTHER kernel issue gfx950's wide-K matrix instructions (v_mfma_f32_32x32x16_bf16 / _f16, 16x16x32_f16:
// e.g. this library's fp16 convolutions) on the same SIMD; compares through VCC and divergent branches are not affected
// (tools/hazard_probe, profiles/r02_experiments.md section 6).  Alone, or beside fp32 kernels, the compare form is exact.  The form
// below keeps every intermediate in VGPRs:
//   for non-negative floats (sign bit cleared; +inf and NaN included) the IEEE order is the order of the bit patterns as
//   integers, and a NaN pattern is larger than every number, so  a < t  <=>  bits(a) < bits(t)  for a number t >= 0;
//   a threshold that is negative, -0 or NaN can never be exceeded downwards: threshold_bits() maps it to 0 (a < 0 is false);
//   both patterns are below 2^31, so the unsigned difference has its top bit set exactly when bits(a) < bits(t).
// fp32 denormals are preserved in this library's kernels (.amdhsa_float_denorm_mode_32 3), like on the reference's CPU, so the
// integer order also agrees for them.  The subtraction is inline asm so that the compiler cannot turn it back into a compare.
//
//
// Round 3: six vector instructions per pixel and detection instead of nine (the kernel is bound by them: ~150 per 16-byte store)
// for every detection whose thresholds are FINITE.  For a number t >= 0,  |d| < t  <=>  the IEEE difference |d| - t is negative:
// a floating-point subtraction rounds monotonically and returns zero only for equal operands (denormal results are kept, see
// above), inf - t = +inf, and a NaN d gives a NaN whose sign bit is clear (|NaN| carries no sign) -- "not inside", like the
// compare.  So sign(|dx| - tx) AND sign(|dy| - ty) is the predicate: two subtractions with the absolute-value source modifier,
// one AND, and the sign shifted into a per-four-pixels bit register by v_alignbit.  Inline asm so that the compiler cannot turn
// the sign tests back into compares.  A threshold of +inf (a box size that overflowed) keeps the integer form above: inf - inf is
// a NaN with the sign SET on this hardware (measured: tests/test_hip_parity.py::test_mask_predicate_sign_form), which would
// read as "inside".
__device__ __forceinline__ unsigned threshold_bits(float t) {
    const unsigned u = __float_as_uint(t);
    return u <= 0x7f800000u ? u : 0u;
}
__device__ __forceinline__ unsigned inside_bit(unsigned ax, unsigned tx, unsigned ay, unsigned ty) {
    return (sub_u32_opaque(ax, tx) & sub_u32_opaque(ay, ty)) >> 31;
}
__device__ __forceinline__ unsigned abs_minus_bits(float d, float t) {      // bits(|d| - t)
    unsigned r;
    asm("v_sub_f32_e64 %0, |%1|, %2" : "=v"(r) : "v"(d), "v"(t));
    return r;
}
__device__ __forceinline__ unsigned shift_in_sign(unsigned acc, unsigned m) {      // (acc << 1) | (m >> 31)
    unsigned r;
    asm("v_alignbit_b32 %0, %1, %2, 31" : "=v"(r) : "v"(acc), "v"(m));
    return r;
}

__global__ __launch_bounds__(256) void post_mask_kernel(const PostParams p) {
    // blockIdx.y = (image, anchor field): the x4 bilinear up-sampling of an anchor's two orientation planes is shared by
    // every detection of that anchor (9 fields but up to 100 detections per image), so it is evaluated ONCE per block of
    // pixels and the detections of the field are looped over with only the predicate inside the loop.
    // One thread = a block of 16 x (up to) 4 output pixels that share their two source rows: output rows
    // 4j-2 .. 4j+1 interpolate between source rows j-1 and j (phases 0.125, 0.375, 0.625, 0.875), so the 24 loads
    // and the horizontal taps are done once per block instead of once per row.
    const int nfields = p.a_off[p.cfg.num_scales];
    const int b = blockIdx.y / nfields, field = blockIdx.y - b * nfields;
    // The detections of this (image, anchor field), compacted into LDS once per workgroup: {cx, cy, bits(tx), bits(ty)} and
    // the output slot k; the loop over them below touches no global memory but its own stores.
    //
    // blockIdx.z = a CHUNK of the field's detections (p.mask_chunk of them, in output-slot order): with few images in the batch a
    // field that holds half of an image's detections was 19 workgroups walking 50 detections each while the rest of the chip had
    // nothing to do (one image: 133 us).  The compaction is by ONE wave in slot order, so every workgroup of a field sees the
    // same list and the chunks partition it.
    __shared__ float4 s_det[SEL_MAXN];
    __shared__ int s_slot[SEL_MAXN];
    __shared__ float s_anchor[2];
    __shared__ int s_n;
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const int count = p.out_count[b];
        const float* dpar = p.det_par + (size_t)b * p.cfg.nms_post * 8;
        int base = 0;
        for (int k0 = 0; k0 < count; k0 += 64) {
            const int k = k0 + lane;
            const float* dp = dpar + k * 8;
            const bool mine = k < count && __float_as_int(dp[6]) == field * 2;
            const unsigned long long votes = __ballot(mine);
            if (mine) {
                const int slot = base + __popcll(votes & ((1ull << lane) - 1ull));
                s_det[slot] = make_float4(dp[0], dp[1], __uint_as_float(threshold_bits(dp[2])), __uint_as_float(threshold_bits(dp[3])));
                s_slot[slot] = k;
                if (slot == 0) {        // grid_anchors of the field: the same for every detection on it
                    s_anchor[0] = dp[4];
                    s_anchor[1] = dp[5];
                }
            }
            base += __popcll(votes);
        }
        if (lane == 0) s_n = base;
    }
    __syncthreads();
    const int i_first = blockIdx.z * p.mask_chunk;
    const int n_det = min(s_n, i_first + p.mask_chunk);
    if (i_first >= n_det) return;         // no detection of this image on this field (in this chunk)
    const int H = p.cfg.image_h, W = p.cfg.image_w;
    const int groups = W / MASK_PX;
    const int oh = H / 4, ow = W / 4;
    const int item = blockIdx.x * 256 + threadIdx.x;
    if (item >= (oh + 1) * groups) return;
    const int jy = item / groups, g = item - jy * groups;      // source rows jy-1 and jy
    const int s = (field >= p.a_off[1]) + (field >= p.a_off[2]);       // a_off is padded with the total for unused scales
    const float nW = (float)p.cfg.grid_w[s], nH = (float)p.cfg.grid_h[s];
    const float* px = p.oriens + ((size_t)b * nfields * 2 + field * 2) * oh * ow;
    const float* py = px + (size_t)oh * ow;

    // source rows (clamped at the borders exactly like torch: src >= 0, i1 = min(i0 + 1, n - 1))
    const int r0 = max(jy - 1, 0), r1 = min(jy, oh - 1);
    // six source columns 4g-1 .. 4g+4 cover the 16 outputs; clamp the loads at the borders
    float ax[2][6], ay[2][6];
    const int c_first = 4 * g - 1;
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        const int col = min(max(c_first + c, 0), ow - 1);
        ax[0][c] = px[(size_t)r0 * ow + col]; ax[1][c] = px[(size_t)r1 * ow + col];
        ay[0][c] = py[(size_t)r0 * ow + col]; ay[1][c] = py[(size_t)r1 * ow + col];
    }
    const bool left_edge = (g == 0);
    // horizontal taps of both source rows, both planes (row(y) = fmaf(v[x0], wx0, v[x1] * wx1))
    float hx[2][MASK_PX], hy[2][MASK_PX];
#pragma unroll
    for (int e = 0; e < MASK_PX; ++e) {
        const int ph = e & 3, j = e >> 2;
        const int c0 = j + (ph < 2 ? 0 : 1);
        float wx1 = ph == 0 ? 0.625f : ph == 1 ? 0.875f : ph == 2 ? 0.125f : 0.375f;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            float v0x, v1x, v0y, v1y;
            if (e < 2) {
                // outputs 0 and 1 of the first group clamp to src = 0: x0 = 0, weight of x1 = 0
                v0x = left_edge ? ax[r][1] : ax[r][0]; v1x = left_edge ? ax[r][2] : ax[r][1];
                v0y = left_edge ? ay[r][1] : ay[r][0]; v1y = left_edge ? ay[r][2] : ay[r][1];
            } else {
                v0x = ax[r][c0]; v1x = ax[r][c0 + 1];
                v0y = ay[r][c0]; v1y = ay[r][c0 + 1];
            }
            const float w1 = (e < 2 && left_edge) ? 0.0f : wx1;
            const float w0 = 1.0f - w1;
            hx[r][e] = fmaf(v0x, w0, v1x * w1);
            hy[r][e] = fmaf(v0y, w0, v1y * w1);
        }
    }
    // the (up to) four output rows of this block
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
        const int y = 4 * jy - 2 + q;
        if (y < 0 || y >= H) continue;
        // y = 4 jy - 2 + q: src = y/4 - 0.375 -> upper source row jy-1, weight of the lower row 0.125, 0.375,
        // 0.625, 0.875 for q = 0..3.  Rows 0 and 1 clamp to src = 0: weight 0 on the second row.
        float wy1 = q == 0 ? 0.125f : q == 1 ? 0.375f : q == 2 ? 0.625f : 0.875f;
        const bool top_edge = (y < 2);
        if (top_edge) wy1 = 0.0f;
        const float wy0 = 1.0f - wy1;
        const float base_y = ((float)y / (float)H) * nH;
        float vx[MASK_PX], vy[MASK_PX];
#pragma unroll
        for (int e = 0; e < MASK_PX; ++e) {
            // top edge (jy == 0): both source rows are row 0 and the second weight is 0, like torch's clamped tap
            const float tx0 = top_edge ? hx[1][e] : hx[0][e];
            const float bx0 = hx[1][e];
            const float ty0 = top_edge ? hy[1][e] : hy[0][e];
            const float by0 = hy[1][e];
            vx[e] = fmaf(tx0, wy0, bx0 * wy1);
            vy[e] = fmaf(ty0, wy0, by0 * wy1);
        }
        // P = (v * grid_anchor) / 2 + base (postprocess.py:142-143) depends on the ANCHOR only, not on the detection: once per
        // pixel; the loop over the field's detections keeps only the two |P - c| < t tests
        const float gax = s_anchor[0], gay = s_anchor[1];
        float Px[MASK_PX], Py[MASK_PX];
#pragma unroll
        for (int e = 0; e < MASK_PX; ++e) {
            const int x = g * MASK_PX + e;
            const float base_x = ((float)x / (float)W) * nW;
            Px[e] = (vx[e] * gax) / 2.0f + base_x;
            Py[e] = (vy[e] * gay) / 2.0f + base_y;
        }
        for (int i = i_first; i < n_det; ++i) {
            const float4 d = s_det[i];                                     // uniform: one LDS broadcast
            const int k = s_slot[i];
            const float cx = d.x, cy = d.y;
            const float tx = d.z, ty = d.w;      // threshold_bits: numbers >= 0 or +inf
            unsigned packed[4] = {0, 0, 0, 0};
            if (__float_as_uint(tx) == 0x7f800000u || __float_as_uint(ty) == 0x7f800000u) {      // uniform; see abs_minus_bits
#pragma unroll
                for (int e = 0; e < MASK_PX; ++e) {
                    const unsigned ax = __float_as_uint(Px[e] - cx) & 0x7fffffffu, ay = __float_as_uint(Py[e] - cy) & 0x7fffffffu;
                    packed[e >> 2] |= inside_bit(ax, __float_as_uint(tx), ay, __float_as_uint(ty)) << ((e & 3) * 8);
                }
            } else
#pragma unroll
            for (int w4 = 0; w4 < MASK_PX / 4; ++w4) {
                // inside = (|Px - cx| < tx) && (|Py - cy| < ty) by the signs of |d| - t (see abs_minus_bits); pixel 4 w4 + e -> bit e
                unsigned bits = 0;
#pragma unroll
                for (int e = 3; e >= 0; --e) {
                    const int px = 4 * w4 + e;
                    bits = shift_in_sign(bits, abs_minus_bits(Px[px] - cx, tx) & abs_minus_bits(Py[px] - cy, ty));
                }
                // bit e -> byte e: the four shifted copies of a 4-bit value do not overlap (e + 7 k are distinct), so no carries
                packed[w4] = __umul24(bits, 0x00204081u) & 0x01010101u;
            }
            uint4 o;
            o.x = packed[0]; o.y = packed[1]; o.z = packed[2]; o.w = packed[3];
            *reinterpret_cast<uint4*>(p.out_mask + (((size_t)b * p.cfg.nms_post + k) * H + y) * W + (size_t)g * MASK_PX) = o;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// standalone NMS (the reference's native export nms(dets, threshold) -> keep): one 1024-thread workgroup sorts,
// fills the bit-matrix in the caller's workspace, reduces it 64 rows at a time and compacts the survivors.
// Everything that grows with n lives in the workspace, so n is bounded only by NMS_MAXN (the reference has no bound).
// ------------------------------------------------------------------------------------------------
constexpr int NMS_MAXN = 1 << 16;

// om_postprocess_masks: the mask constants of detections chosen OUTSIDE the fused path (a caller-supplied NMS callable),
// postprocess.py:156-164 -- the same arithmetic as the end of post_select_kernel.
__global__ void post_detpar_kernel(const PostParams p, const float* dets, const int32_t* field) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.B * p.cfg.nms_post) return;
    const int b = i / p.cfg.nms_post, k = i - b * p.cfg.nms_post;
    float* dp = p.det_par + (size_t)i * 8;
    if (k >= p.out_count[b]) return;
    const int f = field[i];
    const int s = (f >= p.a_off[1]) + (f >= p.a_off[2]);
    const int a = f - p.a_off[s];
    const int aid = p.cfg.anchor_mask[s][a];
    const float bx = dets[(size_t)i * 5 + 0], by = dets[(size_t)i * 5 + 1], bw = dets[(size_t)i * 5 + 2], bh = dets[(size_t)i * 5 + 3];
    const float nW = (float)p.cfg.grid_w[s], nH = (float)p.cfg.grid_h[s];
    dp[0] = nW * bx;
    dp[1] = nH * by;
    dp[2] = (p.cfg.orien_thresh * bw) * nW;
    dp[3] = (p.cfg.orien_thresh * bh) * nH;
    dp[4] = (p.cfg.anchor_w[aid] / (float)p.cfg.image_w) * nW;
    dp[5] = (p.cfg.anchor_h[aid] / (float)p.cfg.image_h) * nH;
    dp[6] = __int_as_float(f * 2);
    dp[7] = __int_as_float(s);
}

struct NmsWs { size_t comp, x1, y1, x2, y2, area, keep_flag, keep_pos, mask, total; };

static NmsWs nms_ws_layout(int n) {
    NmsWs w;
    size_t n_pad = 64;
    while (n_pad < (size_t)n) n_pad <<= 1;
    const size_t words = ((size_t)n + 63) / 64;
    size_t off = 0;
    w.comp = off; off += align_up(n_pad * sizeof(unsigned long long), 256);
    w.x1 = off; off += align_up((size_t)n * 4, 256);
    w.y1 = off; off += align_up((size_t)n * 4, 256);
    w.x2 = off; off += align_up((size_t)n * 4, 256);
    w.y2 = off; off += align_up((size_t)n * 4, 256);
    w.area = off; off += align_up((size_t)n * 4, 256);
    w.keep_flag = off; off += align_up((size_t)n, 256);
    w.keep_pos = off; off += align_up((size_t)n, 256);
    w.mask = off; off += align_up((size_t)n * words * sizeof(unsigned long long), 256);
    w.total = off;
    return w;
}

struct NmsPtrs {
    unsigned long long* comp; float *x1, *y1, *x2, *y2, *area; unsigned char *keep_flag, *keep_pos; unsigned long long* mask;
};
static __host__ __device__ NmsPtrs nms_ptrs(char* ws, const NmsWs& L) {
    NmsPtrs q;
    q.comp = reinterpret_cast<unsigned long long*>(ws + L.comp);
    q.x1 = reinterpret_cast<float*>(ws + L.x1); q.y1 = reinterpret_cast<float*>(ws + L.y1);
    q.x2 = reinterpret_cast<float*>(ws + L.x2); q.y2 = reinterpret_cast<float*>(ws + L.y2);
    q.area = reinterpret_cast<float*>(ws + L.area);
    q.keep_flag = reinterpret_cast<unsigned char*>(ws + L.keep_flag);
    q.keep_pos = reinterpret_cast<unsigned char*>(ws + L.keep_pos);
    q.mask = reinterpret_cast<unsigned long long*>(ws + L.mask);
    return q;
}

// visiting order (score descending, ties by index ascending: torch's CPU sort is stable at these sizes, its CUDA sort leaves
// ties unspecified) + corners and areas in that order
__global__ __launch_bounds__(1024) void nms_sort_kernel(const float* dets, int n, int semantics, char* ws, const NmsWs L) {
    const NmsPtrs q = nms_ptrs(ws, L);
    const int tid = threadIdx.x;
    int n_pad = 64;
    while (n_pad < n) n_pad <<= 1;
    for (int i = tid; i < n_pad; i += 1024) {
        unsigned long long c = 0;
        if (i < n) {
            unsigned u = __float_as_uint(dets[(size_t)i * 5 + 4]);      // order-preserving map of the float score
            u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
            c = ((unsigned long long)u << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)i);
        }
        q.comp[i] = c;
    }
    __syncthreads();
    for (int k = 2; k <= n_pad; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < n_pad; i += 1024) {
                const int partner = i ^ j;
                if (partner > i) {
                    const unsigned long long a = q.comp[i], c = q.comp[partner];
                    const bool desc = (i & k) == 0;
                    if (desc ? (a < c) : (a > c)) { q.comp[i] = c; q.comp[partner] = a; }
                }
            }
            __syncthreads();
        }
    for (int i = tid; i < n; i += 1024) {
        const int pos = (int)(0xFFFFFFFFu - (unsigned)(q.comp[i] & 0xFFFFFFFFull));
        const float* d = dets + (size_t)pos * 5;
        const float cx = d[0], cy = d[1], w = d[2], h = d[3];
        const float x1 = cx - w / 2.0f, y1 = cy - h / 2.0f, x2 = cx + w / 2.0f, y2 = cy + h / 2.0f;
        q.x1[i] = x1; q.y1[i] = y1; q.x2[i] = x2; q.y2[i] = y2;
        q.area[i] = semantics == 1 ? w * h : (x2 - x1) * (y2 - y1);
        q.keep_pos[i] = 0;
    }
}

__global__ __launch_bounds__(256) void nms_mask_kernel(int n, float thr, int semantics, char* ws, const NmsWs L) {
    const NmsPtrs q = nms_ptrs(ws, L);
    const int words = (n + 63) >> 6;
    const long long item = (long long)blockIdx.x * 256 + threadIdx.x;
    if (item >= (long long)n * words) return;
    const int pi = (int)(item / words), wd = (int)(item - (long long)pi * words);
    q.mask[item] = nms_mask_word(q.x1, q.y1, q.x2, q.y2, q.area, n, thr, semantics == 1, pi, wd);
}

__global__ __launch_bounds__(1024) void nms_reduce_kernel(int n, int semantics, int64_t* keep, int32_t* n_keep, char* ws,
                                                          const NmsWs L) {
    __shared__ unsigned long long s_removed[NMS_MAXN / 64];
    __shared__ unsigned long long s_kept_bits;
    __shared__ int s_wave[16];
    const NmsPtrs q = nms_ptrs(ws, L);
    const int tid = threadIdx.x;
    const int words = (n + 63) >> 6;
    nms_reduce_blocked(q.mask, n, words, s_removed, &s_kept_bits, q.keep_flag);
    __syncthreads();
    // ---- output order: ascending original index (nms_cpu.cpp:62) or visiting order (nms_kernel.cu:136-139)
    if (semantics == 0) {
        for (int i = tid; i < n; i += 1024)
            if (q.keep_flag[i]) q.keep_pos[(int)(0xFFFFFFFFu - (unsigned)(q.comp[i] & 0xFFFFFFFFull))] = 1;
        __syncthreads();
    }
    int running = 0;
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        const int flag = i < n ? (semantics == 0 ? q.keep_pos[i] : q.keep_flag[i]) : 0;
        int total;
        const int rank = block_scan_excl(flag, s_wave, total);
        if (flag) keep[running + rank] = semantics == 0 ? i : (int)(0xFFFFFFFFu - (unsigned)(q.comp[i] & 0xFFFFFFFFull));
        running += total;
    }
    if (tid == 0) *n_keep = running;
}

// unit-test entry: the reference-exact elementary functions of ref_math.h on a flat array
__global__ void ref_math_kernel(const float* x, long long n, int func, int C, float* y) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    float r;
    switch (func) {
        case 0: r = expf_glibc(v); break;
        case 1: r = expf_sleef(v); break;
        case 2: r = sigmoid_scalar_ref(v); break;
        case 3: r = sigmoid_vector_ref(v); break;
        case 4: r = sigmoid_class_ref(v, (int)(i % C), C); break;      // x is [rows][C] class logits
        case 6: r = __uint_as_float(abs_minus_bits(v, x[i ^ 1])); break;   // the mask predicate's |x[i]| - x[i ^ 1], bit pattern returned
        default: r = expf_cr(v); break;
    }
    y[i] = r;
}

static int fill_params(const om_post_cfg* cfg, PostParams& p) {
    OM_REQUIRE(cfg->num_scales >= 1 && cfg->num_scales <= OM_MAX_SCALES, OM_EINVAL, "postprocess: %d scales (1..%d supported)",
               cfg->num_scales, OM_MAX_SCALES);
    int a_max = 0;
    for (int s = 0; s < OM_MAX_SCALES; ++s) {
        const int a = s >= cfg->num_scales ? 0 : cfg->anchors_of_scale[s] > 0 ? cfg->anchors_of_scale[s] : cfg->anchors_per_scale;
        OM_REQUIRE(s >= cfg->num_scales || (a >= 1 && a <= 3), OM_EINVAL, "postprocess: scale %d has %d anchors (1..3 supported)", s, a);
        p.a_cnt[s] = a;
        a_max = a > a_max ? a : a_max;
    }
    p.a_off[0] = 0;
    for (int s = 0; s < OM_MAX_SCALES; ++s) p.a_off[s + 1] = p.a_off[s] + p.a_cnt[s];
    p.cand_dets = nullptr; p.cand_cls = nullptr; p.cand_field = nullptr;
    OM_REQUIRE(cfg->image_h % 32 == 0 && cfg->image_w % 32 == 0 && cfg->image_h > 0 && cfg->image_w > 0, OM_EINVAL,
               "postprocess: image %dx%d must be a multiple of 32", cfg->image_h, cfg->image_w);
    OM_REQUIRE(cfg->nms_pre >= 1 && cfg->nms_pre <= SEL_MAXN && cfg->nms_post >= 1 && cfg->nms_post <= cfg->nms_pre,
               OM_EINVAL, "postprocess: nms_pre=%d (max %d), nms_post=%d", cfg->nms_pre, SEL_MAXN, cfg->nms_post);
    OM_REQUIRE(cfg->num_classes >= 1 && a_max * (5 + cfg->num_classes) <= cfg->bbox_pix_stride,
               OM_EINVAL, "postprocess: bbox_pix_stride=%d too small", cfg->bbox_pix_stride);
    OM_REQUIRE(cfg->conf_thresh >= 0.0f, OM_EINVAL, "postprocess: conf_thresh must be >= 0");
    OM_REQUIRE(cfg->nms_semantics == 0 || cfg->nms_semantics == 1, OM_EINVAL,
               "postprocess: nms_semantics=%d (0 = the reference's CPU backend, 1 = its CUDA backend)", cfg->nms_semantics);
    p.cfg = *cfg;
    p.cand_off[0] = 0;
    for (int s = 0; s < OM_MAX_SCALES; ++s) {
        if (s >= cfg->num_scales) {          // an unused scale: no candidates, a 1 x 1 grid that nothing ever addresses
            p.cfg.grid_h[s] = p.cfg.grid_w[s] = 1;
            p.cand_off[s + 1] = p.cand_off[s];
            continue;
        }
        OM_REQUIRE(cfg->grid_h[s] > 0 && cfg->grid_w[s] > 0, OM_EINVAL, "postprocess: bad grid");
        for (int a = 0; a < p.a_cnt[s]; ++a)
            OM_REQUIRE(cfg->anchor_mask[s][a] >= 0 && cfg->anchor_mask[s][a] < OM_MAX_ANCHORS, OM_EINVAL,
                       "postprocess: bad anchor_mask");
        p.cand_off[s + 1] = p.cand_off[s] + p.a_cnt[s] * cfg->grid_h[s] * cfg->grid_w[s];
    }
    p.ncand = p.cand_off[OM_MAX_SCALES];
    const long long npairs = (long long)p.ncand * cfg->num_classes;
    OM_REQUIRE(npairs < (1ll << 30), OM_EINVAL, "postprocess: too many (candidate, class) pairs");
    p.npairs = (int)npairs;
    p.ntiles = (p.npairs + DEC_TILE - 1) / DEC_TILE;
    {      // post_decode_kernel: a sweep visits (candidates of a tile) x (vectorised or scalar classes) items, DEC_VISITS per thread
        const int C = cfg->num_classes, CV = C & ~31, CS = C - CV;
        const int items = ((DEC_TILE - 1) / C + 2) * (CV > CS ? CV : CS) + 63;
        OM_REQUIRE(items <= 256 * DEC_VISITS_MANY, OM_EINVAL, "postprocess: num_classes=%d needs more than %d visits per decode thread", C,
                   DEC_VISITS_MANY);
        p.dec_visits = items <= 256 * DEC_VISITS ? DEC_VISITS : DEC_VISITS_MANY;
    }
    return OM_OK;
}

struct PostWs { size_t keys, tile_count, hist1, list, det_par, nms_mask, total; };

static PostWs post_ws_layout(const PostParams& p, int B) {
    PostWs w;
    size_t off = 0;
    w.keys = off; off += align_up((size_t)B * p.ntiles * DEC_TILE * sizeof(unsigned), 256);
    w.tile_count = off; off += align_up((size_t)B * p.ntiles * sizeof(int), 256);
    w.hist1 = off; off += align_up(((size_t)B * L1_BINS + B) * sizeof(unsigned), 256);      // + list_count[B], zeroed together
    w.list = off; off += align_up((size_t)B * SEL_LIST_MAX * sizeof(unsigned long long), 256);
    w.det_par = off; off += align_up((size_t)B * p.cfg.nms_post * 8 * sizeof(float), 256);
    w.nms_mask = off;
    if (p.cfg.nms_pre > SEL_LDS_MASK_N) off += align_up((size_t)B * SEL_MAXN * (SEL_MAXN / 64) * sizeof(unsigned long long), 256);
    w.total = off;
    return w;
}

static void launch_post_decode(const PostParams& p, int B, hipStream_t stream) {
    if (p.dec_visits == DEC_VISITS) hipLaunchKernelGGL(post_decode_kernel<DEC_VISITS>, dim3(p.ntiles, B), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(post_decode_kernel<DEC_VISITS_MANY>, dim3(p.ntiles, B), dim3(256), 0, stream, p);
}

}  // namespace om

extern "C" {

size_t om_postprocess_workspace_bytes(const om_post_cfg* cfg, int B) {
    if (!cfg || B <= 0) return 0;
    om::PostParams p;
    if (om::fill_params(cfg, p) != OM_OK) return 0;
    return om::post_ws_layout(p, B).total;
}

// shared by the three entries: parameter block, workspace pointers, size checks
static int post_setup(const om_post_cfg* cfg, const float* bbox32, const float* bbox16, const float* bbox8, const float* oriens, int B,
                      void* workspace, size_t ws_bytes, om::PostParams& p, const char* who) {
    OM_REQUIRE(cfg && workspace, OM_EINVAL, "%s: null argument", who);
    OM_REQUIRE(B > 0 && B < 65536, OM_EINVAL, "%s: B=%d", who, B);
    int rc = om::fill_params(cfg, p);
    if (rc != OM_OK) return rc;
    const om::PostWs w = om::post_ws_layout(p, B);
    OM_REQUIRE(ws_bytes >= w.total, OM_ENOMEM, "%s: workspace %zu bytes < %zu needed", who, ws_bytes, w.total);
    OM_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, OM_EINVAL, "%s: workspace must be 256-byte aligned", who);
    OM_REQUIRE((long long)B * p.a_off[cfg->num_scales] < 65536, OM_EINVAL,
               "%s: B * anchor fields must be < 65536 (the mask kernel's grid.y)", who);
    for (int sc = 0; sc < cfg->num_scales; ++sc)      // post_decode_kernel addresses the heads with 24-bit multiplies and 32-bit element offsets
        OM_REQUIRE((long long)B * cfg->grid_h[sc] * cfg->grid_w[sc] < (1ll << 24) && cfg->bbox_pix_stride < (1 << 24) &&
                       (long long)B * cfg->grid_h[sc] * cfg->grid_w[sc] * cfg->bbox_pix_stride < (1ll << 31),
                   OM_EINVAL, "%s: B=%d x grid %dx%d x pixel stride %d is beyond the decode kernel's 32-bit offsets", who, B,
                   cfg->grid_h[sc], cfg->grid_w[sc], cfg->bbox_pix_stride);
    char* base = static_cast<char*>(workspace);
    const float* heads[3] = {bbox32, bbox16, bbox8};
    for (int sc = 0; sc < 3; ++sc) p.bbox[sc] = sc < cfg->num_scales ? heads[sc] : heads[0];     // unused scales are never addressed
    p.oriens = oriens; p.B = B;
    p.keys = reinterpret_cast<unsigned*>(base + w.keys);
    p.tile_count = reinterpret_cast<int*>(base + w.tile_count);
    p.hist1 = reinterpret_cast<unsigned*>(base + w.hist1);
    p.list_count = p.hist1 + (size_t)B * om::L1_BINS;
    p.list = reinterpret_cast<unsigned long long*>(base + w.list);
    p.det_par = reinterpret_cast<float*>(base + w.det_par);
    p.nms_mask = reinterpret_cast<unsigned long long*>(base + w.nms_mask);
    p.out_bbox = nullptr; p.out_cls = nullptr; p.out_mask = nullptr; p.out_count = nullptr; p.out_keep = nullptr;
    return OM_OK;
}

static int launch_post_mask(const om_post_cfg* cfg, const om::PostParams& p, int B, hipStream_t stream) {
    const int items = (cfg->image_h / 4 + 1) * (cfg->image_w / om::MASK_PX);
    // detections of a field in chunks of their own workgroups while the batch alone does not fill the chip (post_mask_kernel)
    const int fields = p.a_off[cfg->num_scales];
    om::PostParams q = p;
    q.mask_chunk = (long long)((items + 255) / 256) * B * fields >= 2048 ? cfg->nms_post : 8;
    const int chunks = (cfg->nms_post + q.mask_chunk - 1) / q.mask_chunk;
    hipLaunchKernelGGL(om::post_mask_kernel, dim3((items + 255) / 256, B * fields, chunks), dim3(256), 0, stream, q);
    OM_CHECK_HIP(hipGetLastError());
    return OM_OK;
}

int om_postprocess_detect(const om_post_cfg* cfg, const float* bbox32, const float* bbox16, const float* bbox8, int B,
                          float* out_bbox, int64_t* out_cls, int32_t* out_count, int32_t* out_keep, void* workspace, size_t ws_bytes,
                          om_stream stream_) {
    OM_REQUIRE(cfg && bbox32 && (cfg->num_scales < 2 || bbox16) && (cfg->num_scales < 3 || bbox8) && out_bbox && out_cls && out_count &&
                   workspace,
               OM_EINVAL, "om_postprocess_detect: null argument");
    om::PostParams p;
    if (int rc = post_setup(cfg, bbox32, bbox16, bbox8, nullptr, B, workspace, ws_bytes, p, "om_postprocess_detect")) return rc;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    p.out_bbox = out_bbox; p.out_cls = out_cls; p.out_count = out_count; p.out_keep = out_keep;
    if (int rc = om::launch_zero_words(p.hist1, (size_t)B * om::L1_BINS + B, stream)) return rc;
    om::launch_post_decode(p, B, stream);
    OM_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(om::post_select_kernel, dim3(B), dim3(om::SEL_THREADS), 0, stream, p);
    OM_CHECK_HIP(hipGetLastError());
    return OM_OK;
}

int om_postprocess_assemble(const om_post_cfg* cfg, const float* oriens, int B, const int32_t* out_count, uint8_t* out_mask,
                            void* workspace, size_t ws_bytes, om_stream stream_) {
    OM_REQUIRE(cfg && oriens && out_count && out_mask && workspace, OM_EINVAL, "om_postprocess_assemble: null argument");
    OM_REQUIRE((reinterpret_cast<uintptr_t>(out_mask) & 15) == 0, OM_EINVAL, "om_postprocess_assemble: out_mask must be 16-byte aligned");
    om::PostParams p;
    if (int rc = post_setup(cfg, oriens, oriens, oriens, oriens, B, workspace, ws_bytes, p, "om_postprocess_assemble")) return rc;   // (heads: not read)
    p.out_mask = out_mask; p.out_count = const_cast<int32_t*>(out_count);
    return launch_post_mask(cfg, p, B, static_cast<hipStream_t>(stream_));
}

int om_postprocess(const om_post_cfg* cfg, const float* bbox32, const float* bbox16, const float* bbox8,
                   const float* oriens, int B, float* out_bbox, int64_t* out_cls, uint8_t* out_mask,
                   int32_t* out_count, int32_t* out_keep, void* workspace, size_t ws_bytes, om_stream stream) {
    OM_REQUIRE(oriens && out_mask, OM_EINVAL, "om_postprocess: null argument");
    OM_REQUIRE((reinterpret_cast<uintptr_t>(out_mask) & 15) == 0, OM_EINVAL, "om_postprocess: out_mask must be 16-byte aligned");
    if (int rc = om_postprocess_detect(cfg, bbox32, bbox16, bbox8, B, out_bbox, out_cls, out_count, out_keep, workspace, ws_bytes, stream))
        return rc;
    return om_postprocess_assemble(cfg, oriens, B, out_count, out_mask, workspace, ws_bytes, stream);
}

int om_postprocess_candidates(const om_post_cfg* cfg, const float* bbox32, const float* bbox16, const float* bbox8, int B,
                              float* cand_dets, int64_t* cand_cls, int32_t* cand_field, int32_t* cand_count, void* workspace,
                              size_t ws_bytes, om_stream stream_) {
    OM_REQUIRE(cfg && bbox32 && (cfg->num_scales < 2 || bbox16) && (cfg->num_scales < 3 || bbox8) && cand_dets && cand_cls &&
                   cand_field && cand_count && workspace,
               OM_EINVAL, "om_postprocess_candidates: null argument");
    om::PostParams p;
    if (int rc = post_setup(cfg, bbox32, bbox16, bbox8, nullptr, B, workspace, ws_bytes, p, "om_postprocess_candidates")) return rc;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    p.cand_dets = cand_dets; p.cand_cls = cand_cls; p.cand_field = cand_field; p.out_count = cand_count;
    if (int rc = om::launch_zero_words(p.hist1, (size_t)B * om::L1_BINS + B, stream)) return rc;
    om::launch_post_decode(p, B, stream);
    OM_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(om::post_select_kernel, dim3(B), dim3(om::SEL_THREADS), 0, stream, p);
    OM_CHECK_HIP(hipGetLastError());
    return OM_OK;
}

int om_postprocess_masks(const om_post_cfg* cfg, const float* oriens, int B, const float* dets, const int32_t* field,
                         const int32_t* count, uint8_t* out_mask, void* workspace, size_t ws_bytes, om_stream stream_) {
    OM_REQUIRE(cfg && oriens && dets && field && count && out_mask && workspace, OM_EINVAL, "om_postprocess_masks: null argument");
    OM_REQUIRE((reinterpret_cast<uintptr_t>(out_mask) & 15) == 0, OM_EINVAL, "om_postprocess_masks: out_mask must be 16-byte aligned");
    om::PostParams p;
    if (int rc = post_setup(cfg, dets, dets, dets, oriens, B, workspace, ws_bytes, p, "om_postprocess_masks")) return rc;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    p.out_mask = out_mask; p.out_count = const_cast<int32_t*>(count);
    const int total = B * cfg->nms_post;
    hipLaunchKernelGGL(om::post_detpar_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, p, dets, field);
    OM_CHECK_HIP(hipGetLastError());
    return launch_post_mask(cfg, p, B, stream);
}

int om_post_kernel_occupancy(int which, int* threads, int* vgprs, int* lds_bytes, int* max_blocks_per_cu) {
    OM_REQUIRE(threads && vgprs && lds_bytes && max_blocks_per_cu, OM_EINVAL, "om_post_kernel_occupancy: null argument");
    OM_REQUIRE(which >= 0 && which <= 2, OM_EINVAL, "om_post_kernel_occupancy: which=%d (0 decode, 1 select, 2 mask)", which);
    const void* fn = which == 0 ? reinterpret_cast<const void*>(om::post_decode_kernel<om::DEC_VISITS>)
                   : which == 1 ? reinterpret_cast<const void*>(om::post_select_kernel)
                                : reinterpret_cast<const void*>(om::post_mask_kernel);
    const int nthreads = which == 1 ? om::SEL_THREADS : 256;
    hipFuncAttributes attr;
    OM_CHECK_HIP(hipFuncGetAttributes(&attr, fn));
    int nb = 0;
    OM_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, nthreads, 0));
    *threads = nthreads; *vgprs = attr.numRegs; *lds_bytes = (int)attr.sharedSizeBytes; *max_blocks_per_cu = nb;
    return OM_OK;
}

int om_ref_math(const float* x, long long n, int func, int num_classes, float* y, om_stream stream_) {
    OM_REQUIRE(x && y && n >= 0 && func >= 0 && func <= 6 && num_classes >= 1 && (func != 6 || n % 2 == 0), OM_EINVAL,
               "om_ref_math: bad argument");
    if (n == 0) return OM_OK;
    hipLaunchKernelGGL(om::ref_math_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream_), x, n,
                       func, num_classes, y);
    OM_CHECK_HIP(hipGetLastError());
    return OM_OK;
}

size_t om_nms_workspace_bytes(int n) {
    if (n <= 0) return 256;
    if (n > om::NMS_MAXN) return 0;
    return om::nms_ws_layout(n).total;
}

int om_nms_ex(const float* dets, int n, float thresh, int semantics, int64_t* keep, int32_t* n_keep, void* workspace,
              size_t ws_bytes, om_stream stream_) {
    OM_REQUIRE(n_keep && (n == 0 || (dets && keep && workspace)), OM_EINVAL, "om_nms: null argument");
    OM_REQUIRE(n >= 0 && n <= om::NMS_MAXN, OM_EINVAL, "om_nms: n=%d, at most %d boxes supported", n, om::NMS_MAXN);
    OM_REQUIRE(semantics == 0 || semantics == 1, OM_EINVAL, "om_nms: semantics=%d (0 = CPU backend, 1 = CUDA backend)", semantics);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (n == 0) return om::launch_zero_words(n_keep, 1, stream);
    const om::NmsWs L = om::nms_ws_layout(n);
    OM_REQUIRE(ws_bytes >= L.total, OM_ENOMEM, "om_nms: workspace %zu bytes < %zu needed", ws_bytes, L.total);
    OM_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, OM_EINVAL, "om_nms: workspace must be 256-byte aligned");
    char* ws = static_cast<char*>(workspace);
    hipLaunchKernelGGL(om::nms_sort_kernel, dim3(1), dim3(1024), 0, stream, dets, n, semantics, ws, L);
    OM_CHECK_HIP(hipGetLastError());
    const long long items = (long long)n * ((n + 63) / 64);
    hipLaunchKernelGGL(om::nms_mask_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, stream, n, thresh, semantics, ws, L);
    OM_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(om::nms_reduce_kernel, dim3(1), dim3(1024), 0, stream, n, semantics, keep, n_keep, ws, L);
    OM_CHECK_HIP(hipGetLastError());
    return OM_OK;
}

int om_nms(const float* dets, int n, float thresh, int64_t* keep, int32_t* n_keep, void* workspace, size_t ws_bytes,
           om_stream stream) {
    return om_nms_ex(dets, n, thresh, 0, keep, n_keep, workspace, ws_bytes, stream);
}

}  // extern "C"
