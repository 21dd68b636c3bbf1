/* CPU oracle for the reference's CUDA NMS backend.  TEST INFRASTRUCTURE ONLY -- never linked into or called
 * from the product library.
 *
 * Plain-C restatement of /root/reference/eval/src/nms_kernel.cu (which cannot be built anywhere today: it
 * includes THC headers that torch >= 1.11 no longer ships).  The algorithm is deterministic and
 * host-restatable:
 *   devIoU      :13-23   left/right/top/bottom from a[0] -+ a[2]/2 (a = cx, cy, w, h), areas Sa = a[2]*a[3],
 *                        IoU = interS / (Sa + Sb - interS)
 *   nms_kernel  :25-69   one 64-bit word per (row, 64-column block): bit i set when devIoU(row, col) > thresh
 *                        (STRICT), only columns after the row inside the diagonal block (:59-61)
 *   nms_cuda    :114-132 host loop over the rows in score-descending order: a row whose bit is not yet in remv[]
 *                        is kept and ORs its words into remv[]
 *               :136-139 returns order[keep]: ORIGINAL indices in score-descending (visiting) order
 * The boxes arrive already sorted (`sorted5`), as boxes_sorted at :76-78; the sort itself (torch's CUDA sort,
 * ties unspecified) is the caller's.  Every operation rounds once (build with -ffp-contract=off); nvcc's default
 * -fmad=true MAY fuse `Sa + Sb - width*height` on a real build -- that choice is the compiler's, not the source's.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static float dev_iou(const float* a, const float* b) {
    const float al = a[0] - a[2] / 2, bl = b[0] - b[2] / 2;
    const float ar = a[0] + a[2] / 2, br = b[0] + b[2] / 2;
    const float at = a[1] - a[3] / 2, bt = b[1] - b[3] / 2;
    const float ab = a[1] + a[3] / 2, bb = b[1] + b[3] / 2;
    const float left = al > bl ? al : bl, right = ar < br ? ar : br;
    const float top = at > bt ? at : bt, bottom = ab < bb ? ab : bb;
    float width = right - left, height = bottom - top;
    if (!(width > 0.f)) width = 0.f;
    if (!(height > 0.f)) height = 0.f;
    const float inter = width * height;
    const float sa = a[2] * a[3], sb = b[2] * b[3];
    return inter / (sa + sb - inter);
}

/* sorted5: [n][5] boxes in visiting order.  keep_sorted receives the kept POSITIONS (in visiting order);
 * returns their number. */
int nms_cuda_ref_f32(const float* sorted5, int n, float thresh, int64_t* keep_sorted) {
    if (n <= 0) return 0;
    const int tpb = 64;
    const int col_blocks = (n + tpb - 1) / tpb;
    uint64_t* mask = (uint64_t*)calloc((size_t)n * col_blocks, sizeof(uint64_t));
    uint64_t* remv = (uint64_t*)calloc((size_t)col_blocks, sizeof(uint64_t));
    if (!mask || !remv) { free(mask); free(remv); return -1; }
    for (int row_start = 0; row_start < col_blocks; ++row_start)
        for (int col_start = 0; col_start < col_blocks; ++col_start) {
            const int row_size = n - row_start * tpb < tpb ? n - row_start * tpb : tpb;
            const int col_size = n - col_start * tpb < tpb ? n - col_start * tpb : tpb;
            for (int t = 0; t < row_size; ++t) {
                const int cur = tpb * row_start + t;
                uint64_t bits = 0;
                const int start = row_start == col_start ? t + 1 : 0;
                for (int i = start; i < col_size; ++i)
                    if (dev_iou(sorted5 + 5 * (size_t)cur, sorted5 + 5 * (size_t)(tpb * col_start + i)) > thresh)
                        bits |= 1ULL << i;
                mask[(size_t)cur * col_blocks + col_start] = bits;
            }
        }
    int num_to_keep = 0;
    for (int i = 0; i < n; ++i) {
        const int nblock = i / tpb, inblock = i % tpb;
        if (!(remv[nblock] & (1ULL << inblock))) {
            keep_sorted[num_to_keep++] = i;
            const uint64_t* p = mask + (size_t)i * col_blocks;
            for (int j = nblock; j < col_blocks; ++j) remv[j] |= p[j];
        }
    }
    free(mask);
    free(remv);
    return num_to_keep;
}
