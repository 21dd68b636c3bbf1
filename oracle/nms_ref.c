/* CPU oracle: greedy IoU suppression.  TEST INFRASTRUCTURE ONLY -- never linked into or
 * called from the product library.
 *
 * Plain-C restatement of nms_cpu_kernel<float>, /root/reference/eval/src/nms_cpu.cpp:4-63:
 *   - corners  x1 = cx - w/2, x2 = cx + w/2 (and y)            nms_cpu.cpp:17-20
 *   - area     (x2 - x1) * (y2 - y1)                           nms_cpu.cpp:22
 *   - visit boxes in descending-score order (order[] is the caller's argsort, :24)
 *   - suppress j when inter / (area_i + area_j - inter) >= threshold   (non-strict, :59)
 *   - the caller returns the un-suppressed ORIGINAL indices ascending   (:62)
 * Build with -ffp-contract=off so every operation rounds once, like the tensor ops and
 * scalar loop of the reference.
 */
#include <stdint.h>
#include <stdlib.h>

int nms_ref_f32(const float* dets, const int64_t* order, int n, float threshold, uint8_t* keep_flag) {
    if (n <= 0) return 0;
    float* buf = (float*)malloc(sizeof(float) * 5 * (size_t)n);
    if (!buf) return -1;
    float *x1 = buf, *y1 = buf + n, *x2 = buf + 2 * n, *y2 = buf + 3 * n, *area = buf + 4 * n;
    for (int i = 0; i < n; ++i) {
        const float cx = dets[5 * i + 0], cy = dets[5 * i + 1], w = dets[5 * i + 2], h = dets[5 * i + 3];
        x1[i] = cx - w / 2.0f;
        y1[i] = cy - h / 2.0f;
        x2[i] = cx + w / 2.0f;
        y2[i] = cy + h / 2.0f;
        area[i] = (x2[i] - x1[i]) * (y2[i] - y1[i]);
        keep_flag[i] = 1;
    }
    int kept = 0;
    for (int pi = 0; pi < n; ++pi) {
        const int64_t i = order[pi];
        if (!keep_flag[i]) continue;
        ++kept;
        for (int pj = pi + 1; pj < n; ++pj) {
            const int64_t j = order[pj];
            if (!keep_flag[j]) continue;
            const float xx1 = x1[i] > x1[j] ? x1[i] : x1[j];
            const float yy1 = y1[i] > y1[j] ? y1[i] : y1[j];
            const float xx2 = x2[i] < x2[j] ? x2[i] : x2[j];
            const float yy2 = y2[i] < y2[j] ? y2[i] : y2[j];
            float w = xx2 - xx1; if (!(w > 0.0f)) w = 0.0f;
            float h = yy2 - yy1; if (!(h > 0.0f)) h = 0.0f;
            const float inter = w * h;
            const float ovr = inter / (area[i] + area[j] - inter);
            if (ovr >= threshold) keep_flag[j] = 0;
        }
    }
    free(buf);
    return kept;
}
