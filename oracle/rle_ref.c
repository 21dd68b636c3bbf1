/* CPU oracle for the COCO RLE string.  TEST INFRASTRUCTURE ONLY -- never linked into or called from the product.
 *
 * The reference's eval/coco_eval.py:120-122 hands masks to pycocotools.mask.encode; pycocotools is a third-party
 * dependency of the reference (requirements.txt:5, no version pinned) that is NOT present in /root/reference or in
 * this image, so its output cannot be generated here.  This file restates the PUBLISHED algorithm of
 * pycocotools' common/maskApi.c (cocodataset/cocoapi, v2.0) independently of the product's Python implementation:
 *   rleEncode     column-major run lengths, the first run counts zeros (may be 0)
 *   rleToString   "similar to LEB128 but using 6 bits/char and ascii chars 48-111": from the fourth run on the value
 *                 written is the difference to the run two back; 5 data bits per char, bit 0x20 = continuation,
 *                 sign-extended from bit 0x10
 *   rleFrString   the inverse
 * Parity of the STRING stays "unpinned" (no pycocotools-produced vector exists offline); the run lengths themselves are
 * pinned by the reference-generated resized masks (tests/golden/coco_format.npz).
 */
#include <stdint.h>
#include <stddef.h>

/* mask: h*w bytes (0/1), COLUMN-major as pycocotools expects (Fortran order).  cnts receives the runs; returns their number. */
long rle_ref_encode(const uint8_t* mask, long h, long w, uint32_t* cnts) {
    const long a = h * w;
    long k = 0;
    uint32_t c = 0;
    uint8_t p = 0;
    for (long j = 0; j < a; ++j) {
        if (mask[j] != p) { cnts[k++] = c; c = 0; p = mask[j]; }
        ++c;
    }
    cnts[k++] = c;
    return k;
}

/* s must hold 6 * m + 1 chars; returns the string length */
long rle_ref_to_string(const uint32_t* cnts, long m, char* s) {
    long p = 0;
    for (long i = 0; i < m; ++i) {
        long x = (long)cnts[i];
        if (i > 2) x -= (long)cnts[i - 2];
        int more = 1;
        while (more) {
            char c = (char)(x & 0x1f);
            x >>= 5;
            more = (c & 0x10) ? x != -1 : x != 0;
            if (more) c |= 0x20;
            c += 48;
            s[p++] = c;
        }
    }
    s[p] = 0;
    return p;
}

/* returns the number of runs decoded from s */
long rle_ref_from_string(const char* s, uint32_t* cnts) {
    long m = 0, p = 0;
    while (s[p]) {
        long x = 0;
        int k = 0, more = 1;
        while (more) {
            const char c = (char)(s[p] - 48);
            x |= (long)(c & 0x1f) << (5 * k);
            more = c & 0x20;
            ++p; ++k;
            if (!more && (c & 0x10)) x |= -1L << (5 * k);
        }
        if (m > 2) x += (long)cnts[m - 2];
        cnts[m++] = (uint32_t)x;
    }
    return m;
}
