// Minimal reproducer (gfx950 / MI355X, ROCm 7.2) of what DESIGN.md 3.2 called "the compare hazard": it is not in the compares.
//   v_pk_add_f32 with a source-half selection on a REGISTER operand (op_sel:[0,1] or op_sel_hi:[1,0] on src1) returns wrong sums
//   in some lanes while ANOTHER wave on the same SIMD issues wide-K matrix instructions (v_mfma_f32_32x32x16_{bf16,f16},
//   v_mfma_f32_16x16x32_f16).  The same instruction without op_sel, scalar v_sub_f32, and neighbours issuing v_mfma_f32_32x32x2_f32
//   are exact.  hipcc emits this form whenever it packs (x - c.x, y - c.x) / (x - c.y, y - c.y) with (c.x, c.y) in one register pair.
// Build + run:  hipcc --offload-arch=gfx950 -O3 pk_opsel_repro.hip -o pk_opsel_repro && ./pk_opsel_repro
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int SEL>      // 0: no op_sel (c.x from a broadcast pair), 1: op_sel_hi:[1,0] (both lanes take c.x), 2: op_sel:[0,1] (both take c.y)
__global__ void victim(const float2* p, const float2* c, unsigned* wrong, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const float2 a = p[t & 65535];
    unsigned bad = 0;
    for (int i = 0; i < iters; ++i) {
        const float2 cc = c[i & 255];
        const float2 bx = {cc.x, cc.x};
        float2 r;
        if (SEL == 0) asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(bx));
        if (SEL == 1) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(cc));
        if (SEL == 2) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(cc));
        const float s = SEL == 2 ? cc.y : cc.x;
        float e0, e1;       // the expectation by scalar subtractions (opaque: never packed)
        asm volatile("v_sub_f32 %0, %2, %4\n\tv_sub_f32 %1, %3, %4" : "=&v"(e0), "=&v"(e1) : "v"(a.x), "v"(a.y), "v"(s));
        bad += (__float_as_uint(r.x) != __float_as_uint(e0)) + (__float_as_uint(r.y) != __float_as_uint(e1));
    }
    if (bad) atomicAdd(wrong, bad);
}
template <int KIND>     // the neighbour: nothing but matrix instructions on registers
__global__ void neighbour(float* sink, int iters) {
    f32x16 acc = {};
    bf16x8 x, y;
    for (int k = 0; k < 8; ++k) { x[k] = (__bf16)(threadIdx.x & 7); y[k] = (__bf16)(1.f / (1 + k)); }
    for (int i = 0; i < iters; ++i)
        acc = KIND ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc, 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x2f32((float)x[0], 0.5f, acc, 0, 0, 0);
    if (acc[3] == 12345.f) sink[0] = acc[0];
}
int main() {
    float2 *p, *c; unsigned* w; float* sink;
    hipMalloc(&p, 65536 * 8); hipMalloc(&c, 256 * 8); hipMalloc(&w, 4); hipMalloc(&sink, 16);
    float2 hp[65536], hc[256];
    for (int i = 0; i < 65536; ++i) hp[i] = {(float)((i * 2654435761u) >> 8) * 1e-4f, (float)((i * 40503u) & 0xffff) * 3e-3f};
    for (int i = 0; i < 256; ++i) hc[i] = {1.5f + i * 0.37f, 7.25f - i * 0.11f};
    hipMemcpy(p, hp, sizeof(hp), hipMemcpyHostToDevice); hipMemcpy(c, hc, sizeof(hc), hipMemcpyHostToDevice);
    hipStream_t s0, s1; hipStreamCreate(&s0); hipStreamCreate(&s1);
    const char* names[3] = {"no op_sel", "op_sel_hi:[1,0]", "op_sel:[0,1]"};
    for (int nb = -1; nb < 2; ++nb)
        for (int sel = 0; sel < 3; ++sel) {
            hipMemset(w, 0, 4); hipDeviceSynchronize();
            if (nb == 0) hipLaunchKernelGGL(neighbour<0>, dim3(2048), dim3(256), 0, s1, sink, 40000);
            if (nb == 1) hipLaunchKernelGGL(neighbour<1>, dim3(2048), dim3(256), 0, s1, sink, 40000);
            if (sel == 0) hipLaunchKernelGGL(victim<0>, dim3(2048), dim3(256), 0, s0, p, c, w, 2000);
            if (sel == 1) hipLaunchKernelGGL(victim<1>, dim3(2048), dim3(256), 0, s0, p, c, w, 2000);
            if (sel == 2) hipLaunchKernelGGL(victim<2>, dim3(2048), dim3(256), 0, s0, p, c, w, 2000);
            hipDeviceSynchronize();
            unsigned h; hipMemcpy(&h, w, 4, hipMemcpyDeviceToHost);
            printf("neighbour %-24s v_pk_add_f32 %-16s wrong sums: %u of %llu\n", nb < 0 ? "none" : nb ? "v_mfma_f32_32x32x16_bf16" : "v_mfma_f32_32x32x2_f32",
                   names[sel], h, 2ull * 2048 * 256 * 2000);
        }
    return 0;
}
