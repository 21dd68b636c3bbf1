"""Does a dense run of floating-point compares survive matrix-instruction waves of another stream on the same SIMD?

    gpurun -- 'python tools/hazard_probe/run.py'

Builds probe.hip (hipcc, gfx950), then launches four forms of the mask predicate (compare: v_cmp_e64 -> SGPR pair -> s_and_b64 ->
v_cndmask; arithmetic: bit patterns in VGPRs; vcc: VOPC -> VCC -> v_cndmask; branches: v_cmp -> s_and_saveexec) alone and while a
second HIP stream runs, in turn: kernels that only issue one matrix instruction on registers, one fp16 / fp32 3x3 convolution of
this library, rocBLAS GEMMs, an elementwise kernel; and counts the output words that differ from the launch that ran alone.  Also
checks that the forms agree bit for bit on random and special operands (NaN, +-inf, denormals, +-0, negative and NaN thresholds).
Result on MI355X (ROCm 7.2): profiles/r02_experiments.md section 6; root cause (round 5: the compiler's PACKED subtractions with a
register half-selection, forms 8-11; not the compares): profiles/r05_experiments.md section 2, pk_opsel_repro.hip.
"""
import ctypes
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from orienmask_amd import lib as omlib                       # noqa: E402
from orienmask_amd.pack import conv_weights_f16               # noqa: E402

so = os.path.join(HERE, "probe.so")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(HERE, "probe.hip")):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC",
                           "-o", so, os.path.join(HERE, "probe.hip")])
P = ctypes.CDLL(so)
L = omlib.load()
dev = torch.device("cuda:0")
_p = lambda t: ctypes.c_void_p(t.data_ptr())
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()

g = torch.Generator().manual_seed(1)
seed = torch.rand(65536, generator=g).to(dev)
special = torch.tensor([float("nan"), float("inf"), -float("inf"), 1e-40, -1e-40, 0.0, -0.0, 1e38], device=dev)
seed[::97][:special.numel() * 50] = special.repeat(50)
dets = (torch.rand(100, 4, generator=g) * torch.tensor([8., 8., 3., 3.])).to(dev)
dets[5, 2] = float("nan"); dets[6, 3] = -1.0; dets[7, 2] = float("inf"); dets[8, 3] = 0.0; dets[9, 2] = -0.0
dets[10, 3] = 1e-41; dets[11, 0] = float("nan"); dets[12, 1] = float("inf"); dets[13, 2] = 1e-40
dets[14] = torch.tensor([1e-40, -1e-40, 1e-39, 2e-40], device=dev)
BLOCKS = 2048


def probe(arithmetic):
    out = torch.empty((100, BLOCKS * 256, 4), dtype=torch.int32, device=dev)
    rc = P.probe_launch(arithmetic, _p(seed), _p(dets), 100, _p(out), BLOCKS, ctypes.c_void_p(s0.cuda_stream))
    assert rc == 0
    return out


# the disturbing kernel: one fp16 3x3 convolution of the library
B, H, W, cin, cout = 1, 136, 136, 128, 256
xd = torch.randn(B, H, W, cin, generator=g).half().to(dev)
wd = conv_weights_f16(torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5, cout).contiguous().to(dev)
sp, hp = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
yd = torch.empty((B, H, W, cout), device=dev, dtype=torch.float16)


def conv_on(stream):
    omlib.check(L.om_conv2d_f16(_p(xd), B, H, W, cin, cin, _p(wd), _p(sp), _p(hp), cout, 3, 1, 1, None, 0, _p(yd), cout, 0,
                                ctypes.c_void_p(stream.cuda_stream)), "om_conv2d_f16")


# other neighbours, to see what about the fp16 convolution matters
xf = torch.randn(B, H, W, cin, generator=g).to(dev)
wf = (torch.randn(cout, 9 * cin, generator=g) / (cin * 9) ** 0.5).to(dev)
yf = torch.empty((B, H, W, cout), device=dev)
h16 = torch.randn(4096, 4096, generator=g).half().to(dev)
f32m = torch.randn(2048, 2048, generator=g).to(dev)
big = torch.randn(1 << 26, generator=g).to(dev)


def conv32_on(stream):
    omlib.check(L.om_conv2d(_p(xf), B, H, W, cin, cin, _p(wf), _p(sp), _p(hp), cout, 3, 1, 1, None, 0, _p(yf), cout,
                            ctypes.c_void_p(stream.cuda_stream)), "om_conv2d")


sink = torch.zeros(4, device=dev)


def mfma_only(kind):
    return lambda s: P.neighbour_launch(kind, _p(sink), 2048, 20000, ctypes.c_void_p(s.cuda_stream))


NEIGHBOURS = [
    ("alone", None, 0),
    ("only v_mfma_f32_32x32x16_f16", mfma_only(0), 2),
    ("only v_mfma_f32_32x32x2_f32", mfma_only(1), 2),
    ("only v_mfma_f32_32x32x16_bf16", mfma_only(2), 2),
    ("only v_mfma_f32_16x16x32_f16", mfma_only(3), 2),
    ("this library's fp16 3x3 conv", conv_on, 60),
    ("this library's fp32 3x3 conv", conv32_on, 12),
    ("rocBLAS fp16 GEMM 4096^3", lambda s: torch.mm(h16, h16), 12),
    ("rocBLAS fp32 GEMM 2048^3", lambda s: torch.mm(f32m, f32m), 12),
    ("elementwise sin over 64 M floats", lambda s: torch.sin(big), 12),
]

FORMS = ("compare", "arithmetic", "vcc", "branches", "asm+0nop", "asm+5nop", "asm+16nop", "asm vcc interleaved", "compare, scalar subs", "asm packed subs", "op_sel packed subs + VGPR predicate", "op_sel packed subs + compares")
with torch.cuda.stream(s0):
    refs = [probe(f) for f in range(len(FORMS))]
torch.cuda.synchronize()
print("alone: forms agree with the compare form: %s (%d words, fraction of set bytes %.3f)"
      % ([int((r != refs[0]).sum()) for r in refs], refs[0].numel(), float((refs[0] & 1).float().mean())))
for it in range(2):
    for name, fn, reps in NEIGHBOURS:
        wrong = []
        for f in range(len(FORMS)):
            torch.cuda.synchronize()
            if fn is not None:
                with torch.cuda.stream(s1):
                    for _ in range(reps):
                        fn(s1)
            with torch.cuda.stream(s0):
                got = probe(f)
            torch.cuda.synchronize()
            wrong.append(int((got != refs[f]).sum()))
            del got
        print("run %d beside %-36s wrong words: %s" % (it, name, "  ".join("%s %7d" % (FORMS[f], wrong[f]) for f in range(len(FORMS)))), flush=True)
