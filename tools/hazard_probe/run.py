"""Does a dense run of floating-point compares survive fp16-MFMA waves of another stream on the same SIMD?

    gpurun -- 'python tools/hazard_probe/run.py'

Builds probe.hip (hipcc, gfx950), then launches the compare form and the arithmetic form of the mask predicate
(a) alone and (b) while 60 launches of one fp16 3x3 convolution of this library (om_conv2d_f16, 1 x 136 x 136, 128 -> 256)
run on a second HIP stream, and counts the output words that differ from the launch that ran alone.  Also checks that the two
forms agree bit for bit on random and special operands (NaN, +-inf, denormals, +-0, negative and NaN thresholds).
Result on MI355X (ROCm 7.2), profiles/r02_experiments.md section 6: compare form ~12 000 wrong words of 52 M per launch in (b),
0 in (a); arithmetic form 0 in both.
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


with torch.cuda.stream(s0):
    ref_c, ref_a = probe(0), probe(1)
torch.cuda.synchronize()
print("alone: compare form vs arithmetic form, differing words: %d of %d (fraction of set bytes %.3f)"
      % (int((ref_c != ref_a).sum()), ref_c.numel(), float((ref_c & 1).float().mean())))
for it in range(6):
    for beside in (False, True):
        torch.cuda.synchronize()
        if beside:
            for _ in range(60):
                conv_on(s1)
        with torch.cuda.stream(s0):
            c, a = probe(0), probe(1)
        torch.cuda.synchronize()
        print("run %d %-28s compare form: %6d wrong words   arithmetic form: %6d wrong words"
              % (it, "beside the fp16 convolution" if beside else "alone", int((c != ref_c).sum()), int((a != ref_a).sum())), flush=True)
