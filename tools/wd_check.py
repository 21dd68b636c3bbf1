"""Bit-identity of the dual-role fused 3x3 kernel against the twelve-wave kernel for any library build (the test
tests/test_hip_parity.py::test_wino14_dual_equals_twelve_wave, per build):   gpurun -- 'python tools/wd_check.py ab/x.so ab/y.so'"""
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CHILD = r'''
import ctypes, os, sys
sys.path.insert(0, os.path.join(%r, ".."))
import torch
from orienmask_amd import lib as omlib
from orienmask_amd.pack import winograd14_weights_split
CASES = [(2, 34, 34, 32, 192, 1, True), (4, 17, 17, 64, 64, 1, False), (1, 40, 136, 32, 64, 0, False), (3, 68, 68, 128, 256, 1, True),
         (2, 17, 17, 512, 128, 1, True), (1, 7, 9, 32, 70, 1, False), (9, 136, 136, 64, 128, 1, False)]
omlib.LIB_PATH = os.path.abspath(sys.argv[1])
L = omlib.load()
dev = torch.device("cuda:0")
p = lambda t: ctypes.c_void_p(t.data_ptr())
res = []
for case in CASES:
    B, H, W, cin, cout, leaky, use_res = case
    g = torch.Generator().manual_seed(sum(case) + 5)
    x = torch.randn(B, H, W, cin, generator=g).to(dev)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    cpad = (cout + 63) // 64 * 64
    us, e = winograd14_weights_split(w, cpad)
    sps = torch.pow(torch.tensor(2.0), -e.float()).to(dev)
    hd = torch.zeros(cpad, device=dev)
    ud = us.to(dev)
    rd = torch.randn(B, H, W, cout, generator=g).to(dev) if use_res else None
    ostride = cout + (4 - cout %% 4) %% 4
    outs = []
    for variant in (0, 1, 1):
        L.om_set_wino14_variant(variant)
        out = torch.full((B, H, W, ostride), float("nan"), device=dev)
        omlib.check(L.om_conv2d_wino14_split(p(x), B, H, W, cin, cin, p(ud), p(sps), p(hd), cout, leaky, p(rd) if use_res else None,
                                             cout if use_res else 0, p(out), ostride, None, omlib.current_stream_ptr(dev)), "w14")
        torch.cuda.synchronize()
        outs.append(out.cpu()[..., :cout])
    d = [(o - outs[0]).abs() for o in outs[1:]]
    bad = [int((~torch.eq(o, outs[0])).sum()) for o in outs[1:]]
    res.append("%%s" %% ("ok" if bad == [0, 0] else "BAD%%s max %%.3g" %% (bad, max(float(torch.nan_to_num(x_, nan=9e9).max()) for x_ in d))))
print("%%-40s %%s" %% (sys.argv[1], " | ".join(res)), flush=True)
''' % HERE
for lib in sys.argv[1:]:
    subprocess.run([sys.executable, "-c", CHILD, lib], timeout=600)
