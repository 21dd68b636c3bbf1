"""Time the dual-role fused 3x3 kernel (conv_wino14d.hip) of several library builds on a few layer shapes at bs=32, one process per
build:   gpurun -- 'python tools/wd_ablate.py orienmask_amd/lib/liborienmask_hip.so ab/wd1.so ...'   (OM_SHAPES=5,2,4 by default)"""
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
SHAPES = [(272, 32, 64), (136, 64, 128), (68, 128, 256), (34, 256, 512), (17, 512, 1024), (136, 128, 256)]
omlib.LIB_PATH = os.path.abspath(sys.argv[1])
L = omlib.load()
dev = torch.device("cuda:0")
B = 32
p = lambda t: ctypes.c_void_p(t.data_ptr())
out_line = []
for i in [int(k) for k in os.environ.get("OM_SHAPES", "5,2,4").split(",")]:
    hw, cin, cout = SHAPES[i]
    x = torch.randn(B, hw, hw, cin, device=dev)
    w = torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5
    out = torch.empty(B, hw, hw, cout, device=dev)
    hd = torch.zeros(cout, device=dev)
    u14, e14 = winograd14_weights_split(w, cout)
    s14 = torch.pow(torch.tensor(2.0), -e14.float()).to(dev)
    u14 = u14.to(dev)
    st = omlib.current_stream_ptr(dev)
    ts = []
    for variant in (1, 0):
        L.om_set_wino14_variant(variant)
        fn = lambda: omlib.check(L.om_conv2d_wino14_split(p(x), B, hw, hw, cin, cin, p(u14), p(s14), p(hd), cout, 1, None, 0, p(out), cout, None, st), "w14")
        for _ in range(3): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): fn()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 10)
    out_line.append("%%dx%%d %%d->%%d: dual %%.3f ms  twelve-wave %%.3f" %% (hw, hw, cin, cout, ts[0], ts[1]))
print("%%-42s %%s" %% (sys.argv[1], " | ".join(out_line)), flush=True)
''' % HERE

for lib in sys.argv[1:]:
    subprocess.run([sys.executable, "-c", CHILD, lib], timeout=300)
