"""orien_head.5-like layer: 136^2, 128 -> 18 (pad 32), 1x1: NCHW fp32 out (out_mode 2) vs NHWC (out_mode 0), tile shapes"""
import ctypes, os, sys, torch
sys.path.insert(0, '/root/repo')
from orienmask_amd import lib as omlib
from orienmask_amd.pack import conv_weights_split
dev = torch.device("cuda:0"); L = omlib.load()
p = lambda t: ctypes.c_void_p(t.data_ptr())
B, hw, cin, cout = 32, 136, 128, 18
x = torch.randn(B, hw, hw, cin, device=dev)
w = torch.randn(cout, cin, 1, 1) / cin ** 0.5
ws, e = conv_weights_split(w, 32)
wd = ws.to(dev); sp = torch.pow(torch.tensor(2.0), -e.float()).to(dev); hp = torch.zeros(32, device=dev)
out = torch.empty(B * hw * hw * 32, device=dev)
st = omlib.current_stream_ptr(dev)
for mode, ops, bm, bn in ((2, cout, 0, 0), (0, 32, 0, 0), (2, cout, 128, 32), (2, cout, 64, 64) if False else (0, 32, 128, 32)):
    def run():
        omlib.check(L.om_conv2d_split(p(x), B, hw, hw, cin, cin, p(wd), p(sp), p(hp), cout if mode == 2 else 32, 1, 1, 0, None, 0, p(out), ops, mode, 1, bm, bn, None, st), "conv")
    for _ in range(3): run()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(20): run()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    print("out_mode %d tile %dx%d: %.4f ms  %.0f GB/s in" % (mode, bm, bn, ms, x.numel() * 4 / ms / 1e6))
