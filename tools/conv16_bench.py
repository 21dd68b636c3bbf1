"""Micro-benchmark of one fp16 convolution through the C ABI.  usage: python tools/conv16_bench.py B H W cin cout k stride [res]"""
import ctypes, sys, torch
sys.path.insert(0, '/root/repo')
from orienmask_amd import lib as omlib
from orienmask_amd.pack import conv_weights_f16
B, H, W, cin, cout, k, stride = [int(v) for v in sys.argv[1:8]]
use_res = len(sys.argv) > 8 and sys.argv[8] == "res"
dev = torch.device("cuda:0"); L = omlib.load()
import os
x = torch.randn(B, H, W, cin, device=dev).half()
if os.environ.get('ZERO_X'): x.zero_()
if os.environ.get('SMALL_X'): x.mul_(float(os.environ['SMALL_X']))
w = conv_weights_f16(torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5, (cout + 31) // 32 * 32).to(dev)
cp = w.shape[0]
sc = torch.ones(cp, device=dev); sh = torch.zeros(cp, device=dev)
Ho, Wo = H // stride, W // stride
out = torch.empty(B, Ho, Wo, cout, device=dev, dtype=torch.float16)
res = torch.randn(B, Ho, Wo, cout, device=dev).half() if use_res else None
p = lambda t: ctypes.c_void_p(t.data_ptr())
def run():
    omlib.check(L.om_conv2d_f16(p(x), B, H, W, cin, cin, p(w), p(sc), p(sh), cout, k, stride, 1, p(res) if use_res else None,
                                cout if use_res else 0, p(out), cout, 0, omlib.current_stream_ptr(dev)), "conv")
for _ in range(5): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
fl = 2.0 * B * Ho * Wo * cout * cin * k * k
print("B=%d %dx%d %d->%d k%d s%d%s: %.4f ms  %.1f TF" % (B, H, W, cin, cout, k, stride, " +res" if use_res else "", ms, fl / ms / 1e9))
