"""Phase times of one steady-state tile of conv_stem2_split_kernel (a measurement build: tools/build_variant.sh s2trace "-DOM_S2_TRACE" conv_stem2):
   gpurun -- 'OM_LIB=ab/s2trace.so python tools/stem2_trace.py'   -> patch staged | conv1 | conv2.0 | epilogue | third layer, in shader cycles"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from orienmask_amd import lib as omlib
from orienmask_amd.pack import conv_weights_split
if os.environ.get("OM_LIB"):
    omlib.LIB_PATH = os.path.abspath(os.environ["OM_LIB"])
L = omlib.load()
dev = torch.device("cuda:0")
B, H, W = 32, 544, 544
p = lambda t: ctypes.c_void_p(t.data_ptr())
x = torch.rand(B, 3, H, W, device=dev)
w1 = (torch.randn(32, 27) * 0.3).to(dev)
ws2, e2 = conv_weights_split(torch.randn(64, 32, 3, 3) / 17.0, 64)
ws3, e3 = conv_weights_split(torch.randn(32, 64, 1, 1) / 8.0, 32)
ws2, ws3 = ws2.to(dev), ws3.to(dev)
one32, zero32, one64, zero64 = torch.ones(32, device=dev), torch.zeros(32, device=dev), torch.ones(64, device=dev), torch.zeros(64, device=dev)
out = torch.empty(B, H // 2, W // 2, 64, device=dev)
out3 = torch.empty(B, H // 2, W // 2, 32, device=dev)
st = omlib.current_stream_ptr(dev)
for third in (False, True):
    tr = torch.zeros(16, dtype=torch.int64, device=dev)
    for _ in range(3):
        if third:
            omlib.check(L.om_conv2d_stem3_split(p(x), B, H, W, p(w1), p(one32), p(zero32), p(ws2), p(one64), p(zero64), 64, 1, p(out), 64,
                                                p(ws3), p(one32), p(zero32), 32, 1, p(out3), 32, p(tr), st), "stem3")
        else:
            omlib.check(L.om_conv2d_stem2_split(p(x), B, H, W, p(w1), p(one32), p(zero32), p(ws2), p(one64), p(zero64), 64, 1, p(out), 64, p(tr), st), "stem2")
    torch.cuda.synchronize()
    t = tr.cpu().tolist()
    print("third layer %-5s  patch %6d  conv1 %6d  conv2.0 %6d  epilogue %6d  third %6d   tile %6d cycles" % (
        third, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[5] - t[3], t[4] - t[5], t[4] - t[0]))
