"""One layer of the latency mode's implicit GEMM, launched back to back (for rocprofv3 --pmc passes: tools/pmc_splitk_layer.sh):
    python tools/splitk_layer.py [--lib ab/NAME.so] [--hw 17 --cin 512 --cout 1024 --k 3 --parts 8 --launches 20]"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--hw", type=int, default=17)
ap.add_argument("--cin", type=int, default=512)
ap.add_argument("--cout", type=int, default=1024)
ap.add_argument("--k", type=int, default=3)
ap.add_argument("--parts", type=int, default=8)
ap.add_argument("--launches", type=int, default=20)
args = ap.parse_args()
from orienmask_amd import lib as omlib          # noqa: E402
if args.lib:
    omlib.LIB_PATH = os.path.abspath(args.lib)
from orienmask_amd.pack import conv_weights_split     # noqa: E402
dev = torch.device("cuda", 0)
L = omlib.load()
p = lambda t: ctypes.c_void_p(t.data_ptr())
x = torch.randn(1, args.hw, args.hw, args.cin, device=dev)
w = torch.randn(args.cout, args.cin, args.k, args.k) / (args.cin * args.k * args.k) ** 0.5
ws, e = conv_weights_split(w, args.cout)
wd = ws.to(dev)
sp = torch.pow(torch.tensor(2.0), -e.float()).to(dev)
hp = torch.zeros(args.cout, device=dev)
out = torch.empty(1, args.hw, args.hw, args.cout, device=dev)
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)      # evicts the Infinity Cache between launches
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ms = []
for i in range(args.launches):
    flush.fill_(i & 1)
    a.record()
    omlib.check(L.om_conv2d_split_k(p(x), 1, args.hw, args.hw, args.cin, args.cin, p(wd), p(sp), p(hp), args.cout, args.k, 1, 1, None, 0,
                                    p(out), args.cout, 0, 1, 64, 64, args.parts, None, omlib.current_stream_ptr(dev)), "conv")
    b.record()
    torch.cuda.synchronize()
    ms.append(a.elapsed_time(b))
print("%s: %d x %d, %d -> %d, k %d, <= %d parts, cold caches: median %.4f ms" % (args.lib or "in-tree", args.hw, args.hw, args.cin, args.cout, args.k,
                                                                               args.parts, sorted(ms)[len(ms) // 2]))
