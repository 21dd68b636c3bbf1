"""Soak of the two-kernel wide form of the fused 3x3 kernel (conv_wino14.hip, round 6) against the fused kernel on the layer shapes
om_forward runs it on, fresh random inputs every iteration, the scratch poisoned in between:   gpurun -- 'python tools/soak_wide.py 300'
Prints how many iterations differed from the fused kernel's output (must be 0)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from orienmask_amd import lib as omlib
from orienmask_amd.pack import winograd14_weights_split

L = omlib.load()
dev = torch.device("cuda:0")
p = lambda t: ctypes.c_void_p(t.data_ptr())
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B, hw, cin, cout = 32, 17, 512, 1024
g = torch.Generator().manual_seed(1)
w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
u, e = winograd14_weights_split(w, cout)
sc = torch.pow(torch.tensor(2.0), -e.float()[:cout]).to(dev)
u = u.to(dev)
hd = (torch.randn(cout, generator=g) * 0.1).to(dev)
nb = L.om_conv2d_wino14_wide_scratch_bytes(B, hw, hw, cin)
scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
st = omlib.current_stream_ptr(dev)
bad = 0
t0 = time.time()
for it in range(N):
    x = torch.randn(B, hw, hw, cin, device=dev) * (1.0 + it % 7)
    res = torch.randn(B, hw, hw, cout, device=dev) if it % 2 else None
    a = torch.empty(B, hw, hw, cout, device=dev); b = torch.empty_like(a)
    omlib.check(L.om_conv2d_wino14_split(p(x), B, hw, hw, cin, cin, p(u), p(sc), p(hd), cout, 1, p(res) if res is not None else None,
                                         cout if res is not None else 0, p(a), cout, None, st), "fused")
    scratch.fill_(0xFF if it % 3 else 0x7C)
    omlib.check(L.om_conv2d_wino14_wide(p(x), B, hw, hw, cin, cin, p(u), p(sc), p(hd), cout, 1, p(res) if res is not None else None,
                                        cout if res is not None else 0, p(b), cout, p(scratch), nb, None, st), "wide")
    torch.cuda.synchronize()
    if not torch.equal(a, b):
        bad += 1
print("wide form against the fused kernel, %d iterations of 32 x 17^2 512 -> 1024 (every second one with a residual): %d differ; %.1f s" % (N, bad, time.time() - t0))
