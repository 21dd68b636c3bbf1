import ctypes, os, sys
sys.path.insert(0, "/root/repo")
import torch
from orienmask_amd import lib as omlib
from orienmask_amd.pack import conv_weights_split
omlib.LIB_PATH = os.path.abspath(os.environ["OM_LIB"])
L = omlib.load(); dev = torch.device("cuda:0")
p = lambda t: ctypes.c_void_p(t.data_ptr())
B, H, W = 32, 544, 544
x = torch.rand(B, 3, H, W, device=dev)
w1 = (torch.randn(32, 27) * 0.3).to(dev); sc1 = torch.ones(32, device=dev); sh1 = torch.zeros(32, device=dev)
ws, e = conv_weights_split(torch.randn(64, 32, 3, 3) / 17, 64)
wsd = ws.to(dev); sp2 = torch.pow(torch.tensor(2.0), -e.float()).to(dev); sh2 = torch.zeros(64, device=dev)
out = torch.empty(B, H // 2, W // 2, 64, device=dev)
tr = torch.zeros(8, dtype=torch.int64, device=dev)
for _ in range(2):
    omlib.check(L.om_conv2d_stem2_split(p(x), B, H, W, p(w1), p(sc1), p(sh1), p(wsd), p(sp2), p(sh2), 64, 1, p(out), 64, p(tr), omlib.current_stream_ptr(dev)), "s2")
torch.cuda.synchronize()
t = tr.cpu().tolist()
print("patch->LDS + barrier %d | conv1 phase %d | patch request + matrix phase %d | barrier + epilogue %d | tile %d cycles" % (t[1]-t[0], t[2]-t[1], t[3]-t[2], t[4]-t[3], t[4]-t[0]))
