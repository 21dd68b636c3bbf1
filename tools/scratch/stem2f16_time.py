import ctypes, sys, torch
lib = ctypes.CDLL(sys.argv[1])
dev = torch.device("cuda:0")
B, H, W = 32, 544, 544
p = lambda t: ctypes.c_void_p(t.data_ptr())
x = torch.rand(B, 3, H, W, device=dev)
w1 = (torch.randn(32, 27) * 0.3).to(dev)
w2 = (torch.randn(64, 288) / 17.0).half().to(dev)
one32, zero32, one64, zero64 = torch.ones(32, device=dev), torch.zeros(32, device=dev), torch.ones(64, device=dev), torch.zeros(64, device=dev)
out = torch.empty(B, H // 2, W // 2, 64, device=dev, dtype=torch.float16)
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
f = lib.om_conv2d_stem2_f16
f.restype = ctypes.c_int
def run():
    rc = f(p(x), B, H, W, p(w1), p(one32), p(zero32), p(w2), p(one64), p(zero64), 64, 1, p(out), 64, st)
    assert rc == 0
for _ in range(3): run()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); a.record()
for _ in range(20): run()
b.record(); torch.cuda.synchronize()
print(sys.argv[1], "%.4f ms" % (a.elapsed_time(b) / 20), float(out.float().abs().mean()))
