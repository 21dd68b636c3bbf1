import ctypes, sys, os
sys.path.insert(0, "/root/repo")
import torch
from orienmask_amd import lib as omlib
from orienmask_amd.pack import winograd14_weights_split
L = omlib.load()
dev = torch.device("cuda:0")
_p = lambda t: ctypes.c_void_p(t.data_ptr())
for case in [(2, 16, 16, 32, 64, 1, True), (1, 17, 17, 64, 128, 1, False)]:
    B, H, W, cin, cout, leaky, use_res = case
    g = torch.Generator().manual_seed(sum(case) + 23)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    scale = torch.ones(cout); shift = torch.zeros(cout)
    res = torch.randn(B, cout, H, W, generator=g) if use_res else None
    want = torch.nn.functional.conv2d(x.double(), w.double(), None, 1, 1)
    if leaky: want = torch.where(want > 0, want, want * 0.1)
    if use_res: want = want + res.double()
    cpad = (cout + 63) // 64 * 64
    us, e = winograd14_weights_split(w, cpad)
    sp = torch.zeros(cpad); sp[:cout] = scale
    sps = (sp.double() * torch.pow(torch.tensor(2.0, dtype=torch.float64), -e.double())).float().to(dev)
    hd = torch.zeros(cpad).to(dev)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    rd = res.permute(0, 2, 3, 1).contiguous().to(dev) if use_res else None
    out = torch.full((B, H, W, cout), float("nan"), device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    ud = us.to(dev)
    rc = L.om_conv2d_wino14_split(_p(xd), B, H, W, cin, cin, _p(ud), _p(sps), _p(hd), cout, leaky, _p(rd) if use_res else None,
                                  cout if use_res else 0, _p(out), cout, _p(status), omlib.current_stream_ptr(dev))
    torch.cuda.synchronize()
    got = out.cpu().permute(0, 3, 1, 2).double()
    err = (got - want).abs()
    print(case, "rc", rc, "status", int(status.item()), "nan", int(torch.isnan(got).sum()), "max err", float(err[~torch.isnan(err)].max()))
    bad = (err > 1e-3) | torch.isnan(err)
    print(" bad frac", float(bad.float().mean()))
    if bad.any():
        print(" bad by batch", bad.float().mean(dim=(1, 2, 3)).tolist())
        print(" bad by channel/8", bad.float().mean(dim=(0, 2, 3)).view(-1, 8).mean(1).tolist())
        print(" bad by row", [round(v, 2) for v in bad.float().mean(dim=(0, 1, 3)).tolist()])
        print(" bad by col", [round(v, 2) for v in bad.float().mean(dim=(0, 1, 2)).tolist()])
