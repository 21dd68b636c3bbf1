import sys, os, ctypes
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from orienmask_amd import lib as omlib, synth
from conftest import post_cfg
from oracle import orienmask_ref as R
L = omlib.load(); dev = torch.device("cuda:0")
f32 = np.float32
def fma(a, b, c): return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)
def sleef_parts(d):
    q = np.rint(d * f32(1.4426950408889634)).astype(np.int32); qf = q.astype(np.float32)
    s = fma(qf, np.full_like(d, -f32(0.693145751953125)), d); s = fma(qf, np.full_like(d, -f32(1.428606765330187045e-06)), s)
    u = np.full_like(d, f32(0.000198527617612853646278381))
    for c in (0.00139304355252534151077271, 0.00833336077630519866943359, 0.0416664853692054748535156, 0.166666671633720397949219, 0.5):
        u = fma(u, s, np.full_like(d, f32(c)))
    u = (f32(1.0) + fma((s * s).astype(np.float32), u, s)).astype(np.float32)
    q1 = q >> 1
    p1 = ((q1 + 127).astype(np.int32) << 23).view(np.float32); p2 = (((q - q1) + 127).astype(np.int32) << 23).view(np.float32)
    return ((u * p1).astype(np.float32) * p2).astype(np.float32), q, s, u
rng = np.random.default_rng(1)
d = rng.uniform(-30, 30, 2_000_000).astype(np.float32)
dd = torch.from_numpy(d).to(dev); out = torch.empty_like(dd)
omlib.check(L.om_ref_math(ctypes.c_void_p(dd.data_ptr()), dd.numel(), 1, 80, ctypes.c_void_p(out.data_ptr()), omlib.current_stream_ptr(dev)), "x")
got = out.cpu().numpy(); want, q, s, u = sleef_parts(d)
tw = torch.from_numpy(d).view(-1, 64).exp()   # contiguous: MKL path, just for info
bad = got.view(np.int32) != want.view(np.int32)
print("expf_sleef device vs numpy emulation: mismatches", bad.sum())
i = np.nonzero(bad)[0][:8]
print("d", d[i], "q", q[i], "s", s[i], "u", u[i], "got", got[i], "want", want[i])
# torch vectorised exp on a [rows,64] slice of [rows,96]
v = torch.from_numpy(d[:(d.size // 96) * 96]).view(-1, 96)[:, :64]
# sigmoid route: exp(-x)
sg = v.sigmoid().numpy()
em = (f32(1) / (f32(1) + sleef_parts((-v.numpy()).reshape(-1))[0])).reshape(sg.shape).astype(np.float32)
print("torch sigmoid vs emulation mismatches", (sg.view(np.int32) != em.view(np.int32)).sum())

# ---- composed path on the failing golden
g = np.load("tests/golden/fwd_f160x128_b1.npz")
sd = synth.synth_state_dict(int(g["wseed"]), obj_bias=float(g["obj_bias"]), head_gain=float(g["head_gain"]))
x = synth.synth_image_batch(int(g["xseed"]), 1, 160, 128)
from orienmask_amd.model import OrienMaskYOLOFPNPlus
from orienmask_amd.eval import OrienMaskYOLOPostProcess
net = OrienMaskYOLOFPNPlus(3, 80).eval(); net.load_state_dict(sd, strict=True); net = net.to(dev)
with torch.no_grad(): out = net(x.to(dev))
post = OrienMaskYOLOPostProcess(device=dev, **post_cfg((160, 128)))
res = post(out)
print("HIP composed: K", res[0]["bbox"].shape[0], "cls", res[0]["cls"][:12].tolist(), "keep", post.last_keep[0][:12].tolist())
print("scores", res[0]["bbox"][:8, 4].tolist())
pc = post_cfg((160, 128))
o = R.PostProcessOracle(pc["grid_size"], pc["image_size"], pc["anchors"], pc["anchor_mask"], 80, conf_thresh=0.005)
hh = [(b.cpu(), oo.cpu()) for b, oo in out]
w = o(hh)[0]
print("oracle on HIP heads: n_cand", w["n_candidates"], "K", w["bbox"].shape[0], "cls", w["cls"][:12].tolist(), "keep", w["keep"][:12].tolist())
res2 = post(tuple((b.contiguous(), oo) for b, oo in out))
print("HIP post on contiguous copy: cls", res2[0]["cls"][:12].tolist())
