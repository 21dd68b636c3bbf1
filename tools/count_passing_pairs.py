import sys, torch
sys.path.insert(0, '/root/repo')
import bench
from orienmask_amd import synth
from orienmask_amd.model import OrienMaskYOLOFPNPlus
dev = torch.device('cuda:0')
net = OrienMaskYOLOFPNPlus(3, 80).eval().set_precision('f32_split')
net.load_state_dict(synth.synth_state_dict(bench.WEIGHT_SEED, obj_bias=bench.OBJ_BIAS, head_gain=bench.HEAD_GAIN), strict=True)
net = net.to(dev)
pc = bench.post_config(544, 544)
print({k: pc[k] for k in pc if k in ('conf_thresh', 'nms_pre', 'nms_post', 'nms_thresh')})
x = synth.synth_image_batch(1000, 4, 544, 544).to(dev)
with torch.no_grad():
    out = net(x)
    for i, (bb, oo) in enumerate(out):
        B = bb.shape[0]
        t = bb.reshape(B, -1, 3, 85) if bb.shape[-1] == 255 else bb.permute(0, 2, 3, 1).reshape(B, -1, 3, 85)
        conf = torch.sigmoid(t[..., 4:5]) * torch.sigmoid(t[..., 5:])
        print('scale', i, bb.shape, 'passing per image', (conf > pc['conf_thresh']).reshape(B, -1).sum(1).tolist())
