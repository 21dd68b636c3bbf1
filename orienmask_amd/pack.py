"""Reference state_dict (524 keys) -> packed weight blob of the HIP graph.

Blob layout comes from the library (``om_model_layer_info``): per convolution
  weights [cout_pad][kh*kw][cin]  (OHWI, rows >= cout zero)   scale [cout_pad]   shift [cout_pad]
with eval-mode BatchNorm folded into the epilogue constants,
  scale = gamma / sqrt(running_var + eps),  shift = beta - running_mean * scale
(/root/reference/model/base.py:113-128; eps = 1e-5 is nn.BatchNorm2d's default) and, for the four
bias-only head convolutions (/root/reference/model/orienmask_yolo_fpnplus.py:60,71), scale = 1 and
shift = bias.  The fold is computed in float64 and rounded once.

Checkpoint formats accepted are the reference's: a raw state_dict or {'state_dict': ...}
(/root/reference/infer.py:81-83, /root/reference/trainer/builder.py:45-52).
"""
import ctypes

import torch

from . import lib as _lib
from .arch import BN_EPS, model_convs


def graph_layers(handle):
    """Layer table of a created om_model as a list of dicts (execution order)."""
    L = _lib.load()
    out = []
    for i in range(L.om_model_num_layers(handle)):
        info = _lib.LayerInfo()
        _lib.check(L.om_model_layer_info(handle, i, ctypes.byref(info)), "om_model_layer_info")
        out.append(dict(name=info.name.decode(), cin=info.cin, cout=info.cout, cout_pad=info.cout_pad,
                        ksize=info.ksize, stride=info.stride, has_bn=bool(info.has_bn), leaky=bool(info.leaky),
                        w_off=info.w_off, scale_off=info.scale_off, shift_off=info.shift_off,
                        wino_off=info.wino_off, wino_planes=info.wino_planes, wino_alt_off=info.wino_alt_off,
                        w16_off=info.w16_off, wsplit_off=info.wsplit_off, wsplit_scale_off=info.wsplit_scale_off,
                        wsplit_direct_off=info.wsplit_direct_off, wsplit_direct_scale_off=info.wsplit_direct_scale_off))
    return out


def check_graph_matches_arch(layers, num_anchors, num_classes, model="OrienMaskYOLOFPNPlus"):
    """The C++ graph and the Python layer table are written independently; they must agree."""
    want = {s.name: s for s in model_convs(model, num_anchors, num_classes)}
    got = {l["name"]: l for l in layers}
    if set(want) != set(got):
        raise _lib.OrienMaskHipError("graph/arch layer names differ: %s" % sorted(set(want) ^ set(got)))
    for name, s in want.items():
        l = got[name]
        if (s.cin, s.cout, s.ksize, s.stride, s.bn) != (l["cin"], l["cout"], l["ksize"], l["stride"], l["has_bn"]):
            raise _lib.OrienMaskHipError("graph/arch disagree on %s: %s vs %s" % (name, s, l))


def unwrap_checkpoint(obj):
    if isinstance(obj, dict) and "state_dict" in obj and isinstance(obj["state_dict"], dict):
        return obj["state_dict"]
    return obj


# Winograd F(2x2,3x3) kernel transform matrix (Lavin & Gray 2015): U = G g G^T
_WINO_G = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]], dtype=torch.float64)


# F(4,3) kernel transform with points {0, +-1, +-2}: the column transform of F(2x4,3x3)
_WINO_G6 = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
                         [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float64)

# Every function below computes ON THE DEVICE ITS WEIGHTS ARE ON (a model moved to the GPU packs there: 0.6 s instead of the
# 12 s the host takes for the three blobs) and returns tensors on that device.  The arithmetic is written so that both devices
# produce the same bits: float64 elementwise products and sums in a fixed order (no einsum / GEMM, whose summation order is the
# library's), exponents through frexp and powers of two built from their bit patterns (no log2 / pow).


def _apply3(x, g):
    """x: the three operands [3][...] (float64) -> [J][...]: y[j] = (x[0] g[j][0] + x[1] g[j][1]) + x[2] g[j][2], in that order
    (scalar times contiguous plane: the same IEEE operations on the host and on the GPU)."""
    rows = g.tolist()
    return torch.stack([(x[0] * r[0] + x[1] * r[1]) + x[2] * r[2] for r in rows])


def _pow2(e):
    """2^e as float64 for an integer-valued tensor e in [-1000, 1000], exactly (the exponent field written directly)."""
    return ((e.to(torch.int64) + 1023) << 52).view(torch.float64)


def _pow2_exponent(amax):
    """Exponents e (float64 tensor) that put each magnitude into [2^13, 2^14); 0 where it is zero."""
    amax = amax.double()
    ex = torch.frexp(amax).exponent.double()                     # amax = m 2^ex, m in [0.5, 1): floor(log2 amax) = ex - 1
    return torch.where(amax > 0, 14 - ex, torch.zeros_like(amax)).clamp(-100, 100)


def winograd_weights(w, cout_pad, planes=16):
    """[cout,cin,3,3] -> U [planes][cout_pad][cin] float32 (rows >= cout zero), computed in float64.
    planes = 16: F(2x2,3x3), U[4 i + j] = G[i] g G[j]^T;  planes = 24: F(2x4,3x3), U[6 i + j] = G[i] g G6[j]^T
    (i: transform index down the rows, j: along the columns)."""
    cout, cin = w.shape[0], w.shape[1]
    gx = _WINO_G if planes == 16 else _WINO_G6
    t = _apply3(w.detach().double().permute(3, 2, 0, 1).contiguous(), gx)          # [s][r][n][c] -> [j][r][n][c]
    u = _apply3(t.transpose(0, 1), _WINO_G)                                          # [r][j][n][c] -> [i][j][n][c]
    out = torch.zeros(planes, cout_pad, cin, dtype=torch.float32, device=w.device)
    out[:, :cout] = u.reshape(planes, cout, cin).float()
    return out


def folded_epilogue(sd, l):
    """(weight tensor, scale float64 [cout], shift float64 [cout]) of layer l: eval-mode BatchNorm folded in float64
    (module docstring), or scale 1 / shift = bias for the bias-only head convolutions."""
    name, cout = l["name"], l["cout"]
    if l["has_bn"]:
        w = sd[name + ".conv_block.0.weight"]
        p = name + ".conv_block.1."
        gamma, beta = sd[p + "weight"].double(), sd[p + "bias"].double()
        mean, var = sd[p + "running_mean"].double(), sd[p + "running_var"].double()
        scale = gamma / torch.sqrt(var + BN_EPS)
        shift = beta - mean * scale
    else:
        w = sd[name + ".weight"]
        scale = torch.ones(cout, dtype=torch.float64, device=w.device)
        shift = sd[name + ".bias"].double()
    return w, scale.to(w.device), shift.to(w.device)


def _blob_device(sd, layers):
    l = layers[0]
    return sd[l["name"] + (".conv_block.0.weight" if l["has_bn"] else ".weight")].device


def pack_state_dict(state_dict, layers, total_floats):
    """Returns a float32 tensor of total_floats elements laid out as the graph expects (on the weights' device)."""
    sd = unwrap_checkpoint(state_dict)
    blob = torch.zeros(total_floats, dtype=torch.float32, device=_blob_device(sd, layers))
    for l in layers:
        name, cin, cout, cpad, k = l["name"], l["cin"], l["cout"], l["cout_pad"], l["ksize"]
        w, scale, shift = folded_epilogue(sd, l)
        if tuple(w.shape) != (cout, cin, k, k):
            raise _lib.OrienMaskHipError("%s: weight shape %s, expected %s" % (name, tuple(w.shape), (cout, cin, k, k)))
        ohwi = w.detach().float().permute(0, 2, 3, 1).reshape(cout, k * k * cin)
        blob[l["w_off"]:l["w_off"] + cout * k * k * cin] = ohwi.reshape(-1)
        blob[l["scale_off"]:l["scale_off"] + cout] = scale.float()
        blob[l["shift_off"]:l["shift_off"] + cout] = shift.float()
        if l.get("wino_off", -1) >= 0:
            planes = l.get("wino_planes", 16) or 16
            blob[l["wino_off"]:l["wino_off"] + planes * cpad * cin] = winograd_weights(w, cpad, planes).reshape(-1)
            if l.get("wino_alt_off", -1) >= 0:       # the F(2x2) planes next to the F(2x4) ones (small-batch forwards)
                blob[l["wino_alt_off"]:l["wino_alt_off"] + 16 * cpad * cin] = winograd_weights(w, cpad, 16).reshape(-1)
    return blob


def conv_weights_f16(w, cout_pad):
    """[cout,cin,k,k] -> fp16 [cout_pad][k*k*cin] (OHWI, rows >= cout zero): the fp16 path's weight rows
    (include/orienmask_hip.h: om_layer_info.w16_off)."""
    cout, cin, k = w.shape[0], w.shape[1], w.shape[2]
    out = torch.zeros(cout_pad, k * k * cin, dtype=torch.float16, device=w.device)
    out[:cout] = w.detach().float().permute(0, 2, 3, 1).reshape(cout, -1).half()
    return out


def pack_state_dict_f16(state_dict, layers, total_halfs):
    """float16 tensor of total_halfs elements (on the weights' device): the convolution weights of every layer but the stem, rounded to
    fp16 (scale / shift / the stem stay in the float32 blob of pack_state_dict)."""
    sd = unwrap_checkpoint(state_dict)
    blob = torch.zeros(total_halfs, dtype=torch.float16, device=_blob_device(sd, layers))
    for l in layers:
        if l["w16_off"] < 0:
            continue
        w = sd[l["name"] + (".conv_block.0.weight" if l["has_bn"] else ".weight")]
        rows = conv_weights_f16(w, l["cout_pad"]).reshape(-1)
        blob[l["w16_off"]:l["w16_off"] + rows.numel()] = rows
    return blob


def split_f16_pairs(x):
    """float32 tensor [..., C] (C % 16 == 0) -> float16 tensor [..., C / 16, 2, 16]: per group of 16 channels the 16 hi halfs
    fp16(x), then the 16 lo halfs fp16(x - hi) -- the operand row layout of the split-operand Winograd GEMM
    (orienmask_amd/csrc/conv_wino24.hip, include/orienmask_hip.h: om_layer_info.wsplit_off)."""
    x = x.float()
    hi = x.half()
    lo = (x - hi.float()).half()
    lead = x.shape[:-1]
    g = x.shape[-1] // 16
    return torch.stack((hi.reshape(*lead, g, 16), lo.reshape(*lead, g, 16)), dim=-2).contiguous()


def winograd_weights_split(w, cout_pad):
    """[cout,cin,3,3] -> (fp16 [24][cout_pad][cin/16][2][16], exponents int32 [cout_pad]): the F(2x4,3x3) weights
    U * 2^e[cout] as hi/lo fp16 pairs.  e[cout] puts the largest |U| of the output channel into [2^13, 2^14): every element
    down to 2^-17 of it keeps ~22 significant bits, and fp16's 65504 is never reached; the epilogue's scale carries 2^-e."""
    u = winograd_weights(w, cout_pad, 24)                         # the fp32 U of the exact path
    e = _pow2_exponent(u.abs().amax(dim=(0, 2)))
    us = (u.double() * _pow2(e).view(1, -1, 1)).float()           # exact: a power of two
    return split_f16_pairs(us), e.to(torch.int32)


def winograd14_weights_split(w, cout_pad):
    """[cout,cin,3,3] -> (fp16 [cout_pad/64][cin/16][6][3][64][4][8], exponents int32 [cout_pad]): the weights of the fused
    F(4,3)-along-the-rows form (orienmask_amd/csrc/conv_wino14.hip), U[ky][j][n][c] = sum_kx G6[j][kx] w[n][c][ky][kx] computed in
    float64 and rounded to float32 once, times 2^e[n] (largest |U| of the output channel in [2^13, 2^14), as for the other split
    weights), as hi/lo fp16 pairs.  Layout: per 64-channel N tile, 16-channel chunk and transform point j the three kernel rows'
    64 x 64-byte rows [8 hi | 8 hi | 8 lo | 8 lo] -- 12 KiB that one weight group of the kernel's LDS-DMA ring fetches contiguously."""
    cout, cin = w.shape[0], w.shape[1]
    assert cout_pad % 64 == 0 and cin % 16 == 0
    u = torch.zeros(3, 6, cout_pad, cin, dtype=torch.float32, device=w.device)
    u[:, :, :cout] = _apply3(w.detach().double().permute(3, 2, 0, 1).contiguous(), _WINO_G6).transpose(0, 1).float()   # [j][r][n][c] -> [r][j]
    e = _pow2_exponent(u.abs().amax(dim=(0, 1, 3)))
    us = (u.double() * _pow2(e).view(1, 1, -1, 1)).float()         # exact: a power of two
    hi = us.half()
    lo = (us - hi.float()).half()

    def tiles(x):       # [3][6][cpad][cin] -> [cpad/64][cin/16][6][3][64][2][8]
        return x.reshape(3, 6, cout_pad // 64, 64, cin // 16, 2, 8).permute(2, 4, 1, 0, 3, 5, 6)

    return torch.cat((tiles(hi), tiles(lo)), dim=-2).contiguous(), e.to(torch.int32)


_SPLIT_PERM = [0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15]


def _pow2_row_scale(rows):
    """Exponents e[row] (float64 tensor) that put each row's largest magnitude into [2^13, 2^14); 0 for all-zero rows."""
    return _pow2_exponent(rows.abs().amax(dim=1))


def conv_weights_split(w, cout_pad):
    """[cout,cin,k,k] -> (fp16 [cout_pad][k*k][cin/16][4][8], exponents int32 [cout_pad]): the direct weights (OHWI rows)
    times 2^e[cout] as hi/lo fp16 pairs in the order conv_igemm_split.hip reads them -- per group of 16 input channels
    hi{0-3,8-11}, hi{4-7,12-15}, lo{0-3,8-11}, lo{4-7,12-15} (include/orienmask_hip.h: om_layer_info.wsplit_off)."""
    cout, cin, k = w.shape[0], w.shape[1], w.shape[2]
    rows = torch.zeros(cout_pad, k * k * cin, dtype=torch.float32, device=w.device)
    rows[:cout] = w.detach().float().permute(0, 2, 3, 1).reshape(cout, -1)
    e = _pow2_row_scale(rows)
    xs = (rows.double() * _pow2(e).view(-1, 1)).float()
    hi = xs.half()
    lo = (xs - hi.float()).half()
    g = cin // 16
    hi = hi.reshape(cout_pad, k * k, g, 16)[..., _SPLIT_PERM].reshape(cout_pad, k * k, g, 2, 8)
    lo = lo.reshape(cout_pad, k * k, g, 16)[..., _SPLIT_PERM].reshape(cout_pad, k * k, g, 2, 8)
    return torch.cat((hi, lo), dim=-2).contiguous(), e.to(torch.int32)


def pack_state_dict_split(state_dict, layers, total_words):
    """float32-typed tensor of total_words 4-byte words (on the weights' device): per layer (all but the stem) the split weights (two fp16 per
    word; the F(2x4) planes of the stride-1 3x3 layers, the direct weights of the others) and [cout_pad] floats scale * 2^-e,
    scale being the folded BatchNorm scale ROUNDED TO FLOAT32 exactly as pack_state_dict stores it (the two precision modes
    then differ in their products only)."""
    sd = unwrap_checkpoint(state_dict)
    blob = torch.zeros(total_words, dtype=torch.float32, device=_blob_device(sd, layers))
    for l in layers:
        if l.get("wsplit_off", -1) < 0:
            continue
        w, scale64, _ = folded_epilogue(sd, l)
        cpad, cin = l["cout_pad"], l["cin"]
        if l.get("wino_planes", 0) == 24:
            us, e = winograd14_weights_split(w, cpad)       # the fused F(4,3) form om_forward runs for these layers
        else:
            us, e = conv_weights_split(w, cpad)
        n = us.numel() // 2
        blob[l["wsplit_off"]:l["wsplit_off"] + n] = us.reshape(-1).view(torch.float32)
        scale = torch.zeros(cpad, dtype=torch.float64, device=blob.device)
        scale[:l["cout"]] = scale64.float().double()
        blob[l["wsplit_scale_off"]:l["wsplit_scale_off"] + cpad] = (scale * _pow2(-e)).float()
        if l.get("wsplit_direct_off", -1) >= 0:              # the latency mode's direct form of a stride-1 3x3 layer
            ud, ed = conv_weights_split(w, cpad)
            nd = ud.numel() // 2
            blob[l["wsplit_direct_off"]:l["wsplit_direct_off"] + nd] = ud.reshape(-1).view(torch.float32)
            blob[l["wsplit_direct_scale_off"]:l["wsplit_direct_scale_off"] + cpad] = \
                (scale * _pow2(-ed)).float()
    return blob
