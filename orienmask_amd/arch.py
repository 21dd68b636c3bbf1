"""Layer table of the OrienMaskYOLOFPNPlus inference graph (host-side view).

This is the Python view of the network the HIP library executes: one entry per
convolution, named with the reference's module prefixes so that the reference's
524-key ``state_dict`` maps onto it one-to-one.

Reference structure this table restates (not copied; rebuilt from the shapes):
  * DarkNet-53 stages, blocks 1/2/8/8/4      /root/reference/model/backbone/darknet.py:18-54
  * necks / routes / heads / skips            /root/reference/model/orienmask_yolo_fpnplus.py:9-72
  * conv -> BN(eps=1e-5) -> LeakyReLU(0.1)    /root/reference/model/base.py:104-137,278-279

The C++ side (``csrc/om_graph.cpp``) builds the same graph independently;
``pack.py`` cross-checks the two through ``om_model_layer_info``.
"""
from collections import namedtuple

ConvSpec = namedtuple("ConvSpec", "name cin cout ksize stride bn")

BN_EPS = 1e-5
LEAKY_SLOPE = 0.1
DARKNET_STAGES = ((2, 32, 1), (3, 64, 2), (4, 128, 8), (5, 256, 8), (6, 512, 4))


def _neck(prefix, cin, cout):
    chans = [(cin, cout, 1), (cout, cout * 2, 3), (cout * 2, cout, 1),
             (cout, cout * 2, 3), (cout * 2, cout, 1)]
    return [ConvSpec("%s.%d" % (prefix, i), a, b, k, 1, True) for i, (a, b, k) in enumerate(chans)]


def fpnplus_convs(num_anchors=3, num_classes=80):
    """Ordered list of the 90 convolutions of the FPNPlus model."""
    return model_convs("OrienMaskYOLOFPNPlus", num_anchors, num_classes)


def yolo_convs(num_anchors=3, num_classes=80):
    """The 87 convolutions of the non-Plus OrienMaskYOLO (/root/reference/model/orienmask_yolo.py:8-86)."""
    return model_convs("OrienMaskYOLO", num_anchors, num_classes)


def model_convs(model, num_anchors=3, num_classes=80):
    plus = model == "OrienMaskYOLOFPNPlus"
    if not plus and model != "OrienMaskYOLO":
        raise ValueError(model)
    convs = [ConvSpec("backbone.conv1", 3, 32, 3, 1, True)]
    for idx, ch, nblocks in DARKNET_STAGES:
        stage = "backbone.conv%d" % idx
        convs.append(ConvSpec(stage + ".0", ch, ch * 2, 3, 2, True))
        for j in range(1, nblocks + 1):
            convs.append(ConvSpec("%s.%d.conv.0" % (stage, j), ch * 2, ch, 1, 1, True))
            convs.append(ConvSpec("%s.%d.conv.1" % (stage, j), ch, ch * 2, 3, 1, True))
    convs += _neck("neck32", 1024, 512)
    convs += _neck("neck16", 768, 256)
    convs += _neck("neck8", 384, 128)
    convs += _neck("neck4", 256 if plus else 192, 128)
    convs.append(ConvSpec("route32.0", 512, 256, 1, 1, True))
    convs.append(ConvSpec("route16.0", 256, 128, 1, 1, True))
    if not plus:
        convs.append(ConvSpec("route8.0", 128, 64, 1, 1, True))
    bbox_dim = num_anchors * (5 + num_classes)
    for s, c in ((8, 128), (16, 256), (32, 512)):
        convs.append(ConvSpec("bbox_head%d.0" % s, c, c * 2, 3, 1, True))
        convs.append(ConvSpec("bbox_head%d.1" % s, c * 2, bbox_dim, 1, 1, False))
    if plus:
        convs.append(ConvSpec("skip32.0", 512, 64, 1, 1, True))
        convs.append(ConvSpec("skip16.0", 256, 64, 1, 1, True))
        convs.append(ConvSpec("skip8.0", 128, 64, 1, 1, True))
        convs.append(ConvSpec("skip4", 128, 64, 1, 1, True))
    for i, (a, b, k) in enumerate([(128, 256, 3), (256, 128, 1), (128, 256, 3),
                                   (256, 128, 1), (128, 256, 3)]):
        convs.append(ConvSpec("orien_head.%d" % i, a, b, k, 1, True))
    convs.append(ConvSpec("orien_head.5", 256, num_anchors * 6, 1, 1, False))
    return convs


def state_dict_entries(spec):
    """(key, shape, role) triples a ConvSpec contributes to the reference state_dict."""
    if spec.bn:
        p = spec.name + ".conv_block"
        return [
            (p + ".0.weight", (spec.cout, spec.cin, spec.ksize, spec.ksize), "conv_w"),
            (p + ".1.weight", (spec.cout,), "bn_gamma"),
            (p + ".1.bias", (spec.cout,), "bn_beta"),
            (p + ".1.running_mean", (spec.cout,), "bn_mean"),
            (p + ".1.running_var", (spec.cout,), "bn_var"),
            (p + ".1.num_batches_tracked", (), "bn_count"),
        ]
    return [
        (spec.name + ".weight", (spec.cout, spec.cin, spec.ksize, spec.ksize), "conv_w"),
        (spec.name + ".bias", (spec.cout,), "conv_b"),
    ]


def is_residual_tail(spec):
    """True for the 3x3 conv that closes a DarkNet residual block."""
    return spec.name.startswith("backbone.") and spec.name.endswith(".conv.1")


def layer_div(spec):
    """Spatial divisor (image / div) of a convolution's OUTPUT; its input is at div / stride."""
    n = spec.name
    if n == "backbone.conv1":
        return 1
    if n.startswith("backbone.conv"):
        return 2 ** (int(n[len("backbone.conv")]) - 1)
    if n.startswith("skip4") or n.startswith("neck4") or n.startswith("orien_head"):
        return 4
    if n.startswith("route8"):
        return 8
    for s in (32, 16, 8):
        if n.split(".")[0].endswith(str(s)):
            return s
    raise ValueError(n)


def layer_work(spec, batch, height, width):
    """Algorithmic work of one fused convolution launch (SURVEY.md section 8d accounting):
    flops = 2 * MACs; bytes = input read once + output written once + residual re-read once
    + weights once, all fp32."""
    div = layer_div(spec)
    ho, wo = height // div, width // div
    hi, wi = ho * spec.stride, wo * spec.stride
    m = batch * ho * wo
    macs = m * spec.cout * spec.cin * spec.ksize * spec.ksize
    elems = batch * hi * wi * spec.cin + m * spec.cout + (m * spec.cout if is_residual_tail(spec) else 0)
    weights = spec.cout * spec.cin * spec.ksize * spec.ksize
    return dict(flops=2 * macs, bytes=4 * (elems + weights), m=m)
