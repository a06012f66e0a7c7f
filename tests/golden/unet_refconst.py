"""The reference's U-Net at its own constants (model/cifar_unet.c:26-46): configuration, the parameter tensors in bucket order
(bla_unet_tensor_info enumerates the same list -- tests/test_unet_model.py checks that), and the seeded parameters / inputs that
tools/unet_conditioning.py, the tests and bench.py all use, so that the committed fp64 prediction (tests/golden/unet_refconst.npz)
belongs to exactly these values.  Data only: no reference code, no device code."""
import numpy as np

from inputs import uniform

CFG = dict(image_h=32, image_w=32, in_channels=3, dims=[128, 256, 256, 256], time_dim=512, kernel=3, group_size=32, key_dim=16)


def tensor_list(cfg):
    """[(name, shape)] in the order of first use by forward() (model/cifar_unet.c:1099-1166)"""
    D, k, t, d, cin0 = cfg["dims"], cfg["kernel"], cfg["time_dim"], cfg["key_dim"], cfg["in_channels"]
    out = []

    def res(name, cin, cout):
        out.append((name + ".conv_1_kernels", (cout, cin, k, k)))
        out.append((name + ".conv_2_kernels", (cout, cout, k, k)))
        out.append((name + ".time_weights", (t, cout)))
        out.append((name + ".time_biases", (cout,)))
        if cin != cout:
            out.append((name + ".residual_conv_kernels", (cout, cin, 1, 1)))

    def att(name, c):
        for p in ("Q_proj", "K_proj", "V_proj"):
            out.append((f"{name}.{p}", (c, d)))
        out.append((name + ".weights", (d, c)))
        out.append((name + ".biases", (c,)))

    def conv(name, cin, cout, present=True):
        if present:
            out.append((name, (cout, cin, k, k)))
    res("down_1_resnet_1", cin0, D[0]); res("down_1_resnet_2", D[0], D[0]); conv("down_1_conv_kernels", D[0], D[1])
    res("down_2_resnet_1", D[1], D[1]); att("down_2_self_attention_1", D[1]); res("down_2_resnet_2", D[1], D[1]); att("down_2_self_attention_2", D[1])
    conv("down_2_conv_kernels", D[1], D[2])
    res("down_3_resnet_1", D[2], D[2]); res("down_3_resnet_2", D[2], D[2]); conv("down_3_conv_kernels", D[2], D[3])
    res("down_4_resnet_1", D[3], D[3]); res("down_4_resnet_2", D[3], D[3])
    res("mid_resnet_1", D[3], D[3]); att("mid_self_attention", D[3]); res("mid_resnet_2", D[3], D[3])
    res("up_1_resnet_1", 2 * D[3], D[3]); res("up_1_resnet_2", D[3], D[3]); conv("up_1_conv_kernels", D[3], D[2], D[3] != D[2])
    res("up_2_resnet_1", 2 * D[2], D[2]); res("up_2_resnet_2", D[2], D[2]); conv("up_2_conv_kernels", D[2], D[1], D[2] != D[1])
    res("up_3_resnet_1", 2 * D[1], D[1]); att("up_3_self_attention_1", D[1]); res("up_3_resnet_2", D[1], D[1]); att("up_3_self_attention_2", D[1])
    conv("up_3_conv_kernels", D[1], D[0], D[1] != D[0])
    res("up_4_resnet_1", 2 * D[0], D[0]); res("up_4_resnet_2", D[0], D[0])
    conv("output_conv_kernels", D[0], cin0)
    return out


def make_params(cfg, seed=7000):
    """name -> float32 array: uniform(-s, s), s = sqrt(3 / fan_in) (biases: 0.05) -- unit-variance-preserving, so 36 norm layers stay in range"""
    P = {}
    for i, (name, shp) in enumerate(tensor_list(cfg)):
        fan_in = int(np.prod(shp[1:])) if len(shp) > 1 else shp[0]
        scale = 0.05 if name.endswith("biases") else float(np.sqrt(3.0 / fan_in))
        P[name] = uniform(seed + i, shp, -scale, scale, np.float32)
    return P


def make_inputs(image, cfg=CFG):
    """(x, time embedding, noise) of image number `image`, float32"""
    c, h, w, t = cfg["in_channels"], cfg["image_h"], cfg["image_w"], cfg["time_dim"]
    return (uniform(9100 + 3 * image, (c, h, w), -1, 1, np.float32), uniform(9101 + 3 * image, (t,), -1, 1, np.float32),
            uniform(9102 + 3 * image, (c, h, w), -1, 1, np.float32))
