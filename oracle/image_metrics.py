"""Image error measures of the parity tests and of bench.py's `parity` fields (north_star: "a stated per-pixel L2 / FLIP tolerance").

* rel_l1      sum |a - b| / sum |b|                                    (the suite's historical measure)
* rmse        sqrt(mean (a - b)^2) over pixels and channels            (per-pixel L2), also relative to the reference's mean
* mean_bias   |mean a - mean b| / mean b
* flip        a FLIP-class perceptual difference in [0, 1]: LDR-FLIP restated from the published algorithm (Andersson, Nilsson,
              Akenine-Moeller, Oskarsson, Astrom, Fairchild: "FLIP: A Difference Evaluator for Alternating Images", HPG 2020):
              contrast-sensitivity filtering in YCxCz, Hunt-adjusted HyAB colour distance with the paper's error redistribution,
              edge / point feature differences, combined as colour^(1 - feature); the image mean is reported.  No third-party code:
              numpy + scipy.ndimage.  Radiance is tone-mapped for it the way a viewer would see the frame: clip to [0, 1], sRGB OETF.
"""
import numpy as np

_PPD = 67.0  # pixels per degree of the paper's default viewing condition (0.7 m, 0.7 m wide 3840-pixel display)


def rel_l1(a, b):
    a, b = np.asarray(a, np.float64)[..., :3], np.asarray(b, np.float64)[..., :3]
    return float(np.abs(a - b).sum() / np.abs(b).sum())


def rmse(a, b):
    a, b = np.asarray(a, np.float64)[..., :3], np.asarray(b, np.float64)[..., :3]
    return float(np.sqrt(np.mean((a - b) ** 2)))


def mean_bias(a, b):
    a, b = np.asarray(a, np.float64)[..., :3], np.asarray(b, np.float64)[..., :3]
    return float(abs(a.mean() - b.mean()) / b.mean())


def srgb_oetf(x):
    x = np.clip(x, 0.0, 1.0)
    return np.where(x <= 0.0031308, 12.92 * x, 1.055 * np.power(x, 1.0 / 2.4) - 0.055)


def _srgb_eotf(x):
    return np.where(x <= 0.04045, x / 12.92, np.power((x + 0.055) / 1.055, 2.4))


_RGB2XYZ = np.array([[0.4124564, 0.3575761, 0.1804375], [0.2126729, 0.7151522, 0.0721750], [0.0193339, 0.1191920, 0.9503041]])
_XYZ2RGB = np.linalg.inv(_RGB2XYZ)
_WHITE = _RGB2XYZ @ np.ones(3)


def _linrgb_to_ycxcz(rgb):
    xyz = rgb @ _RGB2XYZ.T / _WHITE
    return np.stack([116.0 * xyz[..., 1] - 16.0, 500.0 * (xyz[..., 0] - xyz[..., 1]), 200.0 * (xyz[..., 1] - xyz[..., 2])], -1)


def _ycxcz_to_linrgb(ycc):
    y = (ycc[..., 0] + 16.0) / 116.0
    xyz = np.stack([y + ycc[..., 1] / 500.0, y, y - ycc[..., 2] / 200.0], -1) * _WHITE
    return xyz @ _XYZ2RGB.T


def _linrgb_to_lab(rgb):
    xyz = rgb @ _RGB2XYZ.T / _WHITE
    delta = 6.0 / 29.0
    f = np.where(xyz > delta ** 3, np.cbrt(np.maximum(xyz, 0.0)), xyz / (3.0 * delta * delta) + 4.0 / 29.0)
    return np.stack([116.0 * f[..., 1] - 16.0, 500.0 * (f[..., 0] - f[..., 1]), 200.0 * (f[..., 1] - f[..., 2])], -1)


def _hunt(lab):
    return np.stack([lab[..., 0], 0.01 * lab[..., 0] * lab[..., 1], 0.01 * lab[..., 0] * lab[..., 2]], -1)


def _hyab(a, b):
    d = a - b
    return np.abs(d[..., 0]) + np.sqrt(d[..., 1] ** 2 + d[..., 2] ** 2)


def _csf_kernels():
    # contrast sensitivity functions as sums of Gaussians in the spatial domain (x in degrees): a sqrt(pi / b) exp(-pi^2 x^2 / b)
    params = {"A": (1.0, 0.0047, 0.0, 1e-5), "RG": (1.0, 0.0053, 0.0, 1e-5), "BY": (34.1, 0.04, 13.5, 0.025)}
    b_max = 0.04
    radius = int(np.ceil(3.0 * np.sqrt(b_max / (2.0 * np.pi ** 2)) * _PPD))
    x = np.arange(-radius, radius + 1) / _PPD
    xx, yy = np.meshgrid(x, x)
    d2 = xx ** 2 + yy ** 2
    out = {}
    for k, (a1, b1, a2, b2) in params.items():
        g = a1 * np.sqrt(np.pi / b1) * np.exp(-np.pi ** 2 * d2 / b1) + a2 * np.sqrt(np.pi / b2) * np.exp(-np.pi ** 2 * d2 / b2)
        out[k] = g / g.sum()
    return out


def _feature_kernels():
    w = 0.082
    sd = 0.5 * w * _PPD
    radius = int(np.ceil(3.0 * sd))
    x = np.arange(-radius, radius + 1)
    xx, yy = np.meshgrid(x, x)
    g = np.exp(-(xx ** 2 + yy ** 2) / (2.0 * sd * sd))
    edge = -xx * g
    point = (xx ** 2 / (sd * sd) - 1.0) * g

    def normalise(k):  # positive weights sum to 1, negative weights to -1
        pos, neg = k[k > 0].sum(), -k[k < 0].sum()
        return np.where(k > 0, k / pos, k / neg)

    return normalise(edge), normalise(point)


_KERNELS = None


def flip(test_linear, reference_linear):
    """mean LDR-FLIP of two linear-radiance images (H, W, >=3) after clip + sRGB; 0 = identical, 1 = black against white"""
    from scipy.ndimage import convolve
    global _KERNELS
    if _KERNELS is None:
        _KERNELS = (_csf_kernels(), _feature_kernels())
    csf, (edge, point) = _KERNELS
    qc, qf, pc, pt = 0.7, 0.5, 0.4, 0.95

    def prepare(img):
        ldr = srgb_oetf(np.asarray(img, np.float64)[..., :3])  # what the display shows
        return _linrgb_to_ycxcz(_srgb_eotf(ldr))

    def filtered(ycc):
        f = np.stack([convolve(ycc[..., 0], csf["A"], mode="nearest"), convolve(ycc[..., 1], csf["RG"], mode="nearest"),
                      convolve(ycc[..., 2], csf["BY"], mode="nearest")], -1)
        return _hunt(_linrgb_to_lab(np.clip(_ycxcz_to_linrgb(f), 0.0, 1.0)))

    a, b = prepare(test_linear), prepare(reference_linear)
    colour = _hyab(filtered(a), filtered(b)) ** qc
    green, blue = _hunt(_linrgb_to_lab(np.array([[0.0, 1.0, 0.0]]))), _hunt(_linrgb_to_lab(np.array([[0.0, 0.0, 1.0]])))
    cmax = float(_hyab(green, blue)[0]) ** qc
    knee = pc * cmax
    colour = np.where(colour < knee, pt / knee * colour, pt + (colour - knee) / (cmax - knee) * (1.0 - pt))

    def features(ycc):
        y = (ycc[..., 0] + 16.0) / 116.0
        ex, ey = convolve(y, edge, mode="nearest"), convolve(y, edge.T, mode="nearest")
        px, py = convolve(y, point, mode="nearest"), convolve(y, point.T, mode="nearest")
        return np.hypot(ex, ey), np.hypot(px, py)

    (ea, pa), (eb, pb) = features(a), features(b)
    feature = (np.maximum(np.abs(ea - eb), np.abs(pa - pb)) / np.sqrt(2.0)) ** qf
    return float(np.mean(np.clip(colour, 0.0, 1.0) ** (1.0 - np.clip(feature, 0.0, 1.0))))


def summary(test_linear, reference_linear, with_flip=True):
    """the `parity` record of bench.py and of the full-size tests"""
    out = {"rel_l1": rel_l1(test_linear, reference_linear), "rmse": rmse(test_linear, reference_linear),
           "rmse_over_mean": rmse(test_linear, reference_linear) / float(np.asarray(reference_linear, np.float64)[..., :3].mean()),
           "mean_bias": mean_bias(test_linear, reference_linear)}
    if with_flip:
        out["flip"] = flip(test_linear, reference_linear)
    return out
