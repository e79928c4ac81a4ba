"""TEST INFRASTRUCTURE (oracle): numpy restatement of the image preprocessing of the path --
`expand2square` (reference LLaVA/llava/mm_utils.py:14-25), then what `CLIPImageProcessor.preprocess` does to a square
RGB image: PIL bicubic resize to 336 (third-party Pillow, libImaging/Resample.c: two passes, horizontal then vertical,
uint8 in between, 22-bit fixed-point coefficients; published algorithm restated below), /255 and mean/std normalisation
in float32 (transformers image_transforms.rescale / normalize), cast to bf16 (reference
scene_graph_prediction_model_oracle.py:65-67). Pinned against Pillow itself and against the reference-run golden
tests/golden/wrapper_images.npz (tests/test_preprocess.py). Never imported by the product."""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def bicubic(x, a=-0.5):
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size, out_size, support=2.0):
    """-> (bounds (out,2) int32 [xmin, count], kk (out, ksize) int32 fixed-point). Resample.c precompute_coeffs +
    normalize_coeffs_8bpc, box = the whole image."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = support * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = [bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = sum(k)                      # C accumulates left to right in double, as sum() does
        if ww != 0.0:
            k = [v / ww for v in k]
        for x, v in enumerate(k):
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def clip8(v):
    return np.clip(v >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resize_square_u8(img, out_size):
    """img (S, S, 3) uint8 -> (out, out, 3) uint8, Pillow's ImagingResample for 8-bit images."""
    s = img.shape[0]
    assert img.shape[1] == s
    bounds, kk = precompute_coeffs(s, out_size)
    half = 1 << (PRECISION_BITS - 1)
    tmp = np.empty((s, out_size, 3), dtype=np.uint8)
    src = img.astype(np.int64)
    for xx in range(out_size):
        x0, n = bounds[xx]
        acc = half + (src[:, x0:x0 + n, :] * kk[xx, :n].astype(np.int64)[None, :, None]).sum(1)
        tmp[:, xx, :] = clip8(acc)
    out = np.empty((out_size, out_size, 3), dtype=np.uint8)
    src = tmp.astype(np.int64)
    for yy in range(out_size):
        y0, n = bounds[yy]
        acc = half + (src[y0:y0 + n, :, :] * kk[yy, :n].astype(np.int64)[:, None, None]).sum(0)
        out[yy] = clip8(acc)
    return out


def expand2square_u8(img, fill):
    h, w, _ = img.shape
    if h == w:
        return img
    s = max(h, w)
    out = np.empty((s, s, 3), dtype=np.uint8)
    out[:] = np.asarray(fill, dtype=np.uint8)
    out[(s - h) // 2:(s - h) // 2 + h, (s - w) // 2:(s - w) // 2 + w] = img
    return out


def normalise_lut(mean, std):
    """(3, 256) float32: fl32( fl32( fl32(v) * fl32(1/255) ) - mean ) / std ), the float32 op order of the host mirror."""
    v = np.arange(256, dtype=np.float32) * np.float32(1.0 / 255.0)
    return np.stack([(v - np.float32(m)) / np.float32(s) for m, s in zip(mean, std)]).astype(np.float32)


def preprocess(img, mean, std, out_size=336):
    """uint8 (H, W, 3) -> float32 (3, out, out) (the caller rounds to bf16)."""
    fill = tuple(int(c * 255) for c in mean)
    sq = expand2square_u8(img, fill)
    small = resize_square_u8(sq, out_size) if sq.shape[0] != out_size else sq
    lut = normalise_lut(mean, std)
    return np.stack([lut[c][small[:, :, c]] for c in range(3)])
