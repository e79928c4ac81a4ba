"""Device-side image preprocessing: `process_images` with image_aspect_ratio == 'pad'
(reference LLaVA/llava/mm_utils.py:14-40) + CLIPImageProcessor.preprocess + the bf16 cast of
scene_graph_prediction_model_oracle.py:65-67, for decoded uint8 RGB views.

The host builds Pillow's resample tables (libImaging/Resample.c `precompute_coeffs` + `normalize_coeffs_8bpc`:
bicubic a = -0.5, support 2 x scale, double precision, 22-bit fixed point) and the 3 x 256 normalisation table
(float32 op order of CLIPImageProcessor: v * (1/255), - mean, / std; then bf16); the kernels (csrc/misc.hip) do the
padding, both integer resample passes and the table lookup. Bit-exact with `ClipImageProcessor._one` (PIL + numpy)."""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib

_P_BITS = 22


def _bicubic(x, a=-0.5):
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def resample_tables(in_size, out_size):
    """(bounds int32 (out, 2) [first tap, taps], coeffs int32 (out, ksize)) for in_size -> out_size."""
    scale = in_size / out_size
    fscale = max(scale, 1.0)
    support = 2.0 * fscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    coeffs = np.zeros((out_size, ksize), dtype=np.int32)
    inv = 1.0 / fscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        lo = max(int(center - support + 0.5), 0)
        n = min(int(center + support + 0.5), in_size) - lo
        k = [_bicubic((x + lo - center + 0.5) * inv) for x in range(n)]
        tot = 0.0
        for v in k:
            tot += v
        if tot != 0.0:
            k = [v / tot for v in k]
        for x, v in enumerate(k):
            coeffs[xx, x] = int(-0.5 + v * (1 << _P_BITS)) if v < 0 else int(0.5 + v * (1 << _P_BITS))
        bounds[xx] = (lo, n)
    return bounds, coeffs


class DevicePreprocessor:
    """views: list of uint8 (H, W, 3) arrays / tensors of ONE size (or a stacked (n, H, W, 3) tensor)
    -> (n, 3, size, size) bf16 on the device."""

    def __init__(self, image_processor, device):
        self.size = image_processor.crop_size['height']
        if image_processor.size.get('shortest_edge', self.size) != self.size:
            raise ValueError('the device path resizes the padded square straight to the crop size')
        self.device = torch.device(device)
        mean = np.asarray(image_processor.image_mean, dtype=np.float32)
        std = np.asarray(image_processor.image_std, dtype=np.float32)
        v = np.arange(256, dtype=np.float32) * np.float32(1.0 / 255.0)
        lut = np.stack([(v - mean[c]) / std[c] for c in range(3)]).astype(np.float32)
        self.lut = torch.from_numpy(lut).to(torch.bfloat16).to(self.device).contiguous()
        self.fill = (C.c_int32 * 3)(*[int(c * 255) for c in image_processor.image_mean])
        self._tables = {}
        self._tmp = None

    def _tab(self, side):
        if side not in self._tables:
            b, k = resample_tables(side, self.size)
            self._tables[side] = (torch.from_numpy(b).to(self.device), torch.from_numpy(k).to(self.device), k.shape[1])
        return self._tables[side]

    def __call__(self, views):
        if isinstance(views, (list, tuple)):
            # stack straight into pinned host memory (torch caches pinned blocks and keeps one alive until the async copy
            # that reads it has run), so the upload is one DMA that does not block the host
            first = np.asarray(views[0])
            if first.dtype != np.uint8 or first.ndim != 3 or first.shape[-1] != 3:
                raise ValueError('views must be uint8 (H, W, 3)')
            stage = torch.empty((len(views),) + first.shape, dtype=torch.uint8, pin_memory=self.device.type == 'cuda')
            dst = stage.numpy()
            for i, v in enumerate(views):
                v = np.asarray(v)
                if v.shape != first.shape:
                    raise ValueError('views of one call must have one size')
                dst[i] = v
            views = stage
        if views.dtype != torch.uint8 or views.dim() != 4 or views.shape[-1] != 3:
            raise ValueError('views must be uint8 (n, H, W, 3)')
        x = views.to(self.device, non_blocking=True).contiguous()
        n, h, w, _ = x.shape
        side = max(h, w)
        bounds, coeffs, ksize = self._tab(side)
        need = n * side * self.size * 3
        if self._tmp is None or self._tmp.numel() < need:
            self._tmp = torch.empty(need, dtype=torch.uint8, device=self.device)
        out = torch.empty(n, 3, self.size, self.size, dtype=torch.bfloat16, device=self.device)
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(_lib.lib().orc_preprocess_images(
            C.c_void_p(x.data_ptr()), n, h, w, self.size, C.c_void_p(bounds.data_ptr()), C.c_void_p(coeffs.data_ptr()), ksize,
            C.c_void_p(self.lut.data_ptr()), self.fill, C.c_void_p(self._tmp.data_ptr()), C.c_void_p(out.data_ptr()), s),
            'orc_preprocess_images')
        return out
