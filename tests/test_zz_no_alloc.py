"""The allocation-free entry point orc_preprocess_images_noalloc (include/oracle_hip.h) against the existing
orc_preprocess_images, through the C ABI: same kernels, the inter-pass image in the library's static per-GPU scratch
instead of a caller workspace. Bit-exact (integer resampling + table lookup), two calls in a row, the second smaller
(the scratch keeps the first call's bytes), then a size the static buffer cannot hold. Collected last on purpose."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _call(views, noalloc):
    from oracle_amd import _lib
    from oracle_amd.llava.mm_utils import ClipImageProcessor
    from oracle_amd.preprocess import DevicePreprocessor
    l = _lib.lib()
    dev = torch.device('cuda')
    pre = DevicePreprocessor(ClipImageProcessor(), dev)
    x = torch.from_numpy(views).to(dev).contiguous()
    n, h, w, _ = x.shape
    bounds, coeffs, ksize = pre._tab(max(h, w))
    out = torch.full((n, 3, pre.size, pre.size), float('nan'), dtype=torch.bfloat16, device=dev)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())
    if noalloc:
        rc = l.orc_preprocess_images_noalloc(p(x), n, h, w, pre.size, p(bounds), p(coeffs), ksize, p(pre.lut), pre.fill, p(out), s)
    else:
        tmp = torch.empty(n * max(h, w) * pre.size * 3, dtype=torch.uint8, device=dev)
        rc = l.orc_preprocess_images(p(x), n, h, w, pre.size, p(bounds), p(coeffs), ksize, p(pre.lut), pre.fill, p(tmp), p(out), s)
    torch.cuda.synchronize()
    return rc, out


def test_noalloc_preprocess_matches_workspace_entry_point_twice():
    from oracle_amd import _lib
    l = _lib.lib()
    assert l.orc_preprocess_noalloc_capacity() == 16 << 20
    rng = np.random.default_rng(5)
    big = rng.integers(0, 256, size=(4, 1536, 2048, 3), dtype=np.uint8)       # 4 dataset-sized views: 8.3 MB of scratch
    small = rng.integers(0, 256, size=(2, 300, 200, 3), dtype=np.uint8)       # smaller second call, portrait
    for views in (big, small):
        rc0, ref = _call(views, noalloc=False)
        rc1, out = _call(views, noalloc=True)                                 # warm-up: torch's caching allocator now holds every block _call needs
        assert rc0 == 0 and rc1 == 0, l.orc_last_error()
        assert torch.equal(out.view(torch.int16), ref.view(torch.int16))
        assert not torch.isnan(out.float()).any()
        # the entry point itself takes no device memory: the driver's free-memory figure (hipMemGetInfo, which sees a hipMalloc
        # made inside the library, unlike torch's counters) is the same before and after a call that re-uses cached blocks
        del out, ref
        free0 = torch.cuda.mem_get_info()[0]
        rc2, out2 = _call(views, noalloc=True)
        free1 = torch.cuda.mem_get_info()[0]
        assert rc2 == 0 and free1 == free0, (free0, free1)
        del out2
    # beyond the static buffer: 9 views of side 2048 need 18.6 MB > 16 MiB -> refused with a clear message, nothing launched
    nine = np.zeros((9, 8, 2048, 3), dtype=np.uint8)
    rc, _ = _call(nine, noalloc=True)
    assert rc != 0 and b'static buffer' in l.orc_last_error()
    rc, out = _call(small, noalloc=True)                                      # still works afterwards
    assert rc == 0 and torch.equal(out.view(torch.int16), _call(small, noalloc=False)[1].view(torch.int16))
