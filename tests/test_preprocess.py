"""Image preprocessing. CPU: the numpy oracle (oracle/pil_resample.py) against Pillow itself, the
host mirror and the reference-run golden; the product's resample tables against the oracle's. GPU: the device path
(oracle_amd/preprocess.py -> orc_preprocess_images) bit-exact against the oracle and against the host mirror."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

from conftest import GOLDEN
from oracle import pil_resample as pr
from oracle_amd.llava.mm_utils import ClipImageProcessor, expand2square, process_images
from oracle_amd.preprocess import resample_tables

SIZES = [(1536, 2048), (480, 640), (700, 500), (336, 336), (400, 400), (90, 160)]


def _img(h, w, seed):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (h // 8 + 2, w // 8 + 2, 3), dtype=np.uint8)
    smooth = np.asarray(Image.fromarray(base).resize((w, h), Image.BILINEAR))          # image-like content
    noise = rng.integers(-20, 21, (h, w, 3))
    return np.clip(smooth.astype(int) + noise, 0, 255).astype(np.uint8)


@pytest.mark.parametrize('h,w', SIZES)
def test_oracle_resize_equals_pillow(h, w):
    img = _img(h, w, h + w)
    sq = pr.expand2square_u8(img, (122, 116, 104))
    assert np.array_equal(sq, np.asarray(expand2square(Image.fromarray(img), (122, 116, 104))))
    if sq.shape[0] != 336:
        assert np.array_equal(pr.resize_square_u8(sq, 336), np.asarray(Image.fromarray(sq).resize((336, 336), Image.BICUBIC)))


@pytest.mark.parametrize('h,w', SIZES)
def test_oracle_equals_host_mirror_and_tables_match(h, w):
    proc = ClipImageProcessor()
    cfg = type('C', (), {'image_aspect_ratio': 'pad'})()
    img = _img(h, w, 7)
    ref = process_images([Image.fromarray(img)], proc, cfg)[0].numpy()
    mine = pr.preprocess(img, proc.image_mean, proc.image_std)
    assert np.array_equal(ref, mine)
    b0, k0 = pr.precompute_coeffs(max(h, w), 336)
    b1, k1 = resample_tables(max(h, w), 336)
    assert np.array_equal(b0, b1) and np.array_equal(k0, k1)


def test_oracle_against_reference_run_golden():
    """wrapper_images.npz: the image tensor the REAL reference OracleWrapper handed to generate() (process_images through
    transformers' CLIPImageProcessor, then bf16) for the first batch of the 3-view case; the views are regenerated
    from the same seeds (oracle/gen_golden_wrapper.make_dataset)."""
    import json
    import tempfile
    from oracle.gen_golden_wrapper import make_dataset
    kat = json.load(open(os.path.join(GOLDEN, 'wrapper_kat.json')))['plain_mv_b3']
    g = np.load(os.path.join(GOLDEN, 'wrapper_images.npz'))
    data, _ = make_dataset(tempfile.mkdtemp(prefix='prep_'), len(kat['dataset']), [4, 8], False)
    proc = ClipImageProcessor()
    rows = []
    for elem in data[:kat['batch_size']]:
        views = [np.asarray(Image.open(elem['image_paths'][cam - 1]).convert('RGB')) for cam in kat['config']['CAMERAS']]
        rows.append(np.stack([pr.preprocess(v, proc.image_mean, proc.image_std) for v in views]))
    x = torch.from_numpy(np.stack(rows)).to(torch.bfloat16).float().numpy()
    assert list(x.shape) == list(g['plain_mv_b3.images0_shape'])
    assert np.abs(x[..., ::23, ::29] - g['plain_mv_b3.images0_sub']).max() <= 1e-2


@pytest.mark.gpu
@pytest.mark.parametrize('h,w', SIZES)
def test_device_preprocess_is_bit_exact(h, w):
    from oracle_amd.preprocess import DevicePreprocessor
    proc = ClipImageProcessor()
    cfg = type('C', (), {'image_aspect_ratio': 'pad'})()
    views = [_img(h, w, s) for s in (1, 2, 3)]
    dev = DevicePreprocessor(proc, 'cuda:0')
    out = dev(views).cpu()
    for i, v in enumerate(views):
        want = torch.from_numpy(pr.preprocess(v, proc.image_mean, proc.image_std)).to(torch.bfloat16)
        assert torch.equal(out[i], want)
    host = process_images([Image.fromarray(v) for v in views], proc, cfg).to(torch.bfloat16)[0]
    assert torch.equal(out, host)


@pytest.mark.gpu
def test_device_preprocess_of_jpeg_decoded_dataset_views(tmp_path):
    """The 4D-OR camera format: 2048 x 1536 JPEG files written by the synthetic dataset, decoded by Pillow
    (in the wrapper's decode threads), then padded / resized / normalised on the device: identical bits to the host path
    process_images(...).to(bfloat16) on the same decoded views."""
    from oracle_amd.preprocess import DevicePreprocessor
    from oracle_amd.scene_graph_prediction.scene_graph_helpers.dataset.synthetic_dataset import SyntheticORDataset
    ds = SyntheticORDataset({}, 'val', root=str(tmp_path / 'imgs'), n_frames=2, image_size=(2048, 1536), image_format='jpg')
    paths = ds[0]['image_paths'][:4]
    assert all(p.endswith('.jpg') for p in paths)
    views = [Image.open(p).convert('RGB') for p in paths]
    assert views[0].size == (2048, 1536)
    proc = ClipImageProcessor()
    cfg = type('C', (), {'image_aspect_ratio': 'pad'})()
    out = DevicePreprocessor(proc, 'cuda:0')([np.asarray(v) for v in views]).cpu()
    host = process_images(views, proc, cfg).to(torch.bfloat16)[0]
    assert torch.equal(out, host)
    assert os.path.getsize(paths[0]) < 2 << 20          # image-like content: a few hundred KB, like a camera JPEG
