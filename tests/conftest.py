import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def rel_err(a, b):
    """max |a-b| / max |b| -- the 'max rel err' of BASELINE.json's north_star (tolerance 1e-2 for bf16)."""
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-20))


def rms_rel_rows(a, b):
    """Per-row RMS-relative error ||a - b||_2 / ||b||_2 over the last dim: unlike the max norm it is an
    average over every compared element, so it does not sit on the luck of one logit, and it is reported per row so that a
    wrong small-magnitude row cannot hide behind a large one."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return ((a - b).pow(2).sum(-1).sqrt() / b.pow(2).sum(-1).sqrt().clamp_min(1e-30)).float()


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
