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


ROW_TOL = 1e-2
_row_worst = {}


def rows_ok(a, b, what, tol=ROW_TOL):
    """Per-row companion of rel_err (a wrong small-magnitude row can hide behind a large one in a global max
    norm): every row (last dim; all leading dims flattened) within `tol` in RMS-relative error of ITS OWN reference row. Rows whose
    reference is (near) zero against the tensor's scale -- padding -- are compared in absolute terms against that scale instead."""
    a = a.detach().double().cpu().reshape(-1, a.shape[-1])
    b = b.detach().double().cpu().reshape(-1, b.shape[-1])
    scale = b.pow(2).mean().sqrt().clamp_min(1e-30)
    nb = b.pow(2).mean(-1).sqrt()
    err = (a - b).pow(2).mean(-1).sqrt()
    rel = err / torch_max(nb, 1e-3 * scale)
    worst = float(rel.max())
    _row_worst[what] = max(_row_worst.get(what, 0.0), worst)
    assert worst <= tol, (what, worst, int(rel.argmax()), float(nb[rel.argmax()]), float(scale))
    return worst


def torch_max(x, floor):
    return x.clamp_min(float(floor))


def pytest_terminal_summary(terminalreporter):
    if _row_worst:
        terminalreporter.write_line('worst per-row RMS-relative error by stage: ' + ', '.join(f'{k} {v:.4f}' for k, v in sorted(_row_worst.items())))


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
