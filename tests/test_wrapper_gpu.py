"""GPU: the drop-in entry point end to end (scene_graph_prediction.main -> OracleWrapper -> generate() -> native engine)
on the synthetic dataset, with the device-side image preprocessing against the PIL host path: identical raw predictions."""
import os

import pytest

pytestmark = pytest.mark.gpu


def test_cli_end_to_end_device_vs_host_preprocessing(tmp_path, monkeypatch):
    from oracle_amd.scene_graph_prediction import main as cli
    from oracle_amd.scene_graph_prediction.scene_graph_helpers.model import scene_graph_prediction_model_oracle as wmod
    monkeypatch.chdir(tmp_path)
    argv = ['--config', 'oracle_mv_learned.json', '--model_path', 'synthetic:tiny:chain', '--limit_frames', '10']
    dev = cli.main(argv)
    orig = wmod.OracleWrapper.__init__

    def host_init(self, *a, **k):
        k['device_preprocess'] = False
        orig(self, *a, **k)
    monkeypatch.setattr(wmod.OracleWrapper, '__init__', host_init)
    host = cli.main(argv)
    assert len(dev) == 10 and dev == host
