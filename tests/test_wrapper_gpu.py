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


def test_adaptability_cli_end_to_end(tmp_path, monkeypatch, capsys):
    """adaptability_4dor_eval -> AdaptabilityOracleWrapper -> generate() on the native engine, visual-descriptor variant:
    8 / 7 descriptor embeddings spliced per prompt; the report must come out and repeat exactly on a second run."""
    import json

    import numpy as np
    import torch
    from PIL import Image
    from oracle_amd.scene_graph_prediction import adaptability_4dor_eval as cli
    monkeypatch.chdir(tmp_path)
    rng = np.random.default_rng(0)
    d_model = 512                                    # synthetic:tiny
    crops = tmp_path / 'synthetic_or_generation' / 'original_crops'
    crops.mkdir(parents=True)
    g = torch.Generator().manual_seed(1)
    for n in ('anesthesia equipment', 'cementing', 'cutting', 'drilling', 'hammering', 'sawing', 'suturing'):
        torch.save(torch.randn(d_model, generator=g) * 0.02, crops / f'{n}_cam2.pt')
    for label in ('drilling', 'kuka'):
        for sub in ('positives', 'negatives', 'visual_features'):
            (tmp_path / 'adaptability_4dor' / label / sub).mkdir(parents=True)
        base = tmp_path / 'adaptability_4dor' / label
        for sub in ('positives', 'negatives'):
            Image.fromarray(rng.integers(0, 256, size=(48, 64, 3), dtype=np.uint8)).save(base / sub / 'c0.jpg')
        (base / 'positives' / 'c0.json').write_text(json.dumps({
            'vis_img_name': 'v0.jpg', 'test_condition': f'<head surgeon>,<patient>,<{label}>',
            'textual_attributes': {'color': 'red', 'size': 'large', 'shape': 'round', 'texture': 'metal', 'object_type': 'robot'}}))
        torch.save(torch.randn(d_model, generator=g) * 0.02, base / 'visual_features' / 'v0_crop.pt')
    argv = ['--config', 'oracle_mv_learned_symbolic_visual.json', '--model_path', 'synthetic:tiny:chain']
    first = cli.main(argv)
    second = cli.main(argv)
    assert first == second and 'drilling' in first and 'kuka' in first


@pytest.mark.parametrize('config', ['oracle.json', 'oracle_mv_learned_symbolic.json', 'oracle_mv_learned_symbolic_visual.json',
                                    'oracle_mv_learned_temporal_pred.json', 'oracle_temporal_pred.json'])
def test_cli_every_shipped_config_runs_and_repeats(config, tmp_path, monkeypatch):
    """Every configuration file the reference ships for inference, through the CLI on the native engine (single view, symbolic,
    visual descriptors, online temporal memory with batch 1): predictions for every frame, identical on a second run."""
    from oracle_amd.scene_graph_prediction import main as cli
    monkeypatch.chdir(tmp_path)
    argv = ['--config', config, '--model_path', 'synthetic:tiny:chain', '--limit_frames', '6']
    first = cli.main(argv)
    second = cli.main(argv)
    assert len(first) == 6 and first == second


def test_cli_two_ranks_print_the_whole_split_metrics_once(tmp_path, monkeypatch, capsys):
    """Frame-sharded evaluate (BASELINE configs[4]) at world size 2 (two ranks sharing the one GPU over gloo): raw predictions
    and the per-take / overall classification reports must be those of the single-process run over the whole split --
    printed once by rank 0, not one partial report per shard. r3: the ranks exchange ONE fixed-shape int32 all-gather of
    [frames per rank, 1 + 300] token rows and rank 0 detokenises, parses and scores the whole split from it."""
    import re
    import socket
    import subprocess
    import sys
    from conftest import ROOT
    from oracle_amd.scene_graph_prediction import main as cli
    monkeypatch.chdir(tmp_path)
    argv = ['--config', 'oracle_mv_learned.json', '--model_path', 'synthetic:tiny:chain', '--limit_frames', '17']   # shards of 9 and 8: uneven, and rank 0 ends on a one-frame batch
    single = cli.main(argv)
    one = capsys.readouterr().out
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, ORC_DIST_BACKEND='gloo', PYTHONPATH=ROOT + os.pathsep + os.environ.get('PYTHONPATH', ''))
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
                        '--master-port', str(port), '-m', 'oracle_amd.scene_graph_prediction.main', '--mode', 'infer'] + argv,
                       capture_output=True, text=True, timeout=600, env=env, cwd=tmp_path)
    assert r.returncode == 0, r.stderr[-3000:]
    import json
    merged = json.load(open(tmp_path / 'scan_relations_oracle_mv_learned_test.json'))
    # 'infer' runs the unlabeled test split; the evaluate split must agree too:
    r2 = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
                         '--master-port', str(port), '-m', 'oracle_amd.scene_graph_prediction.main'] + argv,
                        capture_output=True, text=True, timeout=600, env=env, cwd=tmp_path)
    assert r2.returncode == 0, r2.stderr[-3000:]
    two = r2.stdout
    assert len(merged) == 17
    tail = lambda text: re.sub(r'\s+', ' ', text[text.rindex('val Results:'):]).strip()
    assert two.count('val Results:') == 1 and one.count('val Results:') == 1
    assert tail(one) == tail(two)
    takes = lambda text: sorted(re.findall(r'Take (\d+)', text))
    assert takes(one) == takes(two) and len(takes(one)) >= 1
    assert len(single) == 17
