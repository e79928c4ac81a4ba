"""CLI of the adaptability evaluation, drop-in for the reference's
`python -u -m scene_graph_prediction.adaptability_4dor_eval --config <name>.json --model_path <dir>`
(scene_graph_prediction/adaptability_4dor_eval.py:46-58): same two flags, seed 42, cases read from ./adaptability_4dor,
one case per generate() call, classification report printed at the end."""
import argparse

from .main import config_loader, seed_everything
from .scene_graph_helpers.dataset.adaptability_or_dataset import AdaptabilityORDataset
from .scene_graph_helpers.model.adaptability_wrapper_oracle import AdaptabilityOracleWrapper


def main(argv=None, root_path='adaptability_4dor', **engine_kw):
    ap = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    ap.add_argument('--config', type=str, default='example.json', help='configuration file name. Relative path under given path')
    ap.add_argument('--model_path', type=str, default=None, help='path to model checkpoint')
    args = ap.parse_args(argv)
    seed_everything(42)
    config = config_loader(args.config)
    print(f'Model path: {args.model_path}')
    dataset = AdaptabilityORDataset(config, root_path)
    model = AdaptabilityOracleWrapper(config, labels=dataset.labels, model_path=args.model_path, **engine_kw)
    return model.validate([dataset[i]] for i in range(len(dataset)))


if __name__ == '__main__':
    import subprocess
    try:   # the reference lists GPUs with nvidia-smi here
        subprocess.call(['rocm-smi', '--showid'])
    except FileNotFoundError:
        pass
    main()
