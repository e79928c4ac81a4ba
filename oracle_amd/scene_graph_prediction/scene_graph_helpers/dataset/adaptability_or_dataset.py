"""Adaptability test cases from a directory tree (reference
scene_graph_prediction/scene_graph_helpers/dataset/adaptability_or_dataset.py:8-41):

    <root>/<label>/positives/<case>.json + <case>.jpg     negatives/<case>.jpg     visual_features/<vis_img_name>

Labels are the sorted sub-directory names; every JSON yields a positive and a negative item with the same test condition.
Item keys: image_path, vis_desc_path, test_condition, textual_attributes, label, is_pos."""
import json
from pathlib import Path


class AdaptabilityORDataset:
    def __init__(self, config, root_path='adaptability_4dor'):
        self.config = config
        self.root_path = Path(root_path)
        self.labels = [d.name for d in sorted(self.root_path.iterdir()) if d.is_dir()]
        self.all_test_cases = []
        for label in self.labels:
            for js in (self.root_path / label / 'positives').glob('*.json'):
                pos = js.with_suffix('.jpg').absolute()
                neg = (pos.parent.parent / 'negatives' / pos.name).absolute()
                with js.open('r') as f:
                    data = json.load(f)
                self.all_test_cases.append((data, pos, label, True))
                self.all_test_cases.append((data, neg, label, False))

    def __len__(self):
        return len(self.all_test_cases)

    def __getitem__(self, index):
        data, image_path, label, is_pos = self.all_test_cases[index]
        return {'image_path': image_path, 'vis_desc_path': image_path.parent.parent / 'visual_features' / data['vis_img_name'],
                'test_condition': data['test_condition'], 'textual_attributes': data['textual_attributes'], 'label': label,
                'is_pos': is_pos}
