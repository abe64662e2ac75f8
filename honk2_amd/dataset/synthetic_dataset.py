"""Synthetic stand-in for the Google Speech Commands corpus (there is no dataset on the build or GPU boxes).

Same item contract as the reference's ``GSCDataset.__getitem__`` (``dataset/gsc_dataset.py:163-174``): returns
``(np.float32[sample_rate], int label)``; exposes ``label_mapping`` (int -> class name) built like
``GSCDatasetPreprocessor`` builds it (targets, then ``__unknown__``, then ``__silence__``).  Clips of the silence
class are exact zeros, as in the reference (``:165-166``).  Disk-backed datasets (wav decoding, speaker-hash
splits, noise mixing) are outside the hot path and are not reimplemented here.
"""
import numpy as np
from torch.utils.data import Dataset

from ..utils import register_cls

LABEL_SILENCE = "__silence__"
LABEL_UNKNOWN = "__unknown__"


@register_cls('dataset.SyntheticKWSDataset')
class SyntheticKWSDataset(Dataset):
    def __init__(self, config):
        super().__init__()
        self.sample_rate = config.get("sample_rate", 16000)
        self.num_samples = int(config.get("num_samples", 1024))
        self.seed = int(config.get("seed", 1234))
        names = list(config.get("target_class", []))
        if config.get("unknown_class", False):
            names.append(LABEL_UNKNOWN)
        if config.get("silence_class", False):
            names.append(LABEL_SILENCE)
        self.label_mapping = dict(enumerate(names))
        rng = np.random.default_rng(self.seed)
        self.labels = rng.integers(0, len(names), size=self.num_samples).tolist()
        self.amplitude = float(config.get("amplitude", 0.1))

    def __len__(self):
        return self.num_samples

    def __getitem__(self, index):
        label = self.labels[index]
        if self.label_mapping[label] == LABEL_SILENCE:
            return np.zeros(self.sample_rate, dtype=np.float32), label
        rng = np.random.default_rng([self.seed, index])
        clip = np.clip(self.amplitude * rng.standard_normal(self.sample_rate), -1.0, 1.0).astype(np.float32)
        return clip, label
