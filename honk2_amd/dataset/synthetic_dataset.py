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
from .dataset_utils import StreamingDataset

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


@register_cls('dataset.SyntheticStreamingDataset')
class SyntheticStreamingDataset(StreamingDataset):
    """Synthetic stand-in for ``GSCStreamingDataset`` (reference ``dataset/gsc_dataset.py:179-217``): utterances of
    0.4 - 1.0 s (zero-padded to one second like the reference pads short files, ``:213``) in a shuffled order form the
    stream; config keys ``window_size_ms, shift_size_ms`` as in ``config/gsc_dev_config.json:56-64``."""

    def __init__(self, config):
        self.sample_rate = config.get("sample_rate", 16000)
        self.seed = int(config.get("seed", 1234))
        n_files = int(config.get("num_files", 32))
        names = list(config.get("target_class", []))
        if config.get("unknown_class", False):
            names.append(LABEL_UNKNOWN)
        if config.get("silence_class", False):
            names.append(LABEL_SILENCE)
        self.label_mapping = dict(enumerate(names))
        rng = np.random.default_rng(self.seed)
        self.audio_files = list(range(n_files))
        self.labels = rng.integers(0, len(names), size=n_files).tolist()
        self._orig_labels = list(self.labels)                         # before the base class shuffles the stream order
        self.amplitude = float(config.get("amplitude", 0.1))
        config = dict(config)
        config['total_num_samples'] = n_files * self.sample_rate      # every file is padded to one second
        super().__init__(config)

    def _utterance(self, file_id, label):
        if self.label_mapping[label] == LABEL_SILENCE:
            return np.zeros(self.sample_rate, dtype=np.float32), label
        rng = np.random.default_rng([self.seed, int(file_id)])
        n = int(rng.integers(int(0.4 * self.sample_rate), self.sample_rate + 1))
        clip = np.zeros(self.sample_rate, dtype=np.float32)
        clip[:n] = np.clip(self.amplitude * rng.standard_normal(n), -1.0, 1.0)
        return clip, label

    def _load_sample(self, index):
        return self._utterance(self.audio_files[index], self.labels[index])
