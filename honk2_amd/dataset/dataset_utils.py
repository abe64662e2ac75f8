"""Split enum and the streaming-dataset base class (reference ``dataset/dataset_utils.py``).

``StreamingDataset`` keeps the reference's contract (``dataset/dataset_utils.py:20-98``): the audio files of a split are
shuffled once (``random.shuffle`` over (file, label) pairs) and concatenated into one long stream; item ``i`` is the
window ``stream[i*shift : i*shift + window]`` and its target is the label that covers most samples of the window (the
lowest label index wins ties); ``len()`` is ``int((total_num_samples - window) / shift)``.  A subclass sets
``audio_files, labels, sample_rate, label_mapping`` before calling ``super().__init__(config)`` and implements
``_load_sample(index) -> (samples, label)``, exactly as the reference's ``GSCStreamingDataset`` / ``HeySnipsStreamingDataset``
do (``dataset/gsc_dataset.py:179-217``).

The reference builds every window with Python list surgery (one label per SAMPLE, O(window) work per item) and only
supports sequential access.  Here the stream and a per-sample label array are extended file by file on demand, targets
come from per-label prefix counts, items can be read in any order, and ``stream_view()`` exposes the whole stream so
that ``AudioDataLoader`` can hand it to the GPU once: consecutive windows overlap by ``window - shift`` samples (99 % with
the shipped 1000 ms / 10 ms configs, ``config/gsc_dev_config.json:62-63``) and ``kws_mfcc_windows`` reads them in place.
"""
import random
from abc import ABC, abstractmethod
from enum import Enum

import numpy as np
from torch.utils.data import Dataset


class DatasetType(Enum):
    TRAIN = "train"
    DEV = "dev"
    TEST = "test"


def shuffle_in_groups(a, b):
    """Shuffle two equally long sequences with the same permutation (one ``random.shuffle`` call over the pairs)."""
    if len(a) != len(b):
        raise ValueError("shuffle_in_groups: sequences differ in length")
    pairs = list(zip(a, b))
    random.shuffle(pairs)
    return zip(*pairs)


class StreamingDataset(ABC, Dataset):
    def __init__(self, config):
        super().__init__()
        self.audio_files, self.labels = shuffle_in_groups(self.audio_files, self.labels)
        per_ms = int(self.sample_rate / 1000)
        self.shift_size = config['shift_size_ms'] * per_ms
        self.window_size = config['window_size_ms'] * per_ms
        self.num_samples = int((config['total_num_samples'] - self.window_size) / self.shift_size)
        self._n_labels = len(self.label_mapping)
        self._chunks = []            # loaded audio, one float32 array per file
        self._chunk_labels = []
        self._loaded = 0             # samples loaded so far
        self._next_file = 0
        self._stream = None          # concatenation cache of _chunks
        self._counts = None          # (n_labels, loaded + 1) prefix counts of per-sample labels

    @abstractmethod
    def _load_sample(self, index):
        """-> (1-D array of samples, int label) of audio file ``index`` of the shuffled order."""

    def __len__(self):
        return self.num_samples

    # ---- stream construction
    def _ensure(self, n_samples):
        grew = False
        while self._loaded < n_samples:
            if self._next_file >= len(self.labels):
                raise IndexError("StreamingDataset: the audio files hold fewer samples than total_num_samples says")
            data, label = self._load_sample(self._next_file)
            data = np.asarray(data, dtype=np.float32).reshape(-1)
            self._chunks.append(data)
            self._chunk_labels.append(int(label))
            self._loaded += len(data)
            self._next_file += 1
            grew = True
        if grew or self._stream is None:
            self._stream = np.concatenate(self._chunks) if self._chunks else np.zeros(0, np.float32)
            per_sample = np.repeat(np.asarray(self._chunk_labels, dtype=np.int64), [len(c) for c in self._chunks])
            onehot = per_sample[None, :] == np.arange(self._n_labels, dtype=np.int64)[:, None]
            self._counts = np.concatenate([np.zeros((self._n_labels, 1), np.int64), np.cumsum(onehot, axis=1)], axis=1)

    def _targets(self, first, count):
        lo = first * self.shift_size + np.arange(count, dtype=np.int64) * self.shift_size
        in_window = self._counts[:, lo + self.window_size] - self._counts[:, lo]
        return in_window.argmax(axis=0)        # first maximum = lowest label index, as the reference's strict '>' scan

    def __getitem__(self, index):
        if not 0 <= index < self.num_samples:
            raise IndexError(index)
        start = index * self.shift_size
        full = (self.num_samples - 1) * self.shift_size + self.window_size
        self._ensure(min(max(start + self.window_size, 2 * self._loaded), full))   # grow geometrically: O(total) rebuilds
        return self._stream[start:start + self.window_size], int(self._targets(index, 1)[0])

    def stream_view(self):
        """-> (stream float32[(num_samples - 1) * shift + window], window, shift, targets int64[num_samples])."""
        n = max(self.num_samples, 0)
        need = (n - 1) * self.shift_size + self.window_size if n else 0
        self._ensure(need)
        return self._stream[:need], self.window_size, self.shift_size, self._targets(0, n)
