#!/usr/bin/env python3
"""Golden vectors for the streaming-dataset mirror (honk2_amd/dataset/dataset_utils.py), generated with the reference's OWN
``StreamingDataset`` (``/root/reference/dataset/dataset_utils.py:20-98``, loaded by file path: the module only needs
numpy / torch) over the synthetic utterances of ``honk2_amd.dataset.SyntheticStreamingDataset``.

Test infrastructure only; run in the build container (the reference is not available on the GPU box):
    python oracle/gen_golden_streaming.py   ->  tests/golden/streaming_dataset.npz
"""
import importlib.util
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ.get("HONK2_REFERENCE", "/root/reference")

CASES = {
    "w1000_s10": dict(window_size_ms=1000, shift_size_ms=10, num_files=9, seed=77),
    "w400_s30": dict(window_size_ms=400, shift_size_ms=30, num_files=14, seed=5),
}
BASE = dict(sample_rate=16000, target_class=["yes", "no", "up"], unknown_class=True, silence_class=True, type="dev")


def main():
    spec = importlib.util.spec_from_file_location("ref_dataset_utils", os.path.join(REF, "dataset", "dataset_utils.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    from honk2_amd.dataset import SyntheticStreamingDataset

    class RefSynthetic(ref.StreamingDataset):
        """The reference base class over the same synthetic utterances (same attribute contract as GSCStreamingDataset)."""

        def __init__(self, config, donor):
            self.sample_rate = donor.sample_rate
            self.audio_files = list(range(len(donor._orig_labels)))
            self.labels = list(donor._orig_labels)
            self.label_mapping = donor.label_mapping
            self._donor = donor
            super().__init__(config)

        def _load_sample(self, index):
            # the donor generates utterance `audio_files[index]`; look it up by original file id
            return self._donor._utterance(self.audio_files[index], self.labels[index])

    out = {}
    for tag, case in CASES.items():
        cfg = dict(BASE, **case)
        random.seed(1234)
        mine = SyntheticStreamingDataset(dict(cfg))
        random.seed(1234)
        refds = RefSynthetic(dict(cfg, total_num_samples=cfg["num_files"] * 16000), mine)
        assert list(refds.audio_files) == list(mine.audio_files) and list(refds.labels) == list(mine.labels)
        n = len(refds)
        targets = np.zeros(n, np.int64)
        sums = np.zeros(n, np.float64)
        ends = np.zeros((n, 2), np.float64)
        for i in range(n):                       # the reference only supports sequential access
            w, t = refds[i]
            assert len(w) == refds.window_size
            targets[i] = t
            sums[i] = np.asarray(w, np.float64).sum()
            ends[i] = (w[0], w[-1])
        out[f"{tag}_targets"] = targets
        out[f"{tag}_sums"] = sums
        out[f"{tag}_ends"] = ends
        out[f"{tag}_len"] = np.int64(n)
        print(tag, "items", n, "label histogram", np.bincount(targets, minlength=5))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "streaming_dataset.npz"), **out)


if __name__ == "__main__":
    main()
