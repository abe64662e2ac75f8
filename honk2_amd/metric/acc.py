"""Top-1 accuracy (reference ``metric/acc.py:7-31``)."""
import torch

from ..utils import register_cls
from .metric_utils import MicroMetric


@register_cls('metric.Acc')
class Acc(MicroMetric):
    def __init__(self):
        super().__init__()
        self.reset_metric()

    def accumulate(self, output, target):
        with torch.no_grad():
            pred = torch.argmax(output, dim=1)
            assert pred.shape[0] == len(target)
            correct = int((pred == target.to(pred.device)).sum().item())
        total = len(target)
        self.add_counts(correct, total)
        return correct / total

    def add_counts(self, correct, total):
        self.correct += int(correct)
        self.total += int(total)

    def get_metric(self):
        return self.correct / self.total

    def reset_metric(self):
        self.correct = 0
        self.total = 0
