"""Per-class accuracy keyed by integer label (reference ``metric/per_class_acc.py:7-55``).  Classes that never
occur as a target are absent from the result, as in the reference."""
import torch

from ..utils import register_cls
from .metric_utils import MacroMetric


@register_cls('metric.PerClassAcc')
class PerClassAcc(MacroMetric):
    def __init__(self):
        super().__init__()
        self.reset_metric()

    def accumulate(self, output, target):
        with torch.no_grad():
            pred = torch.argmax(output, dim=1)
            assert pred.shape[0] == len(target)
            pred = pred.tolist()
            target = target.tolist()
        total, correct = {}, {}
        for guess, truth in zip(pred, target):
            total[truth] = total.get(truth, 0) + 1
            correct[truth] = correct.get(truth, 0) + int(guess == truth)
        self.add_counts(correct, total)
        return {k: correct[k] / v for k, v in total.items()}

    def add_counts(self, correct, total):
        """correct/total: dict or sequence indexed by class; zero-total classes are skipped."""
        items = total.items() if isinstance(total, dict) else enumerate(total)
        for k, v in items:
            if int(v) == 0:
                continue
            self.total[k] = self.total.get(k, 0) + int(v)
            self.correct[k] = self.correct.get(k, 0) + int(correct[k])

    def get_metric(self):
        return {k: self.correct[k] / v for k, v in self.total.items()}

    def reset_metric(self):
        self.correct = {}
        self.total = {}
