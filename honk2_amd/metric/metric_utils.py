"""Metric interfaces and result collection (reference ``metric/metric_utils.py:6-52``).

The reference's class names are swapped relative to the types they report (``MicroMetric`` reports
``MetricType.MACRO`` and vice versa, ``metric_utils.py:23-36``); only the ``get_type()`` values matter to
``collect_metrics`` and those are preserved: a scalar metric is reported as-is, a per-class metric is re-keyed
through ``label_mapping``.
"""
from abc import ABC, abstractmethod
from enum import Enum


class MetricType(Enum):
    MACRO = "MACRO"
    MICRO = "MICRO"


class Metric(ABC):
    @abstractmethod
    def accumulate(self, output, target):
        ...

    @abstractmethod
    def get_metric(self):
        ...

    @abstractmethod
    def reset_metric(self):
        ...


class MicroMetric(Metric):      # scalar-valued (name kept from the reference)
    def __init__(self):
        self.type = MetricType.MACRO

    def get_type(self):
        return self.type


class MacroMetric(Metric):      # dict-valued, one entry per class (name kept from the reference)
    def __init__(self):
        self.type = MetricType.MICRO

    def get_type(self):
        return self.type


def collect_metrics(metrics, label_mapping):
    results = {}
    for name, metric in metrics.items():
        value = metric.get_metric()
        if metric.get_type() == MetricType.MICRO:
            value = {label_mapping[k]: v for k, v in value.items()}
        results[f"metric_{name}"] = value
    return results
