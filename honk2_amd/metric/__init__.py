from .acc import Acc
from .per_class_acc import PerClassAcc
from .metric_utils import MetricType, collect_metrics
