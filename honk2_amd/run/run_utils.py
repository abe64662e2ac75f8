"""Seeding, config merging and the dataset/data-loader factory (reference ``run/run_utils.py:15-49``).

Importing this module imports every plugin package so their ``register_cls`` decorators run, as the reference's
module-level imports do (``run/run_utils.py:7-11``).
"""
import copy
import random

import numpy as np
import torch

from .. import dataset as dataset_modules          # noqa: F401
from .. import data_loader as data_loader_modules  # noqa: F401
from .. import model as model_modules              # noqa: F401
from .. import metric as metric_modules            # noqa: F401
from .. import loss_function as loss_fn_modules    # noqa: F401
from ..utils import find_cls


def set_seed(seed):
    torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)


def merge_configs(base_config, additional_config):
    """Shallow override: keys of ``additional_config`` replace those of a deep copy of ``base_config``."""
    merged = copy.deepcopy(base_config)
    merged.update(additional_config)
    return merged


def init_data_loader(config, type):
    split = config["datasets"][type.value]
    dataset_name = split["dataset"]["name"]
    dataset_config = merge_configs(config[dataset_name], split["dataset"]["config"])
    for key in ("target_class", "unknown_class", "silence_class"):
        dataset_config[key] = config[key]
    dataset_config["type"] = type
    dataset_class = find_cls(f"dataset.{dataset_name}")
    if dataset_class is None:
        raise KeyError(f"dataset.{dataset_name} is not registered: disk-backed honk2 datasets are outside the "
                       "MI355X hot path; register your own Dataset under that key or use SyntheticKWSDataset")
    dataset = dataset_class(dataset_config)

    loader_name = split["data_loader"]["name"]
    loader_config = merge_configs(config[loader_name], split["data_loader"]["config"])
    return find_cls(f"data_loader.{loader_name}")(loader_config, dataset)
