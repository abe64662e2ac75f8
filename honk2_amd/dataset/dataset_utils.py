"""Split enum shared by datasets and the run scripts (reference ``dataset/dataset_utils.py:9-12``)."""
from enum import Enum


class DatasetType(Enum):
    TRAIN = "train"
    DEV = "dev"
    TEST = "test"
