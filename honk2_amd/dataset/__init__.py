from .dataset_utils import DatasetType
from .synthetic_dataset import SyntheticKWSDataset
