from .dataset_utils import DatasetType, StreamingDataset, shuffle_in_groups
from .synthetic_dataset import SyntheticKWSDataset, SyntheticStreamingDataset
