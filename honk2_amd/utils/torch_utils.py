"""Device selection and conv/pool output-size arithmetic (reference ``utils/torch_utils.py:9-65``)."""
import math
from collections.abc import Iterable

import torch

from .color_print import ColorEnum, print_color


def prepare_device(n_gpu_use):
    """Clamp the configured GPU count to what is present; returns (device, list of device ids)
    (reference ``utils/torch_utils.py:9-22``).  On this framework a GPU is mandatory for compute, so a CPU
    device is only useful for host-side plumbing."""
    n_gpu = torch.cuda.device_count()
    if n_gpu_use > 0 and n_gpu == 0:
        print_color(ColorEnum.YELLOW, "No GPU visible: honk2_amd has no CPU compute path, forward() will raise.")
        n_gpu_use = 0
    if n_gpu_use > n_gpu:
        print_color(ColorEnum.YELLOW, f"Warning: {n_gpu_use} GPUs configured but only {n_gpu} present.")
        n_gpu_use = n_gpu
    device = torch.device("cuda:0" if n_gpu_use > 0 else "cpu")
    return device, list(range(n_gpu_use))


def _per_dim(x, n):
    return tuple(x) if isinstance(x, Iterable) else (x,) * n


def _out_size(input_size, kernel_size, stride, padding, dilation, ceil_mode=False):
    n = len(input_size)
    stride, padding, dilation = _per_dim(stride, n), _per_dim(padding, n), _per_dim(dilation, n)
    rounder = math.ceil if ceil_mode else math.floor
    out = []
    for i, size in enumerate(input_size):
        span = dilation[i] * (kernel_size[i] - 1) + 1
        out.append(rounder((size + 2 * padding[i] - span) / stride[i] + 1))
    return out


def calculate_conv_output_size(input_size, kernel_size, stride=1, padding=0, dilation=1):
    return _out_size(input_size, kernel_size, stride, padding, dilation)


def calculate_pool_output_size(input_size, kernel_size, stride=None, padding=0, dilation=1, ceil_mode=False):
    return _out_size(input_size, kernel_size, kernel_size if stride is None else stride, padding, dilation, ceil_mode)
