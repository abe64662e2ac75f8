from .audio_processor import AudioProcessor
from .class_registry import register_cls, find_cls, install_into
from .color_print import ColorEnum, print_color
from .file_utils import ensure_dir, load_json, save_json, load_pkl, save_pkl
from .torch_utils import calculate_conv_output_size, calculate_pool_output_size, prepare_device
from .checkpoint import load_checkpoint_state
