"""ANSI coloured warnings (reference ``utils/color_print.py``)."""
from enum import Enum


class ColorEnum(Enum):
    RED = "\033[91m"
    GREEN = "\033[92m"
    YELLOW = "\033[93m"
    BLUE = "\033[94m"
    END = "\033[0m"


def print_color(color, *msg):
    print(color.value + " ".join(str(m) for m in msg) + ColorEnum.END.value)
