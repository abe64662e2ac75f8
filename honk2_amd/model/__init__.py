from .model_utils import BaseModel
from .resnet import ResNet
from .cnn import CNN
