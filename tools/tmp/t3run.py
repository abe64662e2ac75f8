import sys, os
sys.path.insert(0, "/root/repo")
import torch
from honk2_amd.utils import find_cls
cfg = {"n_feature_maps": 45, "n_layers": 13, "use_dilation": True, "n_labels": 12, "dtype": os.environ.get("DT", "bf16")}
torch.manual_seed(3)
m = find_cls("model.ResNet")(cfg).cuda().eval()
x = torch.randn(1024, 101, 40, device="cuda") * 2.5 + 0.65
for _ in range(2): y = m(x)
torch.cuda.synchronize()
