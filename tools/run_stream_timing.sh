cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
KWS_LIB=$PWD/honk2_amd/variants/lib_stream_t.so timeout -k 10 120 python - <<'PY' 2>&1 | grep -v amdgpu.ids | grep "stream L" | sort | uniq -c | sort -rn | head -40
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from honk2_amd.utils import find_cls
from oracle import weights
cfg = {"n_feature_maps": 45, "n_layers": 13, "use_dilation": True, "n_labels": 12}
sd = weights.make_state_dict("ResNet", cfg, seed=11)
m = find_cls("model.ResNet")(dict(cfg, dtype="bf16"))
m.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()})
m = m.cuda().eval()
x = torch.randn(1024, 101, 40, device="cuda")
y = m(x); torch.cuda.synchronize()
PY
