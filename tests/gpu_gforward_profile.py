"""a few no-grad generator forwards at config 2 / batch 32 (for `rocprofv3 --kernel-trace --stats`: exact per-kernel durations of
the adaptive-conv forward path). Test infrastructure.   python tests/gpu_gforward_profile.py [n]"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device('cuda', 0)
gan = bench.build_gan(256, dev, use_hip_graphs=False)
with torch.no_grad():
    for _ in range(n):
        gan.G(noise=torch.randn(32, 64, device=dev))
torch.cuda.synchronize()
print('done')
