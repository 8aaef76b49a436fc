"""gg_wgrads: launch time against the number of workgroups (= split-K slices = length of each workgroup's contiguous row run); test infrastructure."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from gigagan_pytorch_amd import kernels as K   # noqa: E402
from gpu_wgrads_ab import timed   # noqa: E402
dev = torch.device('cuda', 0)
for n, H, W, ci, co, ks in [(64, 256, 256, 32, 32, 3), (64, 128, 128, 8, 32, 1), (64, 128, 128, 64, 64, 3)]:
    x = torch.randn(n, H, W, ci, device=dev).bfloat16(); dy = torch.randn(n, H, W, co, device=dev).bfloat16()
    by = n * H * W * (ci + co) * 2
    for sk in (256, 255, 250, 240, 224, 200, 192, 128, 300, 384, 512):
        t = timed(lambda: K.conv2d_wgrad_nhwc(x, dy, ksize=ks, force_tile=13, force_splitk=sk))
        print(f'{ci}->{co} k{ks} @{H} b={n} splitk {sk:4d}: {t:7.1f} us {by / t / 1e6:5.2f} TB/s', flush=True)
