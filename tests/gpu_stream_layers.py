"""runs each of the round-4 streaming kernels once per shape (no timing): the workload behind tests/gpu_pmc_stream.sh. Test infrastructure."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from gigagan_pytorch_amd import kernels as K   # noqa: E402
dev = torch.device('cuda', 0)
torch.manual_seed(0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
x32 = torch.randn(64, 256, 256, 32, device=dev).bfloat16(); dy32 = torch.randn(64, 256, 256, 32, device=dev).bfloat16()
x8 = torch.randn(64, 256, 256, 8, device=dev).bfloat16()
x64 = torch.randn(64, 128, 128, 64, device=dev).bfloat16(); dy64 = torch.randn(64, 128, 128, 64, device=dev).bfloat16()
w32 = (torch.randn(32, 288, device=dev) * 0.1).bfloat16(); w8 = (torch.randn(32, 72, device=dev) * 0.1).bfloat16()
w64 = (torch.randn(64, 576, device=dev) * 0.1).bfloat16(); b32 = torch.randn(32, device=dev); b64 = torch.randn(64, device=dev)
for _ in range(reps):
    K.conv2d_wgrad_nhwc(x32, dy32, ksize=3, force_tile=13)                      # 32 -> 32 @256x256 b=64: 537 MB algorithmic
    K.conv2d_wgrad_nhwc(x8, dy32, ksize=3, force_tile=13)                       # stem 8 -> 32: 336 MB
    K.conv2d_wgrad_nhwc(x64, dy64, ksize=3, force_tile=13)                      # 64 -> 64 @128x128: 268 MB
    K.conv2d_wgrad_nhwc(x32, dy32[:, ::2, ::2].contiguous(), ksize=2, stride=2, pad=0, force_tile=13)   # 2x2 / stride 2: 336 MB
    K.conv2d_nhwc(x32, w32, ksize=3, bias=b32, act='lrelu', force_tile=14)      # 537 MB
    K.conv2d_nhwc(x8, w8, ksize=3, bias=b32, act='lrelu', force_tile=14)        # 336 MB
    K.conv2d_nhwc(x64, w64, ksize=3, bias=b64, act='lrelu', force_tile=14)      # 268 MB
torch.cuda.synchronize()
