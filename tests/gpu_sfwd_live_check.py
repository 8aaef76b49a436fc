"""runs training steps of a bench workload and, for every convolution launch the planner puts on gg_sfwd (tile 14), repeats it on the 4-wave
implicit GEMM (force_tile 3) on the SAME live operands; reports the first launches whose results differ (or hold non-finite values the
other path does not). Test infrastructure.   python tests/gpu_sfwd_live_check.py text|uncond|upsampler [steps]"""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench   # noqa: E402
from gigagan_pytorch_amd import kernels as K   # noqa: E402
from gigagan_pytorch_amd.data import SyntheticImages   # noqa: E402
from gigagan_pytorch_amd.gigagan import cycle   # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else 'text'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device('cuda', 0)
B = 32 if workload == 'uncond' else 16
gan = bench.build_gan(256, dev, use_hip_graphs=False, workload=workload)
it = iter(bench.SyntheticTextImages(B, 256, dev)) if workload == 'text' else cycle(SyntheticImages(B, 256, device=dev))
orig = K.conv2d_nhwc
bad, seen = [], 0


def checked(x, w, **kw):
    global seen
    y = orig(x, w, **kw)
    if kw.get('force_tile') or kw.get('plan_only'):
        return y
    tile, sk = orig(x, w, **dict(kw, plan_only=True))
    if tile == 14:
        seen += 1
        ref = orig(x, w, **dict(kw, force_tile=3))
        fy, fr = torch.isfinite(y).all().item(), torch.isfinite(ref).all().item()
        diff = float((y.float() - ref.float()).abs().nan_to_num(1e30).max())
        if (not fy and fr) or diff > 0.25 * max(float(ref.float().abs().max()), 1e-3):
            bad.append(dict(step=gan._steps_host, x=tuple(x.shape), w=tuple(w.shape), finite_sfwd=fy, finite_ref=fr, max_abs_diff=diff,
                            x_finite=bool(torch.isfinite(x).all()), x_absmax=float(x.float().abs().nan_to_num(0, 0, 0).max()),
                            kw={k: (tuple(v.shape) if torch.is_tensor(v) else v) for k, v in kw.items()}))
    return y


K.conv2d_nhwc = checked
for s in range(steps):
    d, g = gan.train_step(it, B)
    print('step', s + 1, 'tile-14 launches so far', seen, 'bad', len(bad), 'losses', float(d.divergence), float(g.divergence), flush=True)
for b_ in bad[:8]:
    print(b_)
