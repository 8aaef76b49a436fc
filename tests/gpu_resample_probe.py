"""gg_resample: per-pixel kernels (GG_RESAMPLE_2X2=0) against the 2 x 2-blocked kernel on the generator's up-sampling shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gigagan_pytorch_amd import kernels as K
dev = torch.device('cuda', 0)
torch.manual_seed(0)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for kind, b, H, C in (('upblur', 32, 128, 32), ('upblur', 32, 64, 64), ('upblur', 32, 32, 128), ('upblur', 32, 16, 256), ('upblur', 32, 128, 8),
                      ('blur', 32, 128, 64), ('bilinear_up', 32, 64, 64), ('upblur', 32, 128, 3), ('upblur', 32, 64, 3), ('bilinear_down', 32, 256, 3), ('upblur_T', 32, 128, 3)):
    spec = (K.ResampleSpec.upsample_blur(H, H) if kind == 'upblur' else K.ResampleSpec.blur(H, H) if kind == 'blur'
            else K.ResampleSpec.bilinear(H, H, H // 2, H // 2) if kind == 'bilinear_down'
            else K.ResampleSpec.upsample_blur(H, H).transposed() if kind == 'upblur_T' else K.ResampleSpec.bilinear(H, H, 2 * H, 2 * H))
    if kind == 'upblur_T':
        H = 2 * H
    x = torch.randn(b, spec.ih, spec.iw, C, device=dev).bfloat16()
    res = {}
    for env in ('0', '1'):
        os.environ['GG_RESAMPLE_2X2'] = env
        y = K.resample_nhwc(x, spec)
        res[env] = (y, timeit(lambda: K.resample_nhwc(x, spec)))
    nbytes = (x.numel() + res['1'][0].numel()) * 2
    print('%-12s in (%d, %d, %d, %d) taps %d: GG_RESAMPLE_2X2=0 %6.1f us %5.2f TB/s | default %6.1f us %5.2f TB/s | max diff %.1e' % (
        kind, b, H, H, C, spec.ty, res['0'][1], nbytes / res['0'][1] / 1e6, res['1'][1], nbytes / res['1'][1] / 1e6,
        float((res['0'][0].float() - res['1'][0].float()).abs().max())), flush=True)
