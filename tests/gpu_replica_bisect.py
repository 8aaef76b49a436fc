"""Which op makes the 16 copies of the same 2 samples inside a batch of 32 differ? (tests/test_config2_parity.py found images that
are not bit-identical across replicas.) Wraps every HipOps method and reports the replica spread of inputs and outputs of each call
of one no-grad generator forward and one discriminator forward at config 2. Test infrastructure.
    python tests/gpu_replica_bisect.py"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
import c2_common as c2   # noqa: E402
from gigagan_pytorch_amd import ops   # noqa: E402

BATCH = 32


def spread(t):
    if not torch.is_tensor(t) or t.dim() == 0 or t.shape[0] % BATCH or not t.is_floating_point():
        return None
    r = t.reshape(t.shape[0] // BATCH, BATCH // c2.BASE_BATCH, c2.BASE_BATCH, -1).float()
    return float((r - r[:, :1]).abs().max())


def main():
    dev = torch.device('cuda', 0)
    G, D = c2.build_models()
    G, D = G.to(dev), D.to(dev)
    H = ops.HipOps
    names = [n for n in dir(H) if not n.startswith('_') and callable(getattr(H, n)) and n not in ('prepare',)]
    log = []
    for n in names:
        orig = getattr(H, n)

        def wrap(self, *a, _orig=orig, _n=n, **k):
            ins = [spread(t) for t in list(a) + list(k.values())]
            out = _orig(self, *a, **k)
            outs = [spread(t) for t in (out if isinstance(out, (tuple, list)) else [out])]
            shp = [tuple(t.shape) for t in a if torch.is_tensor(t)][:2]
            log.append((_n, shp, [i for i in ins if i is not None], [o for o in outs if o is not None]))
            return out
        setattr(H, n, wrap)
    with c2.randn_replay(), torch.no_grad():
        img, rgbs = G(noise=c2.latents(BATCH).to(dev), return_all_rgbs=True)
    print('== generator forward (no grad)')
    for n, shp, i, o in log:
        flag = '  <-- first divergence' if (max(i, default=0) == 0 and max(o, default=0) > 0) else ''
        if max(o, default=0) > 0 or flag:
            print(f'{n:18s} {str(shp):60s} in {max(i, default=0):.3e} out {max(o, default=0):.3e}{flag}')
    print('image spread', spread(img))
    log.clear()
    real = c2.real_images(BATCH).to(dev)
    with torch.no_grad():
        logits, ms, _ = D(real, D.real_images_to_rgbs(real), calc_aux_loss=False)
    print('== discriminator forward (no grad)')
    for n, shp, i, o in log:
        flag = '  <-- first divergence' if (max(i, default=0) == 0 and max(o, default=0) > 0) else ''
        if max(o, default=0) > 0 or flag:
            print(f'{n:18s} {str(shp):60s} in {max(i, default=0):.3e} out {max(o, default=0):.3e}{flag}')
    lg = logits.float().view(16, BATCH // 2, 2)
    print('logit spread', float((lg - lg[:, :1]).abs().max()))


if __name__ == '__main__':
    main()
