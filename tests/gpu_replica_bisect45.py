"""Which op makes the 8 copies of the same 2 samples inside a batch of 16 differ at configs 4 / 5? (test_config45_parity.py measures a
replica spread of up to 0.0625-0.094 in the generated images.) Wraps every HipOps method and reports the replica spread of the inputs
and outputs of each call of one no-grad generator forward. Test infrastructure.   python tests/gpu_replica_bisect45.py c4|c5"""
import sys
import tempfile
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
import c2_common as c2   # noqa: E402
import c45_common as cc   # noqa: E402
from gigagan_pytorch_amd import ops   # noqa: E402

BATCH = 16


def spread(t):
    if not torch.is_tensor(t) or t.dim() == 0 or t.shape[0] % BATCH or not t.is_floating_point():
        return None
    r = t.reshape(t.shape[0] // BATCH, BATCH // cc.BASE_BATCH, cc.BASE_BATCH, -1).float()
    return float((r - r[:, :1]).abs().max())


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else 'c4'
    dev = torch.device('cuda', 0)
    G, D = cc.build_models(cfg)
    with tempfile.TemporaryDirectory() as tmp:
        gan = cc.make_trainer(cfg, G, D, dev, tmp)
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
    gan.G.train()
    text = dict(text_encodings=cc.text_encodings(BATCH).to(dev)) if cfg == 'c4' else {}
    real = cc.real_images(BATCH).to(dev)
    with ops.use_impl(ops.HipOps()), c2.randn_replay(), torch.no_grad():
        if cfg == 'c5':
            lowres = ops.impl.resize_nearest(real, (64, 64))
            img, rgbs = gan.G(lowres_image=lowres, noise=cc.latents(BATCH).to(dev), return_all_rgbs=True)
        else:
            img, rgbs = gan.G(noise=cc.latents(BATCH).to(dev), return_all_rgbs=True, **text)
    print(f'== {cfg} generator forward (no grad), batch {BATCH}')
    for n, shp, i, o in log:
        first = max(i, default=0) == 0 and max(o, default=0) > 0
        if first:
            print(f'{n:22s} {str(shp):70s} in {max(i, default=0):.3e} out {max(o, default=0):.3e}  <-- inputs identical across replicas, outputs not')
    nz = [(k, n, shp, max(i, default=0), max(o, default=0)) for k, (n, shp, i, o) in enumerate(log) if max(i, default=0) > 0 or max(o, default=0) > 0]
    print('first calls with any replica spread (call index, op, shapes, in, out):')
    for k, n, shp, i, o in nz[:8]:
        print(f'   #{k:4d} {n:22s} {str(shp):70s} in {i:.3e} out {o:.3e}')
    grew = [(n, shp, max(i, default=0), max(o, default=0)) for n, shp, i, o in log if max(o, default=0) > 4 * max(max(i, default=0), 1e-9)]
    print('calls whose output spread exceeds 4x their input spread:', len(grew))
    for n, shp, i, o in grew[:12]:
        print(f'   {n:22s} {str(shp):70s} in {i:.3e} out {o:.3e}')
    print('image spread', spread(img))


if __name__ == '__main__':
    main()
