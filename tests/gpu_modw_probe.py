"""times the batched modulation launch of config 2's generator (15 layers, batch 32) with and without the cached Gram rows,
per path group. Test infrastructure: python tests/gpu_modw_probe.py"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
from gigagan_pytorch_amd import kernels as K, ops   # noqa: E402
from gpu_modconv_layers import LAYERS, graph_us, DEV, B, N   # noqa: E402


def build(with_gram, only=None):
    layers = []
    for I, O, R, excited in LAYERS:
        path = ops.HipOps._modconv_path(B, N, O, I, R, R)
        if only and path != only:
            continue
        w = torch.randn(N, O, I, 3, 3, device=DEV) * 0.1
        ly = dict(w=w, mod=torch.randn(B, I, device=DEV) * 0.3, kmod=torch.randn(B, N, device=DEV), demod=True, eps=1e-8, Ip=I, Op=O)
        if with_gram:
            wf = w.flatten(3)
            ly['gram'] = torch.stack([(wf[0] * wf[0]).sum(-1), 2 * (wf[0] * wf[1]).sum(-1), (wf[1] * wf[1]).sum(-1)]).contiguous()
        if path == 'pimg':
            ly.update(coef=False, wmix=torch.empty(B, O, 9 * I, dtype=torch.bfloat16, device=DEV), layout=1)
        elif path == 'sconv':
            ly.update(coef=False, wmix=torch.zeros(B, 9, I // 16, 32, 16, dtype=torch.bfloat16, device=DEV), layout=2)
        layers.append(ly)
    return layers


def main():
    torch.manual_seed(0)
    for only in (None, 'bank', 'pimg', 'sconv'):
        for with_gram in (False, True):
            layers = build(with_gram, only)
            us, _ = graph_us(lambda: K.modw_multi(layers))
            print(f'paths {only or "all":5s} layers {len(layers):2d} gram {int(with_gram)}: {us:7.1f} us', flush=True)
    # equality of the two variants
    la, lb = build(False, 'bank'), None
    torch.manual_seed(0)
    oa = K.modw_multi(la)
    for ly in la:
        wf = ly['w'].flatten(3)
        ly['gram'] = torch.stack([(wf[0] * wf[0]).sum(-1), 2 * (wf[0] * wf[1]).sum(-1), (wf[1] * wf[1]).sum(-1)]).contiguous()
    ob = K.modw_multi(la)
    torch.cuda.synchronize()
    print('d with / without cached Gram: max rel diff', max(float(((a['d'] - b['d']).abs() / b['d'].abs()).max()) for a, b in zip(oa, ob)))


if __name__ == '__main__':
    main()
