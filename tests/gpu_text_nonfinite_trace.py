"""text-conditional trainer (config 4), the bench's restore-every-cycle loop: after every step, how many non-finite elements do the
flat gradient / parameter buffers hold, and in which parameters do they first appear?  (python tests/gpu_text_nonfinite_trace.py [graphs|eager] [steps])"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

mode = sys.argv[1] if len(sys.argv) > 1 else 'graphs'        # graphs | eager | poison (eager, every torch.empty* float buffer pre-filled with NaN)
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 48
dev = torch.device('cuda', 0)
gan = bench.build_gan(256, dev, use_hip_graphs=None if mode == 'graphs' else False, workload='text')
torch.manual_seed(1)
it = iter(bench.SyntheticTextImages(16, 256, dev, seed=0))
snap = gan.state_snapshot()

if mode == 'poison':
    _empty, _empty_like, _new_empty, _empty_strided = torch.empty, torch.empty_like, torch.Tensor.new_empty, torch.empty_strided

    def _fill(t):
        if t.is_floating_point() and t.device.type == 'cuda':
            t.fill_(float('nan'))
        return t
    torch.empty = lambda *a, **k: _fill(_empty(*a, **k))
    torch.empty_like = lambda *a, **k: _fill(_empty_like(*a, **k))
    torch.empty_strided = lambda *a, **k: _fill(_empty_strided(*a, **k))
    torch.Tensor.new_empty = lambda self, *a, **k: _fill(_new_empty(self, *a, **k))


def names(opt, model, flat):
    bad = ~torch.isfinite(flat)
    out = []
    byid = {id(p): n for n, p in model.named_parameters()}
    for p, o in zip(opt._all, opt.offsets):
        c = int(bad[o:o + p.numel()].sum())
        if c:
            out.append((byid.get(id(p), '?'), c, p.numel()))
    return out


first = None
for s in range(steps):
    if (gan._steps_host - 1) % 4 == 0:
        gan.state_restore(snap)
    d_l, g_l = gan.train_step(it, 16)
    torch.cuda.synchronize()
    nf = {k: int((~torch.isfinite(v)).sum()) for k, v in (('g_p', gan.G_opt.flat_p), ('d_p', gan.D_opt.flat_p), ('g_g', gan.G_opt.flat_g),
                                                          ('d_g', gan.D_opt.flat_g))}
    ls = [float(v) for v in (*d_l, *g_l) if v is not None]
    print(s, 'host step', gan._steps_host - 1, nf, ['%.3g' % v for v in ls], flush=True)
    if any(nf.values()) and first is None:
        first = s
        for tag, opt, model, flat in (('G grads', gan.G_opt, gan.G, gan.G_opt.flat_g), ('D grads', gan.D_opt, gan.D, gan.D_opt.flat_g)):
            lst = names(opt, model, flat)
            print('   ', tag, len(lst), 'parameters hold non-finite values; first 12:', lst[:12], flush=True)
print('first non-finite step:', first)
