"""config 4 (text-conditional): ONE discriminator step with the gradient penalty, eager vs hipGraph replay, from the same weights, batch and
draws. Prints the losses and, per parameter, where the flat gradients differ.   python tests/gpu_text_graph_vs_eager.py [gp|plain]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import bench
from test_config2_parity import _DeviceRandnReplay

gp = (sys.argv[1] if len(sys.argv) > 1 else 'gp') == 'gp'
dev = torch.device('cuda', 0)
gan = bench.build_gan(256, dev, use_hip_graphs=True, workload='text')
for m in gan.D.modules():
    if isinstance(m, torch.nn.Dropout):
        m.p = 0.
it = iter(bench.SyntheticTextImages(16, 256, dev, seed=0))
snap = gan.state_snapshot()
names = {id(p): n for n, p in gan.D.named_parameters()}
res = {}
with _DeviceRandnReplay(dev) as rr:
    inner = gan._d_micro

    def d_micro(*a, **k):
        rr.reset()
        return inner(*a, **k)
    gan._d_micro = d_micro
    for tag, graphs in (('eager', False), ('eager2', False), ('replay1', True), ('replay2', True), ('replay3', True)):
        gan.state_restore(snap)
        gan.use_hip_graphs = graphs
        out = gan.train_discriminator_step(dl_iter=it, apply_gradient_penalty=gp)
        torch.cuda.synchronize()
        res[tag] = (gan.D_opt.flat_g.clone(), [float(v) for v in out if v is not None])
        print(tag, ['%.6g' % v for v in res[tag][1]], 'nonfinite grads', int((~torch.isfinite(res[tag][0])).sum()), flush=True)

ref = res['eager'][0]
for tag in ('eager2', 'replay1', 'replay2', 'replay3'):
    g = res[tag][0]
    print(f'== {tag} vs eager: rel l2', float((g - ref).norm() / ref.norm()))
    rows = []
    for p, o in zip(gan.D_opt._all, gan.D_opt.offsets):
        a, b = ref[o:o + p.numel()], g[o:o + p.numel()]
        d = float((a - b).norm() / (a.norm() + 1e-30))
        rows.append((names.get(id(p), '?'), d, float(a.norm()), float(b.norm())))
    bad = [r for r in rows if not (r[1] < 1e-3)]
    print('   parameters with rel diff >= 1e-3:', len(bad), 'of', len(rows))
    for r in bad[:60]:
        print('     %-70s rel %.3g  |eager| %.4g  |this| %.4g' % r)
