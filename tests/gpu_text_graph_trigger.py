"""which host-side action between two replays of the captured text-conditional D step changes its result?  [gp|plain]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import bench
from gigagan_pytorch_amd import ops
from test_config2_parity import _DeviceRandnReplay

gp = (sys.argv[1] if len(sys.argv) > 1 else 'gp') == 'gp'
dev = torch.device('cuda', 0)
gan = bench.build_gan(256, dev, use_hip_graphs=True, workload='text')
for m in gan.D.modules():
    if isinstance(m, torch.nn.Dropout):
        m.p = 0.
it = iter(bench.SyntheticTextImages(16, 256, dev, seed=0))
snap = gan.state_snapshot()
real_step = gan.D_opt.step
step_on = [False]
gan.D_opt.step = lambda *a, **k: real_step(*a, **k) if step_on[0] else None
ref = [None]


def run(tag, graphs):
    gan.use_hip_graphs = graphs
    out = gan.train_discriminator_step(dl_iter=it, apply_gradient_penalty=gp)
    torch.cuda.synchronize()
    g = gan.D_opt.flat_g.clone()
    if ref[0] is None:
        ref[0] = g
    print('%-44s' % tag, ['%.6g' % float(v) for v in out if v is not None], 'rel l2 vs eager %.3g' % float((g - ref[0]).norm() / ref[0].norm()), flush=True)


with _DeviceRandnReplay(dev) as rr:
    inner = gan._d_micro

    def d_micro(*a, **k):
        rr.reset()
        return inner(*a, **k)
    gan._d_micro = d_micro
    run('eager', False)
    run('replay 1 (capture)', True)
    run('replay 2 (nothing in between)', True)
    if os.environ.get('TRIG_SHORT'):
        g = gan.D_opt.flat_g
        names = {id(p): n for n, p in gan.D.named_parameters()}
        bad = []
        for p, o in zip(gan.D_opt._all, gan.D_opt.offsets):
            a, b = ref[0][o:o + p.numel()], g[o:o + p.numel()]
            d = float((a - b).norm() / (a.norm() + 1e-30))
            if not d < 1e-3:
                bad.append((names[id(p)], '%.3g' % d))
        print('   differing parameters:', len(bad), bad[:8], flush=True)
        sys.exit(0)
    ops.pack_cache_clear()
    run('replay 3 (after pack_cache_clear)', True)
    gan.G_opt.pack_table.refresh()
    run('replay 4 (after G pack_table.refresh)', True)
    gan.D_opt.pack_table.refresh()
    run('replay 5 (after D pack_table.refresh)', True)
    gan.state_restore(snap)
    run('replay 6 (after state_restore)', True)
    run('eager again', False)
    step_on[0] = True
    run('replay 7 (this one ends with D_opt.step)', True)
    step_on[0] = False
    gan.state_restore(snap)
    run('replay 8 (after that step + state_restore)', True)
    run('eager again', False)
