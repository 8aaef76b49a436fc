"""phase-by-phase run of the smoke configuration with synchronisation after every phase (fault localisation)."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from gigagan_pytorch_amd import GigaGAN   # noqa: E402
from gigagan_pytorch_amd.data import SyntheticImages   # noqa: E402
from gigagan_pytorch_amd.gigagan import cycle   # noqa: E402

graphs = sys.argv[1] == 'graphs'
if 'poison' in sys.argv:      # torch.empty() returns NaN-filled memory: reads of never-written elements surface
    torch.use_deterministic_algorithms(True, warn_only=True)
    torch.utils.deterministic.fill_uninitialized_memory = True
dev = torch.device('cuda', 0)
torch.manual_seed(0)
S = 64
def say(*a):
    torch.cuda.synchronize(); print(*a, flush=True)
gan = GigaGAN(generator=dict(image_size=S, dim_capacity=8, style_network=dict(dim=64, depth=4), unconditional=True, num_skip_layers_excite=4),
              discriminator=dict(image_size=S, dim_capacity=8, unconditional=True, num_skip_layers_excite=4),
              apply_gradient_penalty_every=(1000 if 'nogp' in sys.argv else 2), device=dev, model_folder='/tmp/gg-b-m', results_folder='/tmp/gg-b-r', use_hip_graphs=graphs)
say('built', graphs)
gan.G._debug_taps = {}
it = cycle(SyntheticImages(2, S, device=dev))
for i in range(3):
    gp = (gan._steps_host % 2 == (1 if 'gpfirst' in sys.argv else 0)) and 'nogp' not in sys.argv
    if 'eagergp' in sys.argv and gp:
        gan.use_hip_graphs = False
    if 'eagerg' in sys.argv:
        gan.use_hip_graphs = True
    d = gan.train_discriminator_step(dl_iter=it, apply_gradient_penalty=gp)
    gan.use_hip_graphs = graphs
    if 'eagerg' in sys.argv:
        gan.use_hip_graphs = False
    say('D step', i, 'gp', gp, float(d.divergence), 'gp_loss', float(d.gradient_penalty), 'D finite', bool(torch.isfinite(gan.D_opt.flat_p).all()),
        'gradmax', float(gan.D_opt.flat_g.abs().max()), 'pmax', float(gan.D_opt.flat_p.abs().max()))
    g = gan.train_generator_step(dl_iter=it, batch_size=2)
    say('G step', i, float(g.divergence), 'G finite', bool(torch.isfinite(gan.G_opt.flat_p).all()), 'gradmax', float(gan.G_opt.flat_g.abs().max()),
        'pmax', float(gan.G_opt.flat_p.abs().max()))
    gan._steps_host += 1
    bad = [(n, int((~torch.isfinite(p.grad)).sum()), p.numel()) for n, p in gan.G.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    good = [n for n, p in gan.G.named_parameters() if p.grad is not None and torch.isfinite(p.grad).all()]
    top = sorted(((float(p.grad.abs().max()), n) for n, p in gan.G.named_parameters() if p.grad is not None), reverse=True)
    if top and (top[0][0] > 1e6 or top[0][0] != top[0][0]):
        say('  top grads:', [(f'{v:.3g}', n) for v, n in top[:20]])
    T = gan.G._debug_taps
    if T:
        say('  taps:', {k: (tuple(v.shape), f'{float(v.float().abs().max()):.3g}', bool(torch.isfinite(v).all())) for k, v in T.items()})
        gy, xx = T['gy'].float(), T['x'].float()
        say('  recomputed d_excite max', float((gy * xx).sum(dim=(2, 3)).abs().max()), 'recorded ge max', float(T['ge'].float().abs().max()))
    if bad:
        say('  NaN grads in', len(bad), 'params; finite in', len(good)); say('  bad:', bad[:12]); say('  good:', good[:12])
z = torch.randn(2, 64, device=dev)
gan.G.eval()
with torch.no_grad():
    img = gan.G(noise=z)
say('eager G forward', float(img.float().abs().mean()))
sd = {k: v.detach().cpu() for k, v in gan.G.state_dict().items()}
say('state dict ok', len(sd))
