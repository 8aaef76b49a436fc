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
dev = torch.device('cuda', 0)
torch.manual_seed(0)
S = 64
def say(*a):
    torch.cuda.synchronize(); print(*a, flush=True)
gan = GigaGAN(generator=dict(image_size=S, dim_capacity=8, style_network=dict(dim=64, depth=4), unconditional=True, num_skip_layers_excite=4),
              discriminator=dict(image_size=S, dim_capacity=8, unconditional=True, num_skip_layers_excite=4),
              apply_gradient_penalty_every=2, device=dev, model_folder='/tmp/gg-b-m', results_folder='/tmp/gg-b-r', use_hip_graphs=graphs)
say('built', graphs)
it = cycle(SyntheticImages(2, S, device=dev))
for i in range(2):
    gp = (gan._steps_host % 2 == 0)
    d = gan.train_discriminator_step(dl_iter=it, apply_gradient_penalty=gp)
    say('D step', i, 'gp', gp, float(d.divergence))
    g = gan.train_generator_step(dl_iter=it, batch_size=2)
    say('G step', i, float(g.divergence))
    gan._steps_host += 1
z = torch.randn(2, 64, device=dev)
gan.G.eval()
with torch.no_grad():
    img = gan.G(noise=z)
say('eager G forward', float(img.float().abs().mean()))
sd = {k: v.detach().cpu() for k, v in gan.G.state_dict().items()}
say('state dict ok', len(sd))
