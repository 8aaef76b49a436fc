"""loss trajectory of the config-2 trainer, eager vs hipGraph replay (test infrastructure)."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench   # noqa: E402
from gigagan_pytorch_amd.data import SyntheticImages   # noqa: E402
from gigagan_pytorch_amd.gigagan import cycle   # noqa: E402

dev = torch.device('cuda', 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
for graphs in (False, True):
    torch.manual_seed(0)
    gan = bench.build_gan(256, dev, use_hip_graphs=graphs)
    it = cycle(SyntheticImages(32, 256, device=dev, seed=0))
    for s in range(n):
        d, g = gan.train_step(it, 32)
        pn = float(gan.D_opt.flat_p.norm()), float(gan.G_opt.flat_p.norm())
        print(f'graphs={graphs} step {s+1}: D {float(d.divergence):.4g} MSD {float(d.multiscale_divergence):.4g} GP {float(d.gradient_penalty):.4g} '
              f'SSL {float(d.aux_reconstruction):.4g} | G {float(g.divergence):.4g} MSG {float(g.multiscale_divergence):.4g} | |D| {pn[0]:.6g} |G| {pn[1]:.6g}', flush=True)
    del gan
    torch.cuda.empty_cache()
