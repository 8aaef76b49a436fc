"""worker of tests/test_rccl_two_gpus.py: launched by torch.distributed.run with one process per GPU."""
import sys
from pathlib import Path

import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
from gigagan_pytorch_amd import GigaGAN, _C, distributed as gdist   # noqa: E402
from gigagan_pytorch_amd.data import SyntheticImages   # noqa: E402
from gigagan_pytorch_amd.gigagan import cycle   # noqa: E402
from helpers import C1_G, C1_D   # noqa: E402


def main():
    rank, local, world = gdist.init_from_env('cuda')
    dev = torch.device('cuda', local)
    torch.manual_seed(0)
    gan = GigaGAN(generator=dict(C1_G), discriminator=dict(C1_D), apply_gradient_penalty_every=2, device=dev,
                  model_folder=f'/tmp/gg-rccl-m{rank}', results_folder=f'/tmp/gg-rccl-r{rank}')
    assert gdist.native_comm() is not None and _C.lib().lib.gg_comm_world() == world
    torch.manual_seed(10 + rank)
    it = cycle(SyntheticImages(2, 64, device=dev, seed=rank))
    d0 = gan.D_opt.flat_p.clone()
    for _ in range(4):                      # plain and gradient-penalty steps, hipGraph capture on the way
        gan.train_step(it, 2)
    flat = torch.cat([gan.D_opt.flat_p, gan.G_opt.flat_p])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    same = all(torch.equal(gathered[0], g) for g in gathered)
    moved = not torch.equal(d0, gan.D_opt.flat_p)
    ok = bool(same and moved and torch.isfinite(flat).all())
    print(f'rank {rank}: replicas identical {same}, weights moved {moved}, graphs {gan.use_hip_graphs}', flush=True)
    dist.barrier()
    gdist.shutdown()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
