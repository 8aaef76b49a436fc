"""Checkpoint interchange with the reference trainer (gp.py:2033-2108 save/load, :2173-2185 EMA): a
package written by the unmodified reference loads into our trainer and one written by ours loads into the reference, with
identical generate() outputs either way. Live only (needs /root/reference; the package layout itself is also asserted in
test_trainer_cpu.py without the reference)."""
import torch

from gigagan_pytorch_amd import GigaGAN, ops
from gigagan_pytorch_amd.data import SyntheticImages
from gigagan_pytorch_amd.gigagan import cycle
from oracle.torch_ops import OracleOps
from helpers import rel_err, TINY_G, TINY_D


def test_checkpoints_round_trip_with_the_reference_trainer(reference, tmp_path):
    torch.manual_seed(0)
    ref_gan = reference.GigaGAN(generator=dict(TINY_G), discriminator=dict(TINY_D), model_folder=str(tmp_path / 'rm'),
                                results_folder=str(tmp_path / 'rr'))
    ref_ckpt, our_ckpt = tmp_path / 'ref.ckpt', tmp_path / 'ours.ckpt'
    ref_gan.save(str(ref_ckpt))
    pkg = torch.load(ref_ckpt, weights_only=False)
    assert {'G', 'D', 'G_opt', 'D_opt', 'steps', 'version', 'G_ema'} <= set(pkg)

    gan = GigaGAN(generator=dict(TINY_G), discriminator=dict(TINY_D), device='cpu', model_folder=str(tmp_path / 'm'),
                  results_folder=str(tmp_path / 'r'))
    assert list(gan.G.state_dict().keys()) == list(pkg['G'].keys())
    assert list(gan.D.state_dict().keys()) == list(pkg['D'].keys())
    assert set(gan.G_ema.state_dict().keys()) == set(pkg['G_ema'].keys())
    gan.load(ref_ckpt)
    z = torch.randn(2, 32)

    def both():
        with ops.use_impl(OracleOps()):
            torch.manual_seed(1)
            ours = gan.generate(noise=z)             # the EMA generator (gp.py:2165-2169)
        torch.manual_seed(1)
        theirs = ref_gan.generate(noise=z)
        return ours, theirs
    a, b = both()
    assert rel_err(a, b) < 1e-5

    # one optimisation step on our side (weights, EMA bookkeeping and step counter move), then hand the package back
    with ops.use_impl(OracleOps()):
        gan.train_step(cycle(SyntheticImages(2, 16)), 2)
    gan.save(our_ckpt)
    ref_gan.load(str(our_ckpt))
    assert int(ref_gan.steps.item()) == gan._steps_host == 2
    a, b = both()
    assert rel_err(a, b) < 1e-5
    with ops.use_impl(OracleOps()):
        torch.manual_seed(1)
        a = gan.G(noise=z)
    torch.manual_seed(1)
    b = ref_gan.unwrapped_G(noise=z)
    assert rel_err(a, b) < 1e-5                                 # the live (stepped) generator, too
    assert not torch.equal(gan.G.state_dict()['init_block'], pkg['G']['init_block'])
