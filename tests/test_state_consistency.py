"""Regression tests for state that lives outside the parameters (CPU, emulator build of the C ABI): the EMA schedule resumes
from a checkpoint, packed bf16 operands follow in-place parameter edits, parameters without a gradient in a step are skipped
by the fused optimizer, data-parallel ranks read disjoint shards."""
import torch
from torch.utils.data import DataLoader, TensorDataset

from gigagan_pytorch_amd import GigaGAN, ops
from gigagan_pytorch_amd.data import shard_dataloader, DevicePrefetcher
from gigagan_pytorch_amd.ema import EMA
from gigagan_pytorch_amd.gigagan import cycle
from gigagan_pytorch_amd.modules import Conv2d
from gigagan_pytorch_amd.optimizer import FlatAdamW
from helpers import rel_err, TINY_G, TINY_D


def test_ema_schedule_survives_a_checkpoint_and_warms_up(tmp_path):
    """after load() the first EMA updates must NOT overwrite the loaded average with the online weights;
    the decay follows ema_pytorch's warm-up 1 - (1 + step - update_after_step) ** (-2/3), capped at beta."""
    lin = torch.nn.Linear(4, 4)
    ema = EMA(lin, beta=0.995, update_after_step=3, update_every=2)
    for _ in range(12):
        with torch.no_grad():
            lin.weight.add_(1.0)
        ema.update()
    assert ema._step_host == 12 and bool(ema.initted)
    assert abs(ema.get_current_decay(10) - (1 - (1 + 6) ** (-2 / 3))) < 1e-6 and ema.get_current_decay(4) == 0.
    assert ema.get_current_decay(10 ** 6) == 0.995
    sd = {k: v.clone() for k, v in ema.state_dict().items()}
    assert any(k.startswith('online_model.') for k in sd) and 'ema_model.weight' in sd     # ema_pytorch's key layout
    lin2 = torch.nn.Linear(4, 4)
    ema2 = EMA(lin2, beta=0.995, update_after_step=3, update_every=2)
    ema2.load_state_dict(sd)
    assert ema2._step_host == 12 and ema2._initted_host
    with torch.no_grad():
        lin2.weight.fill_(100.)
    before = ema2.ema_model.weight.clone()
    ema2.update()                       # step 12: an averaged update, not a copy
    after = ema2.ema_model.weight
    assert not torch.equal(after, lin2.weight) and not torch.equal(after, before)
    d = ema2.get_current_decay(13)
    assert torch.allclose(after, before * d + 100. * (1 - d), atol=1e-4)


def test_trainer_load_resumes_ema_counters(tmp_path):
    torch.manual_seed(0)
    kw = dict(generator=dict(TINY_G), discriminator=dict(TINY_D), device='cpu', model_folder=str(tmp_path / 'm'),
              results_folder=str(tmp_path / 'r'))
    gan = GigaGAN(**kw)
    gan.G_ema._step_host = 205
    gan.G_ema.step.fill_(205)
    gan.G_ema._initted_host = True
    gan.G_ema.initted.fill_(True)
    with torch.no_grad():
        for p in gan.G_ema.ema_model.parameters():
            p.add_(0.5)                 # the average differs from the online weights
    gan.save(str(tmp_path / 'ck.pt'))
    gan2 = GigaGAN(**kw)
    gan2.load(str(tmp_path / 'ck.pt'))
    assert gan2.G_ema._step_host == 205 and gan2.G_ema._initted_host
    w_ema = next(gan2.G_ema.ema_model.parameters()).clone()
    w_on = next(gan2.G.parameters())
    gan2.G_ema.update()                 # 205 % 10 != 0: nothing may change; at 210 a lerp, never a copy
    assert torch.equal(next(gan2.G_ema.ema_model.parameters()), w_ema)
    for _ in range(5):
        gan2.G_ema.update()
    w2 = next(gan2.G_ema.ema_model.parameters())
    assert not torch.equal(w2, w_on) and not torch.equal(w2, w_ema)


def test_packed_operands_follow_load_state_dict():
    """nn.Module.load_state_dict / in-place edits bump the parameter's version; the persistent bf16
    operands of the pack table (and the per-parameter cache) must be re-packed before the next use."""
    torch.manual_seed(0)
    conv = Conv2d(8, 16, 3, padding=1)
    opt = FlatAdamW(list(conv.parameters()), lr=1e-2)
    x = torch.randn(2, 8, 8, 8)
    with ops.use_impl(ops.HipOps()), torch.no_grad():
        y0 = conv(x).float()
        sd = {k: v * 2.0 for k, v in conv.state_dict().items()}
        conv.load_state_dict(sd)
        y1 = conv(x).float()
        assert rel_err(y1, 2 * y0) < 1e-2, 'stale packed weights after load_state_dict'
        conv.weight.mul_(0.5)
        conv.bias.mul_(0.5)
        assert rel_err(conv(x).float(), y0) < 1e-2
    free = Conv2d(8, 16, 3, padding=1)          # not owned by an optimizer: the per-parameter cache
    with ops.use_impl(ops.HipOps()), torch.no_grad():
        z0 = free(x).float()
        free.weight.mul_(3.0)
        free.bias.mul_(3.0)
        assert rel_err(free(x).float(), 3 * z0) < 1e-2
    del opt


def test_parameters_without_gradient_are_skipped_for_the_step():
    """reference semantics: a parameter whose .grad is None is neither decayed nor moved by its moments (torch AdamW)."""
    torch.manual_seed(0)
    a, b = torch.nn.Parameter(torch.randn(300, 3)), torch.nn.Parameter(torch.randn(40, 5))
    opt = FlatAdamW([a, b], lr=1e-1)
    opt.flat_g.fill_(0.5)
    opt.step()
    b1 = b.detach().clone()
    a1 = a.detach().clone()
    opt.zero_grad()
    a.grad.fill_(0.25)
    opt.step(skip=[b])
    assert torch.equal(b.detach(), b1) and not torch.equal(a.detach(), a1)
    opt.step()                                  # not skipped: decay + momentum move it
    assert not torch.equal(b.detach(), b1)


def test_shard_dataloader_gives_disjoint_shards_and_prefetcher_passes_batches_through():
    ds = TensorDataset(torch.arange(64).float().view(64, 1))
    dl = DataLoader(ds, batch_size=4, shuffle=True, drop_last=True)
    seen = []
    for r in range(2):
        sh = shard_dataloader(dl, r, 2)
        assert sh.batch_size == 4
        seen.append(torch.cat([b[0].flatten() for b in sh]))
    assert len(seen[0]) == 32 and len(seen[1]) == 32
    assert set(seen[0].tolist()).isdisjoint(seen[1].tolist())
    assert shard_dataloader(dl, 0, 1) is dl
    pf = DevicePrefetcher(DataLoader(ds, batch_size=8), 'cpu')
    got = torch.cat([b[0].flatten() for b in pf])
    assert torch.equal(got, torch.arange(64).float()) and pf.batch_size == 8


def test_shard_dataloader_keeps_custom_sampling_schemes_and_refuses_what_it_cannot_shard():
    """accelerator.prepare semantics (gp.py:2161) beyond the stock samplers: custom samplers / batch samplers are dealt out
    round-robin by batch, loaders without automatic batching by index, and an unshardable stream raises instead of silently
    handing every rank the same data (VERDICT r4 missing 5)."""
    import pytest
    from torch.utils.data import BatchSampler, Sampler, WeightedRandomSampler

    ds = TensorDataset(torch.arange(40).float().view(40, 1))

    class Evens(Sampler):                     # a custom sampling scheme: even indices first, then odd ones
        def __iter__(self):
            return iter(list(range(0, 40, 2)) + list(range(1, 40, 2)))

        def __len__(self):
            return 40

    dl = DataLoader(ds, batch_size=4, sampler=Evens())
    ref = [b[0].flatten().tolist() for b in dl]                      # the batch stream the scheme defines
    got = [[b[0].flatten().tolist() for b in shard_dataloader(dl, r, 2)] for r in range(2)]
    assert got[0] == ref[0::2] and got[1] == ref[1::2]
    # an explicit batch_sampler (batch_size is None on such a loader), 7 batches on 2 ranks: the odd one out is completed from the
    # epoch's first batch, so both ranks run the same number of steps
    dl = DataLoader(ds, batch_sampler=BatchSampler(Evens(), batch_size=6, drop_last=False))
    ref = [b[0].flatten().tolist() for b in dl]
    assert len(ref) == 7
    got = [[b[0].flatten().tolist() for b in shard_dataloader(dl, r, 2)] for r in range(2)]
    assert got[0] == ref[0::2] and got[1] == ref[1::2] + [ref[0]]
    assert len(shard_dataloader(dl, 0, 2)) == 4
    # a seeded WeightedRandomSampler: same generator seed on every rank -> the ranks deal out ONE stream
    def weighted():
        return DataLoader(ds, batch_size=5, sampler=WeightedRandomSampler(torch.ones(40), 40, replacement=False,
                                                                          generator=torch.Generator().manual_seed(3)))
    ref = [b[0].flatten().tolist() for b in weighted()]
    got = [[b[0].flatten().tolist() for b in shard_dataloader(weighted(), r, 2)] for r in range(2)]
    assert got[0] == ref[0::2] and got[1] == ref[1::2]
    assert set(sum(got[0], [])).isdisjoint(sum(got[1], []))
    # an UNSEEDED random sampler (generator=None draws from each rank's global RNG: the shards would overlap): every rank gets the same
    # seeded generator installed, so the ranks still deal out one stream
    def unseeded(r):
        torch.manual_seed(100 + r)          # the ranks' global generators differ
        return DataLoader(ds, batch_size=5, sampler=WeightedRandomSampler(torch.ones(40), 40, replacement=False))
    got = [[b[0].flatten().tolist() for b in shard_dataloader(unseeded(r), r, 2, seed=7)] for r in range(2)]
    assert set(sum(got[0], [])).isdisjoint(sum(got[1], [])) and len(sum(got[0], [])) == len(sum(got[1], [])) == 20
    got = [[float(b[0][0]) for b in shard_dataloader(DataLoader(ds, batch_size=None, shuffle=True), r, 2)] for r in range(2)]
    assert set(got[0]).isdisjoint(got[1]) and len(got[0]) == len(got[1]) == 20
    # automatic batching off: the dataset yields ready batches, the sampler's indices are dealt out
    class Ready(torch.utils.data.Dataset):
        def __len__(self):
            return 6

        def __getitem__(self, i):
            return torch.full((2, 1), float(i))
    got = [[float(b[0, 0]) for b in shard_dataloader(DataLoader(Ready(), batch_size=None), r, 2)] for r in range(2)]
    assert got == [[0., 2., 4.], [1., 3., 5.]]
    # a plain iterable cannot be re-dealt: refused unless the caller declares it per-rank
    stream = [torch.zeros(2, 1)] * 3
    with pytest.raises(ValueError, match='is_rank_sharded'):
        shard_dataloader(stream, 0, 2)

    class PerRank(list):
        is_rank_sharded = True
    assert shard_dataloader(PerRank(stream), 0, 2) is not None
    assert shard_dataloader(stream, 0, 1) is stream


def test_ragged_bias_reads_its_zero_padded_flat_slot():
    """a bias of O % 8 != 0 channels (rgb / logit convolutions): for a FlatAdamW-owned parameter the o8 floats the kernels read are a
    longer VIEW of its 256-float slot in the flat buffer (no pad launch per forward) whose tail stays zero across optimizer steps and
    follows load_state_dict; any other tensor is padded by a copy. The conv result equals the explicitly padded one either way."""
    torch.manual_seed(0)
    conv = Conv2d(8, 3, 3, padding=1)
    b0 = ops._bias8(conv.bias, 8)
    assert b0.shape == (8,) and b0.data_ptr() != conv.bias.data_ptr() and torch.equal(b0[:3], conv.bias.detach()) and not b0[3:].any()
    opt = FlatAdamW(list(conv.parameters()), lr=1e-2)
    b1 = ops._bias8(conv.bias, 8)
    assert b1.data_ptr() == conv.bias.data_ptr() and torch.equal(b1[:3], conv.bias.detach()) and not b1[3:].any()
    x = torch.randn(2, 8, 6, 6)
    for _ in range(3):
        opt.zero_grad()
        y = conv(x)
        assert y.shape == (2, 3, 6, 6)
        with ops.sinking():
            y.float().square().mean().backward()
        opt.step()
        b2 = ops._bias8(conv.bias, 8)
        assert b2.data_ptr() == conv.bias.data_ptr() and torch.equal(b2[:3], conv.bias.detach()) and not b2[3:].any()
    conv.load_state_dict({'weight': conv.weight.detach().clone(), 'bias': torch.tensor([1., 2., 3.])})
    assert torch.equal(ops._bias8(conv.bias, 8), torch.tensor([1., 2., 3., 0., 0., 0., 0., 0.]))
    # the invariant behind the view (optimizer.FlatAdamW._build): slot tails are zero; a write that bypasses the optimizer and dirties
    # them (a whole-buffer copy from elsewhere) is repaired by zero_slot_tails, which snapshot restores and state-dict loads call
    assert opt.slot_tails_are_zero()
    opt.flat_p.add_(1.)
    assert not opt.slot_tails_are_zero()
    opt.zero_slot_tails()
    assert opt.slot_tails_are_zero() and not ops._bias8(conv.bias, 8)[3:].any() and torch.equal(conv.bias.detach(), torch.tensor([2., 3., 4.]))
