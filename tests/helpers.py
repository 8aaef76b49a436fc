import torch


def rel_err(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm().clamp(min=1e-12)).item()


def bf(t):
    return t.to(torch.bfloat16)


SMALL_G = dict(image_size=32, dim_capacity=8, dim_max=32, dim_latent=32, style_network=dict(dim=32, depth=2),
               unconditional=True, num_skip_layers_excite=2, self_attn_heads=2, self_attn_dim_head=16)
SMALL_D = dict(image_size=32, dim_capacity=8, dim_max=32, unconditional=True, num_skip_layers_excite=2, attn_heads=2,
               attn_dim_head=16)
C1_G = dict(image_size=64, dim_capacity=8, style_network=dict(dim=64, depth=4), unconditional=True, num_skip_layers_excite=4)
C1_D = dict(image_size=64, dim_capacity=8, unconditional=True, num_skip_layers_excite=4)
# trainer / multi-rank tests: the smallest model that still has every block type (attention at 8x8, one multi-scale
# input, predictor, aux decoder, squeeze-excite) — the host-side emulator pays ~1 s per shuffle-heavy launch
TINY_G = dict(image_size=16, dim_capacity=8, dim_max=32, dim_latent=32, style_network=dict(dim=32, depth=2),
              unconditional=True, num_skip_layers_excite=1, self_attn_resolutions=(8,), self_attn_heads=2,
              self_attn_dim_head=16)
TINY_D = dict(image_size=16, dim_capacity=8, dim_max=32, unconditional=True, num_skip_layers_excite=1,
              attn_resolutions=(8,), attn_heads=2, attn_dim_head=16, multiscale_input_resolutions=(8,))
