"""ORACLE — test infrastructure only (bench.py's cpu_baseline leg and tests). The optimizer update of the CPU restatement:
torch AdamW semantics (reference optimizer.py:10-34: AdamW, weight decay 1e-2 on parameters with ndim >= 2, parameters that
never receive a gradient skipped) written as plain tensor algebra over the flat buffers FlatAdamW lays out, so that OUR trainer
can be timed end to end on the host cores without the GPU extension."""
import math

import torch


def install_cpu_adamw(opt):
    """replace `opt.step` (the fused HIP launch) of a FlatAdamW whose buffers live on the CPU by the same update in torch."""
    @torch.no_grad()
    def step(grad_scale: float = 1.0, skip=()):
        # parameters in `skip` got no gradient this step: neither stepped nor decayed (torch optimizers skip None grads)
        flags = opt.flags_without(skip).repeat_interleave(256)
        active = (flags & 1).bool()
        decay = ((flags & 2) != 0) & active
        g0 = opt.param_groups[0]
        lr, (b1, b2), eps = g0['lr'], g0['betas'], g0['eps']
        opt.step_count += 1
        t = opt.step_count
        g = opt.flat_g * grad_scale
        m = torch.where(active, opt.flat_m * b1 + g * (1 - b1), opt.flat_m)
        v = torch.where(active, opt.flat_v * b2 + g * g * (1 - b2), opt.flat_v)
        p = torch.where(decay, opt.flat_p * (1 - lr * opt.wd), opt.flat_p)
        upd = (m / (1 - b1 ** t)) / (v.sqrt() / math.sqrt(1 - b2 ** t) + eps)
        opt.flat_p.copy_(torch.where(active, p - lr * upd, p))
        opt.flat_m.copy_(m)
        opt.flat_v.copy_(v)

    opt.step = step
    return opt
