"""torch.profiler attribution of one plain + one gradient-penalty training step (which aten ops / autograd Functions
the non-kernel glue time goes to). Test infrastructure.  python tests/gpu_profile_step.py > gpurun_out/step_profile.log"""
import sys
from pathlib import Path

import torch
from torch.profiler import profile, ProfilerActivity

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench   # noqa: E402
from gigagan_pytorch_amd.data import SyntheticImages   # noqa: E402
from gigagan_pytorch_amd.gigagan import cycle   # noqa: E402

dev = torch.device('cuda', 0)
gan = bench.build_gan(256, dev)
it = cycle(SyntheticImages(32, 256, device=dev, seed=0))
for _ in range(4):
    gan.train_step(it, 32)          # steps 1..4 (4 = GP step)
torch.cuda.synchronize()
for label, nsteps in (('PLAIN step (D + G)', 1), ('GP step', None)):
    while label.startswith('GP') and gan._steps_host % 4 != 0:
        gan.train_step(it, 32)
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        gan.train_step(it, 32)
        torch.cuda.synchronize()
    print('=' * 30, label, 'host step', gan._steps_host - 1)
    print(prof.key_averages(group_by_input_shape=True).table(sort_by='self_cuda_time_total', row_limit=70,
                                                             max_name_column_width=60, max_shapes_column_width=90))
