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
gan = bench.build_gan(256, dev, use_hip_graphs=False)
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
    evs = [e for e in prof.key_averages(group_by_input_shape=True) if not e.key.startswith('void ') and not e.key.startswith('gg_')
           and e.self_device_time_total > 0]
    evs.sort(key=lambda e: -e.self_device_time_total)
    tot = sum(e.self_device_time_total for e in evs)
    print(f'non-kernel-named rows: total self device time {tot/1e3:.1f} ms')
    import collections
    by_name = collections.Counter(); by_calls = collections.Counter()
    for e in evs:
        by_name[e.key] += e.self_device_time_total; by_calls[e.key] += e.count
    print('-- aggregated by name')
    for k, t in by_name.most_common(40):
        print(f'{k[:36]:36s} calls {by_calls[k]:5d} self_dev {t/1e3:8.2f} ms')
    print('-- aten / memcpy rows by shape')
    aten = [e for e in evs if e.key.startswith('aten::') or e.key.startswith('Mem')]
    for e in aten[:70]:
        print(f'{e.key[:28]:28s} calls {e.count:5d} self_dev {e.self_device_time_total/1e3:8.2f} ms  shapes {str(e.input_shapes)[:150]}')
