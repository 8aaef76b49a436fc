"""kernel-level accounting of the REPLAYED graphs (4 timed steps): wall time vs sum of kernel durations, launch count,
share of tiny kernels. Test infrastructure.  python tests/gpu_graph_profile.py > gpurun_out/graph_profile.log"""
import sys
import time
from pathlib import Path

import torch
from torch.profiler import profile, ProfilerActivity

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench   # noqa: E402
from gigagan_pytorch_amd.data import SyntheticImages   # noqa: E402
from gigagan_pytorch_amd.gigagan import cycle   # noqa: E402

dev = torch.device('cuda', 0)
gan = bench.build_gan(256, dev, use_hip_graphs=True)
it = cycle(SyntheticImages(32, 256, device=dev, seed=0))
for _ in range(8):
    gan.train_step(it, 32)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(4):
    gan.train_step(it, 32)
torch.cuda.synchronize()
wall_plain = (time.perf_counter() - t0) * 1e3
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    t0 = time.perf_counter()
    for _ in range(4):
        gan.train_step(it, 32)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
tot = sum(e.device_time for e in evs) / 1e3
print(f'4 steps: wall {wall_plain:.1f} ms unprofiled, {wall:.1f} ms profiled; kernels {len(evs)}; sum of kernel time {tot:.1f} ms')
starts = sorted((e.time_range.start, e.time_range.end) for e in evs)
busy = 0.0; cur_s, cur_e = starts[0]
for s, e in starts[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = starts[-1][1] - starts[0][0]
print(f'GPU busy (union of kernel intervals) {busy/1e3:.1f} ms of span {span/1e3:.1f} ms -> idle {100*(1-busy/span):.1f} %')
for lim in (3, 5, 10, 20):
    small = [e for e in evs if e.device_time < lim]
    print(f'  kernels < {lim} us: {len(small)} ({sum(e.device_time for e in small)/1e3:.1f} ms)')
import collections
agg = collections.defaultdict(lambda: [0, 0.0])
for e in evs:
    a = agg[e.name[:90]]; a[0] += 1; a[1] += e.device_time
for name, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f'{t/1e3:8.2f} ms {n:6d}  {name}')

print('-- kernels shorter than 6 us, by name')
tiny = collections.defaultdict(lambda: [0, 0.0])
for e in evs:
    if e.device_time < 6:
        a = tiny[e.name[:150]]; a[0] += 1; a[1] += e.device_time
for name, (n, t) in sorted(tiny.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f'{n:6d} {t/1e3:7.2f} ms  {name}')
