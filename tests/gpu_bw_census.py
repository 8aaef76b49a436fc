"""Which of the non-contraction kernels of one training cycle run far from the HBM roofline? Every python wrapper in kernels.py that is not a
GEMM / attention is bracketed with HIP events during four eager steps (three plain + the gradient-penalty step); bytes = all tensor arguments
and results. Entries >= 8 MB, by total time.   python tests/gpu_bw_census.py [uncond|text|upsampler]"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gigagan_pytorch_amd import kernels as K
from gigagan_pytorch_amd.data import SyntheticImages
from gigagan_pytorch_amd.gigagan import cycle

workload = sys.argv[1] if len(sys.argv) > 1 else 'uncond'
dev = torch.device('cuda', 0)
B = 32 if workload == 'uncond' else 16
gan = bench.build_gan(256, dev, use_hip_graphs=False, workload=workload)
it = iter(bench.SyntheticTextImages(B, 256, dev)) if workload == 'text' else cycle(SyntheticImages(B, 256, device=dev))
SKIP = ('gemm', 'conv2d_nhwc', 'conv2d_wgrad_nhwc', 'conv2d_dgrad_d2s', 'attn_fwd', 'attn_bwd', 'attn_bwd2', 'attn_gen_fwd', 'attn_gen_bwd',
        'capture_graph', 'modw_eligible', 'sconv', 'wgrad_finish', 'colsum_finish')
rec = []


def tensors(o):
    if isinstance(o, torch.Tensor):
        yield o
    elif isinstance(o, (tuple, list)):
        for x in o:
            yield from tensors(x)
    elif isinstance(o, dict):
        for x in o.values():
            yield from tensors(x)


def wrap(name, fn):
    def w(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*a, **k)
        e1.record()
        ts = list(tensors(a)) + list(tensors(k)) + list(tensors(out))
        seen, nbytes = set(), 0
        for t in ts:
            if t.data_ptr() not in seen:
                seen.add(t.data_ptr())
                nbytes += t.numel() * t.element_size()
        big = max(ts, key=lambda t: t.numel()) if ts else None
        rec.append((name, tuple(big.shape) if big is not None else (), nbytes, e0, e1))
        return out
    return w


for _ in range(4):
    gan.train_step(it, B)
torch.cuda.synchronize()
for name in dir(K):
    fn = getattr(K, name)
    if callable(fn) and not name.startswith('_') and getattr(fn, '__module__', '') == K.__name__ and not isinstance(fn, type) and name not in SKIP:
        setattr(K, name, wrap(name, fn))
for _ in range(4):
    gan.train_step(it, B)
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0., 0])
for name, shape, nbytes, e0, e1 in rec:
    a = agg[(name, shape)]
    a[0] += 1
    a[1] += e0.elapsed_time(e1)
    a[2] = nbytes
tot = sum(v[1] for v in agg.values())
print('bracketed wrappers: %.2f ms per step over %d calls per step' % (tot / 4, len(rec) / 4))
byfn = collections.defaultdict(float)
for (name, shape), v in agg.items():
    byfn[name] += v[1]
print('by function (ms/step):', {k: round(v / 4, 2) for k, v in sorted(byfn.items(), key=lambda kv: -kv[1])[:24]})
for (name, shape), v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if v[2] >= 8e6:
        us = v[1] / v[0] * 1e3
        print('%7.3f ms/step x%5.1f  %8.1f us  %7.1f MB  %5.2f TB/s  %-18s %s' % (v[1] / 4, v[0] / 4, us, v[2] / 1e6, v[2] / us / 1e6, name, shape))
