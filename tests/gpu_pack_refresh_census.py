"""who launches gg_pack_weights, and how often per step? (eager calls counted per step; calls recorded into a hipGraph are listed with their
call site - those are replayed with every step of that kind)"""
import sys, os, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gigagan_pytorch_amd import kernels as K
from gigagan_pytorch_amd.data import SyntheticImages
from gigagan_pytorch_amd.gigagan import cycle

dev = torch.device('cuda', 0)
gan = bench.build_gan(256, dev)
it = cycle(SyntheticImages(32, 256, device=dev, seed=0))
snap = gan.state_snapshot()
orig = K.PackTable.refresh
eager = collections.Counter()
captured = collections.Counter()
step = [0]


def refresh(self):
    site = ' <- '.join(f'{f.name}:{f.lineno}' for f in traceback.extract_stack()[-5:-1][::-1])
    which = 'G' if self is gan.G_opt.pack_table else 'D' if self is gan.D_opt.pack_table else '?'
    if torch.cuda.is_current_stream_capturing():
        captured[(step[0], which, site)] += 1
    else:
        eager[(step[0], which, site)] += 1
    return orig(self)


K.PackTable.refresh = refresh
for s in range(1, 13):
    step[0] = s
    if (gan._steps_host - 1) % 4 == 0:
        gan.state_restore(snap)
    gan.train_step(it, 32)
torch.cuda.synchronize()
print('--- recorded into graphs (replayed with every step of that kind)')
for k, v in sorted(captured.items()):
    print(v, k)
print('--- eager, steps 9..12')
for k, v in sorted(eager.items()):
    if k[0] >= 9:
        print(v, k)
