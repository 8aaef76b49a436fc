"""-m gpu: the hipGraph capture helper (kernels.capture_graph) and the repair of captured memset nodes (gg_graph_patch_memsets)."""
import ctypes

import pytest
import torch

from gigagan_pytorch_amd import kernels as K

pytestmark = pytest.mark.gpu


def _warm(fn):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()


@pytest.mark.parametrize('nbytes', [4, 32, 256, 4096, 65536])
def test_captured_memset_writes_its_value_on_every_replay(nbytes):
    """hipMemsetAsync inside a captured graph, replayed four times: the cleared prefix must read 0 every time and the rest of the
    buffer must be untouched. (Unrepaired, this runtime writes 16 / 57 / 64 ... from the second replay on:
    profiles/r04_graph_memset_probe.log.)"""
    dev = torch.device('cuda', 0)
    hip = ctypes.CDLL('libamdhip64.so')
    hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    hip.hipMemsetAsync.restype = ctypes.c_int
    x = torch.ones(1 << 16, dtype=torch.uint8, device=dev)
    y = torch.empty_like(x)
    rcs = []

    def fn():
        x.add_(1)
        rcs.append(hip.hipMemsetAsync(x.data_ptr(), 0, nbytes, torch.cuda.current_stream().cuda_stream))
        y.copy_(x)
        return y

    torch.cuda.synchronize()
    graph, out, n = K.capture_graph(fn)
    assert rcs == [0] and n == 1
    x.fill_(1)
    for r in range(4):
        graph.replay()
        torch.cuda.synchronize()
        assert int(out[:nbytes].max()) == 0, (nbytes, r)
        if nbytes < out.numel():
            assert int(out[nbytes:].min()) == int(out[nbytes:].max()) == r + 2, (nbytes, r)


@pytest.mark.parametrize('shape', [(2048, 1024), (1024, 1024), (512, 4096), (1024, 256)])
def test_captured_column_sum_is_right_on_every_replay(shape):
    """torch's split column reduction (semaphores cleared by a memset, ATen/native/cuda/Reduce.cuh:1301) - what folds per-workgroup
    gradient partials in the step - captured through kernels.capture_graph and replayed with fresh inputs."""
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    x = torch.randn(*shape, device=dev)
    _warm(lambda: x.sum(0))
    graph, y, n = K.capture_graph(lambda: x.sum(0))
    for r in range(5):
        x.copy_(torch.randn(*shape, device=dev) * 3. ** r)
        graph.replay()
        torch.cuda.synchronize()
        want = x.double().sum(0)
        assert float((y.double() - want).abs().max() / want.abs().max()) < 1e-5, (shape, r, n)
