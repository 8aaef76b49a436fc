"""hipMemsetAsync captured into a hipGraph: is the memset node re-executed on every replay? (torch's split reductions clear their
semaphores with one before each launch: ATen/native/cuda/Reduce.cuh:1301)"""
import ctypes
import torch
dev = torch.device('cuda', 0)
hip = ctypes.CDLL('libamdhip64.so')
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemsetAsync.restype = ctypes.c_int
print(torch.__version__, torch.version.hip)
for nbytes in (4, 8, 16, 32, 64, 128, 256, 1024, 4096, 65536):
    x = torch.ones(1 << 16, dtype=torch.uint8, device=dev)
    y = torch.empty_like(x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        x.add_(1)                       # 2, 3, 4 ... on successive replays
        rc = hip.hipMemsetAsync(x.data_ptr(), 0, nbytes, torch.cuda.current_stream().cuda_stream)
        y.copy_(x)
    assert rc == 0
    res = []
    for i in range(4):
        g.replay()
        torch.cuda.synchronize()
        res.append((int(y[:nbytes].max()), int(y[nbytes:].min()) if nbytes < y.numel() else None))
    print(nbytes, 'bytes: (max of the cleared prefix, min of the rest) per replay', res, flush=True)
