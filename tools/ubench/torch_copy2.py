"""hipMemcpyAsync called directly (ctypes) from a torch process: torch-pinned vs hipHostMalloc'ed destination, torch stream vs own stream."""
import ctypes, sys, torch
pat = sys.argv[1]
hip = ctypes.CDLL('libamdhip64.so')
dev = torch.device('cuda:0')
a = torch.randint(-100, 100, (32, 8, 8, 8, 64), dtype=torch.int32, device=dev)
n = a.numel() * 4
host_t = torch.empty((32, 8, 8, 8, 64), dtype=torch.int32, pin_memory=True)
own = ctypes.c_void_p()
assert hip.hipHostMalloc(ctypes.byref(own), ctypes.c_size_t(n), ctypes.c_uint(0)) == 0
st = ctypes.c_void_p()
assert hip.hipStreamCreateWithFlags(ctypes.byref(st), ctypes.c_uint(1)) == 0
side = torch.cuda.Stream(dev)
torch.cuda.synchronize()
D2H = 2
for it in range(10):
    if pat == 'own_host_own_stream':
        assert hip.hipMemcpyAsync(own, ctypes.c_void_p(a.data_ptr()), ctypes.c_size_t(n), D2H, st) == 0
    elif pat == 'torch_host_own_stream':
        assert hip.hipMemcpyAsync(ctypes.c_void_p(host_t.data_ptr()), ctypes.c_void_p(a.data_ptr()), ctypes.c_size_t(n), D2H, st) == 0
    elif pat == 'own_host_torch_stream':
        assert hip.hipMemcpyAsync(own, ctypes.c_void_p(a.data_ptr()), ctypes.c_size_t(n), D2H, ctypes.c_void_p(side.cuda_stream)) == 0
    elif pat == 'torch_host_torch_stream':
        assert hip.hipMemcpyAsync(ctypes.c_void_p(host_t.data_ptr()), ctypes.c_void_p(a.data_ptr()), ctypes.c_size_t(n), D2H, ctypes.c_void_p(side.cuda_stream)) == 0
    hip.hipDeviceSynchronize()
print(pat, 'done')
