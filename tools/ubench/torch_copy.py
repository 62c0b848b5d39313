"""Which engine serves torch's pinned D2H copies?  One pattern per run (argv[1]); run under `rocprofv3 --kernel-trace --stats`
and look for __amd_rocclr_copyBuffer in the kernel list (= the runtime's blit kernel; absent = SDMA)."""
import sys, torch
pat = sys.argv[1]
dev = torch.device('cuda:0')
a = torch.randint(-100, 100, (32, 8, 8, 8, 64), dtype=torch.int32, device=dev)
host16 = torch.empty((32, 64, 8, 8, 8), dtype=torch.int16, pin_memory=True)
host32 = torch.empty((32, 8, 8, 8, 64), dtype=torch.int32, pin_memory=True)
hin = torch.empty((32, 64, 8, 8, 8), dtype=torch.int16, pin_memory=True)
side = torch.cuda.Stream(dev)
main = torch.cuda.current_stream(dev)
big = torch.randn(4096, 4096, device=dev)
torch.cuda.synchronize()
for it in range(10):
    c = big @ big                                   # some work on the main stream
    ev = torch.cuda.Event(); ev.record(main)
    if pat == 'plain':                              # contiguous same-dtype copy on the side stream
        with torch.cuda.stream(side):
            side.wait_event(ev); host32.copy_(a, non_blocking=True)
    elif pat == 'plain_main':
        host32.copy_(a, non_blocking=True)
    elif pat == 'narrow':                           # what _copy_out does: permute + narrow on the side stream, then copy
        with torch.cuda.stream(side):
            side.wait_event(ev)
            s = a.permute(0, 4, 1, 2, 3).contiguous().to(torch.int16)
            host16.copy_(s, non_blocking=True)
    elif pat == 'narrow_h2d':                       # ... with an H2D copy on the main stream at the same time
        with torch.cuda.stream(side):
            side.wait_event(ev)
            s = a.permute(0, 4, 1, 2, 3).contiguous().to(torch.int16)
            host16.copy_(s, non_blocking=True)
        d = hin.to(dev, non_blocking=True)
    elif pat == 'three':                            # three copies back to back
        with torch.cuda.stream(side):
            side.wait_event(ev)
            s = a.permute(0, 4, 1, 2, 3).contiguous().to(torch.int16)
            host16.copy_(s, non_blocking=True); host32.copy_(a, non_blocking=True); host16.copy_(s, non_blocking=True)
    torch.cuda.synchronize()
print(pat, 'done')
