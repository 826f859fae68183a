import torch, sys
sys.path.insert(0, 'nonstationary-precip_amd')
dev = torch.device('cuda', 0)
for dt in (torch.float32, torch.float64):
    K = torch.empty(16384, 16384, dtype=dt, device=dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    best = 1e9
    for _ in range(6):
        ev[0].record(); K.fill_(1.5); ev[1].record(); torch.cuda.synchronize()
        best = min(best, ev[0].elapsed_time(ev[1]))
    print(dt, 'fill_ %.1f us  %.0f GB/s' % (best * 1e3, K.numel() * K.element_size() / best / 1e6))
