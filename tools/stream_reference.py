#!/usr/bin/env python3
"""Context for the roofline numbers: what stock streaming kernels (torch copy / row-sum) reach on the SAME operator sizes,
timed with the same graph+events harness as bench.py.  A 30.7 MB pass costs >= 11 us on MI355X whatever the kernel."""
import sys, torch
sys.path.insert(0, '.')
import bench
dev = torch.device('cuda:0')
for shape, tag in [((256, 3, 100, 100), 'cfg-2 operator 30.7 MB'), ((64, 3, 1000, 1000), 'cfg-3 operator 768 MB')]:
    n_sets = 11 if shape[0] == 256 else 2
    Gs = [torch.rand(shape, device=dev) for _ in range(n_sets)]
    out = torch.empty(shape, device=dev)
    small = torch.empty((shape[0] * shape[1], 1), device=dev)
    nbytes = Gs[0].numel() * 4
    ms = bench.time_kernel(lambda i: out.copy_(Gs[i]), n_sets, 60)
    print(tag, 'torch copy (read+write): %.1f us -> %.0f GB/s read+write' % (ms * 1e3, 2 * nbytes / ms / 1e6))
    ms = bench.time_kernel(lambda i: torch.sum(Gs[i].view(shape[0] * shape[1], -1), dim=1, keepdim=True, out=small), n_sets, 60)
    print(tag, 'torch row-sum (read only): %.1f us -> %.0f GB/s' % (ms * 1e3, nbytes / ms / 1e6))
    del Gs, out
