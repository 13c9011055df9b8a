import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bench
from pemp_amd import ops
dev = torch.device('cuda:0')
net, sd = bench.build_model(dev)
pool = bench.episode_pool(dev, 1, 4, 0, n_groups=1)
ep = pool[0]
with torch.no_grad():
    p32, _ = net.lowres(ep['sup_img'], ep['sup_mask'], ep['qry_img'])
    p32 = p32.clone()
    with net.precision('bf16'):
        p16, _ = net.lowres(ep['sup_img'], ep['sup_mask'], ep['qry_img'])
        p16 = p16.clone()
print('max |d pred|', (p32 - p16).abs().max().item(), 'mean', (p32 - p16).abs().mean().item(), 'range', p32.abs().max().item())
print('argmax flips at feature resolution', int((p32.argmax(1) != p16.argmax(1)).sum()), 'of', p32[:, 0].numel())
