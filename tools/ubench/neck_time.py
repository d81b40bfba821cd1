"""HRNet-W48 segmentation neck alone at 512x1024 B=24 (sources 128x256x48 ... 16x32x384): forward + backward, commuted order
(engine/neck.py) vs direct order, ms per call and run-to-run bit equality.   python tools/ubench/neck_time.py [--batch 24]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torchok_amd as T                      # noqa: E402
from torchok_amd.engine import neck as EN    # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=24)
ap.add_argument('--iters', type=int, default=10)
ap.add_argument('--only', default='')
a = ap.parse_args()
chans = (48, 96, 192, 384)
dev = 'cuda:0'
g = torch.Generator().manual_seed(0)
xs = [torch.randn(a.batch, c, 128 >> i, 256 >> i, generator=g).to(dev).to(torch.bfloat16).to(memory_format=torch.channels_last)
      for i, c in enumerate(chans)]
gout = torch.randn(a.batch, sum(chans), 128, 256, generator=g).to(dev).to(torch.bfloat16).to(memory_format=torch.channels_last)
img = torch.zeros(a.batch, 3, 8, 8, device=dev)
for mode in (True, False):
    if a.only and (a.only == 'commuted') != mode:
        continue
    EN.NECK_COMMUTE = mode
    neck = T.NECKS.get('HRNetSegmentationNeck')(chans).to(dev).train()
    snaps = []

    def step():
        xd = [t.detach().requires_grad_(True) for t in xs]
        for p in neck.parameters():
            p.grad = None
        out = neck([img] + xd)[1]
        out.backward(gout)
        return out, xd
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        step()
    e1.record()
    torch.cuda.synchronize()
    for _ in range(2):
        neck.convbnact.bn.reset_running_stats()
        out, xd = step()
        torch.cuda.synchronize()
        snaps.append([out.detach().clone()] + [t.grad.clone() for t in xd] + [p.grad.clone() for p in neck.parameters()])
    same = [bool(torch.equal(u, v)) for u, v in zip(*snaps)]
    print(f"{'commuted' if mode else 'direct  '}: {e0.elapsed_time(e1) / a.iters:.3f} ms per forward + backward;  "
          f"run-to-run bit-equal: {same}")
