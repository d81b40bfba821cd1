"""Forward of SwinV2-T with and without the branch stream: where do the token rows first differ?"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import test_fullsize_properties_gpu as TF
from torchok_amd.engine import core as EC
task = TF._swin_task().train()
x, y = TF._batch(int(os.environ.get('B', '64')), seed=2)
bb = task.backbone


def feats():
    if os.environ.get('GRAD'):
        f = bb.forward_features(x)
    else:
        with torch.no_grad():
            f = bb.forward_features(x)
    torch.cuda.synchronize()
    return [t.detach().float().clone() for t in f[1:]]


EC.BRANCH_STREAMS = False
a = feats(); a2 = feats()
print('no-branch reproducible', all(torch.equal(p, q) for p, q in zip(a, a2)))
EC.BRANCH_STREAMS = True
for it in range(3):
    b = feats()
    print('branch run', it, [bool(torch.equal(p, q)) for p, q in zip(a, b)], [float((p - q).abs().max()) for p, q in zip(a, b)])
if os.environ.get('SYNC'):
    from torchok_amd.models.backbones import swin as SW
    orig = SW.WindowAttention.prepare
    def prep(self, r, stream=1):
        orig(self, r, stream); torch.cuda.synchronize()
    SW.WindowAttention.prepare = prep
    for it in range(2):
        b = feats()
        print('branch+sync run', it, [bool(torch.equal(p, q)) for p, q in zip(a, b)])
