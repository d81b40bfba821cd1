"""Which SwinV2-T parameter gradients differ between identical steps (bit-reproducibility probe)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import test_fullsize_properties_gpu as TF
task = TF._swin_task().train()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
x, y = TF._batch(B, seed=2)


def grads():
    for p in task.parameters():
        p.grad = None
    out = task.forward_with_gt({'image': x, 'target': y})
    loss = task.losses(**out)[0]
    torch.cuda.synchronize()
    print('   loss', float(loss.detach().double()), 'prediction checksum', float(out['prediction'].detach().double().abs().sum()), flush=True)
    loss.backward()
    torch.cuda.synchronize()
    return {n: p.grad.detach().clone() for n, p in task.named_parameters() if p.grad is not None}


from torchok_amd.engine import core as EC
EC.BRANCH_STREAMS = False
ref = grads(); ref2 = grads()
print('no-branch reproducible:', all(torch.equal(ref[n], ref2[n]) for n in ref))
EC.BRANCH_STREAMS = True
names = list(ref)
for i in range(4):
    r = grads()
    bad = [n for n in names if not torch.equal(ref[n], r[n])]
    print(f'branch run {i} vs no-branch: {len(bad)} of {len(names)} tensors differ', flush=True)
    if bad:
        print('   first', bad[0], '| last', bad[-1], '| maxdiff', max(float((ref[n] - r[n]).abs().max()) for n in bad))
