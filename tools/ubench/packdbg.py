import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch, bench
from torchok_amd.engine import functional as EF
from torchok_amd.engine.step import train_step
task = bench.build_task('resnet50', 1000).cuda().train()
opt = task.configure_optimizers()[0]['optimizer']
x = torch.randn(256, 3, 224, 224, device='cuda').to(torch.bfloat16); y = torch.randint(0, 1000, (256,), device='cuda')
b = {'image': x, 'target': y}
for i in range(4): train_step(task, opt, b, i)
names = {p.data_ptr(): n for n, p in task.named_parameters()}
orig = EF.get_packs
log = []
def spy(weight, bias, kp, sp, cp, want_dgrad, refresh):
    pk = EF._packs_for(weight)
    key = (weight.data_ptr(), kp, sp, cp, weight.device)
    stale = pk.key != key or (refresh and pk.synced != weight._version)
    will = (want_dgrad and (pk.fwd is None or pk.dgrad is None or stale)) or ((not want_dgrad) and (pk.fwd is None or stale))
    if will: log.append((names.get(weight.data_ptr(), '?'), want_dgrad, pk.fwd is None, pk.dgrad is None, stale, pk.key != key))
    return orig(weight, bias, kp, sp, cp, want_dgrad, refresh)
EF.get_packs = spy
train_step(task, opt, b, 5)
torch.cuda.synchronize()
for l in log: print(l)
print(len(log))
