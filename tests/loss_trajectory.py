"""Diagnostic script (GPU, not collected by pytest): per-step loss of bench.py's default workload (ResNet-50,
B=256, SGD lr 0.1 / momentum 0.9 / wd 1e-4, one fixed random batch) on the HIP path next to the oracle model
(oracle/torchok_ref.py) run by torch on the same GPU in fp32 and under bf16 autocast, from the same initial
state.  Shows how far a trajectory on random labels at lr 0.1 is reproducible at all: the runs agree for the
first steps and then separate by O(1) — the reason bench.py's `final_loss` moves whenever the order of
two bf16 gradient additions changes.

    python tests/loss_trajectory.py [--steps 25] [--batch 256] [--old-order]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def product_run(args, image, target, state=None, old_order=False):
    import bench
    from torchok_amd.engine import functional as EF
    from torchok_amd.models.backbones import resnet as RN
    if old_order:
        from torchok_amd import engine

        def fwd(self, x):     # projection shortcut recorded after conv2 (the tape order of the earlier r01 runs)
            r = engine.current_region()
            y = EF.conv_bn_act(r, x, self.conv1, self.bn1, relu=True)
            y = EF.conv_bn_act(r, y, self.conv2, self.bn2, relu=True)
            shortcut = RN._shortcut_branch(r, x, self.downsample)
            return EF.conv_bn_act(r, y, self.conv3, self.bn3, relu=True, shortcut=shortcut)
        RN.Bottleneck.forward = fwd
    torch.manual_seed(1234)
    task = bench.build_task('resnet50', 1000).cuda().train()
    if state is not None:
        task.load_state_dict(state, strict=False)
    init = {k: v.detach().clone() for k, v in task.state_dict().items() if not k.startswith('input_tensors')}
    opt = task.configure_optimizers()[0]['optimizer']
    losses = []
    for i in range(args.steps):
        out = task.training_step({'image': image, 'target': target}, i)
        opt.zero_grad(set_to_none=True)
        out['loss'].backward()
        opt.step()
        losses.append(out['loss'].detach())
    torch.cuda.synchronize()
    return [float(v) for v in losses], init


def torch_run(args, image, target, init, autocast):
    import oracle.torchok_ref as R
    model = R.ClassificationModel('resnet50', 1000).cuda().train()
    missing = model.load_state_dict(init, strict=False)
    assert not [k for k in missing.missing_keys if 'num_batches_tracked' not in k], missing
    opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
    x = image.float()
    losses = []
    for _ in range(args.steps):
        with torch.autocast('cuda', dtype=torch.bfloat16, enabled=autocast):
            loss, _ = R.training_step(model, {'image': x, 'target': target}, opt)
        losses.append(float(loss))
    return losses


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=25)
    ap.add_argument('--batch', type=int, default=256)
    args = ap.parse_args()
    g = torch.Generator(device='cuda').manual_seed(1234)
    image = torch.randn(args.batch, 3, 224, 224, generator=g, device='cuda').to(torch.bfloat16)
    target = torch.randint(0, 1000, (args.batch,), generator=g, device='cuda')
    new, init = product_run(args, image, target)
    again, _ = product_run(args, image, target, init)
    ac = torch_run(args, image, target, init, True)
    fp = torch_run(args, image, target, init, False)
    old, _ = product_run(args, image, target, init, old_order=True)
    print('step   hip(now)  hip(again)  hip(old order)  torch bf16-autocast  torch fp32')
    for i in range(args.steps):
        print(f'{i:4d}  {new[i]:9.4f}  {again[i]:9.4f}  {old[i]:13.4f}  {ac[i]:19.4f}  {fp[i]:10.4f}')
    assert new == again, 'the HIP path is not run-to-run deterministic'


if __name__ == '__main__':
    main()
