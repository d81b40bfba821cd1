#!/usr/bin/env python
"""Real-size sanity of the rows that bench.py does not cover (SURVEY.md config C5 and the shipped segmentation recipe):
a few training steps each, finite losses, ms per step.  python tools/ubench/scale_sanity.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import torchok_amd as T  # noqa: E402
from torchok_amd.constructor.config import apply_schema  # noqa: E402


def run(name, cfg, batch, steps=6):
    task = T.TASKS.get(cfg.task.name)(cfg, **cfg.task.params).cuda().train()
    opt = task.configure_optimizers()[0]['optimizer']
    losses = []
    for i in range(steps):
        if i == 2:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        out = task.training_step(batch, i)
        opt.zero_grad(set_to_none=True)
        out['loss'].backward()
        opt.step()
        losses.append(out['loss'].detach())
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / (steps - 2)
    vals = [float(v) for v in losses]
    assert all(v == v and abs(v) < 1e4 for v in vals), vals
    print(f'{name}: {ms:.1f} ms/step, losses {vals[0]:.3f} -> {vals[-1]:.3f}')


def main():
    g = torch.Generator(device='cuda').manual_seed(0)
    img = lambda b, h, w: torch.randn(b, 3, h, w, generator=g, device='cuda').to(torch.bfloat16)   # noqa: E731
    opt = [{'optimizer': {'name': 'SGD', 'params': {'lr': 0.01, 'momentum': 0.9, 'weight_decay': 1e-4}}}]
    base = {'data': {}, 'trainer': {'precision': 'bf16'}, 'optimization': opt}
    bb = {'pretrained': False, 'in_channels': 3}
    # C5: ResNet-50 + ArcFace over the 11 318 classes of Stanford Online Products, 224 x 224, B = 128
    cfg = apply_schema(dict(base, task={'name': 'ClassificationTask', 'params': {
        'backbone_name': 'resnet50', 'backbone_params': bb, 'pooling_name': 'PoolingLinear',
        'pooling_params': {'out_channels': 512}, 'head_name': 'ArcFaceHead', 'head_params': {'num_classes': 11318},
        'inputs': [{'shape': [3, 224, 224], 'dtype': 'float32'}]}},
        joint_loss={'losses': [{'name': 'CrossEntropyLoss', 'mapping': {'input': 'prediction', 'target': 'target'}}]}))
    run('resnet50 + PoolingLinear(512) + ArcFaceHead(11318), B=128', cfg,
        {'image': img(128, 224, 224), 'target': torch.randint(0, 11318, (128,), generator=g, device='cuda')})
    # C5: PairwiseLearnTask, ResNet-50 + LinearHead(512, normalize) + ContrastiveLoss, B = 128
    cfg = apply_schema(dict(base, task={'name': 'PairwiseLearnTask', 'params': {
        'backbone_name': 'resnet50', 'backbone_params': bb, 'pooling_name': 'Pooling', 'head_name': 'LinearHead',
        'head_params': {'out_channels': 512, 'normalize': True}, 'inputs': [{'shape': [3, 224, 224], 'dtype': 'float32'}]}},
        joint_loss={'losses': [{'name': 'ContrastiveLoss', 'params': {'margin': 0.5},
                                'mapping': {'emb1': 'emb1', 'emb2': 'emb2', 'R': 'R'}}]}))
    run('resnet50 + LinearHead(512) + ContrastiveLoss (PairwiseLearnTask), B=128', cfg,
        {'image': img(128, 224, 224), 'target': torch.randint(0, 16, (128,), generator=g, device='cuda')})
    # the shipped segmentation recipe at a real size: HRNet-W18 + CE + Dice, 512 x 512, B = 8
    cfg = apply_schema(dict(base, task={'name': 'SegmentationTask', 'params': {
        'backbone_name': 'hrnet_w18', 'backbone_params': bb, 'neck_name': 'HRNetSegmentationNeck',
        'head_name': 'SegmentationHead', 'head_params': {'num_classes': 3},
        'inputs': [{'shape': [3, 512, 512], 'dtype': 'float32'}]}},
        joint_loss={'losses': [{'name': 'CrossEntropyLoss', 'mapping': {'input': 'prediction', 'target': 'target'}},
                               {'name': 'DiceLoss', 'params': {'mode': 'multiclass'},
                                'mapping': {'input': 'prediction', 'target': 'target'}}]}))
    run('hrnet_w18 + HRNetSegmentationNeck + SegmentationHead(3) + CE + Dice, 512x512, B=8', cfg,
        {'image': img(8, 512, 512), 'target': torch.randint(0, 3, (8, 512, 512), generator=g, device='cuda')})


if __name__ == '__main__':
    main()
